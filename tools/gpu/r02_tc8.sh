#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_prs_tcrossprod.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
BSN_TCROSS_WAVES=4 timeout 600 python -m pytest tests/test_gpu_prs_tcrossprod.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -1
for w in 8 4; do
rm -rf /tmp/kt; BSN_TCROSS_WAVES=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python tools/probe_tcross.py --n 16384 --m 65536 --reps 1 2>&1 | grep entry
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
r=list(csv.reader(open(sys.argv[1])))[1]
print("   %-40s avg_ms %9.2f -> %.1f TFLOP/s executed" % (r[0][:40], float(r[3])/1e6, 1.773e13/(float(r[3])/1e9)/1e12))
PY
done
timeout 300 python tools/probe_tcross.py --n 2000 --m 100000 --reps 2 2>&1 | grep entry
timeout 300 python tools/probe_tcross.py --n 8192 --m 32768 --reps 2 2>&1 | grep entry
