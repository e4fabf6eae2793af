#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_prs_tcrossprod.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 600 python tools/probe_tcross.py --n 8192 --m 32768 2>&1 | grep entry
timeout 600 python tools/probe_tcross.py --n 2000 --m 100000 --reps 1 2>&1 | grep entry
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python tools/probe_tcross.py --n 16384 --m 65536 --reps 1 2>&1 | grep entry
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:5]:
    print("%-60s calls %5s avg_ms %9.2f" % (r[0][:60], r[1], float(r[3])/1e6))
PY
