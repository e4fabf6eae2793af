#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02r
timeout 300 python tools/probe_calls.py 2>&1 | grep entry
for e in bed_prodVec bed_cprodVec bed_counts; do
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python tools/probe_calls.py --only $e > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); echo "== $e"; python - "$f" <<'PY'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:10]:
    print("%-60s calls %5s avg_us %9.1f tot_ms %8.2f" % (r[0][:60], r[1], float(r[3])/1e3, float(r[2])/1e6))
PY
done 2>&1 | tee gpurun_out/r02r/calls.txt
