#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for args in "--m 125000" "--m 125000 --force-dist" "--m 250000 --force-dist" "--m 500000 --force-dist"; do
echo "== $args"; BSN_TIMING=1 timeout 300 python bench.py $args --steps 6 --warmup 2 --no-cpu-baseline --no-ingest 2>&1 >/tmp/b.json | grep "host wall" | tail -1 | sed -e 's/.*solve/solve/'; python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['ms_per_step'], d['niter'], round(d['passes_per_solve'],2), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})"
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --m 125000 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > /tmp/kt.log 2>&1
cd $GRAFT_REPO_ROOT; T=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T | tail -6
