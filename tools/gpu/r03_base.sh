#!/bin/bash
# round-3 baseline: the 125K-column shard solve (per-GPU work of an 8-GPU run) and the block-16 solve, before the changes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
BSN_TIMING=1 timeout 300 python bench.py --variants 125000 --steps 6 --warmup 2 --no-cpu-baseline --no-ingest > $O/b125.json 2> $O/b125.err
python - <<'P'
import json; d=json.load(open('gpurun_out/r03a/b125.json')); print('m=125000:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})
P
grep "host wall" $O/b125.err | tail -2
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p125 -o st -- python $GRAFT_REPO_ROOT/bench.py --variants 125000 --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > /dev/null 2> /tmp/p125.err
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/p125 -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f > $O/timeline_125k.txt; tail -6 $O/timeline_125k.txt
timeout 300 python bench.py --block 16 --steps 4 --warmup 1 --no-cpu-baseline --no-ingest > $O/b16.json 2> $O/b16.err
python - <<'P'
import json; d=json.load(open('gpurun_out/r03a/b16.json')); print('block16:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})
P
