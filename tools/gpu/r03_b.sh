#!/bin/bash
# round 3: GPU tests + the 125K-column shard solve + the default solve after the fused block step / two-launch quantiser
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/gpu.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
BSN_TIMING=1 timeout 300 python bench.py --variants 125000 --steps 6 --warmup 2 --no-cpu-baseline --no-ingest > $O/b125.json 2> $O/b125.err
python - <<'P'
import json; d=json.load(open('gpurun_out/r03b/b125.json')); print('m=125000:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})
P
grep "host wall" $O/b125.err | tail -1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p125 -o st -- python $GRAFT_REPO_ROOT/bench.py --variants 125000 --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > /dev/null 2> /tmp/p125.err
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/p125 -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f > $O/timeline_125k.txt; tail -3 $O/timeline_125k.txt
for b in 8 16; do
timeout 300 python bench.py --block $b --steps 4 --warmup 1 --no-cpu-baseline --no-ingest > $O/b$b.json 2> $O/b$b.err
python - <<P
import json; d=json.load(open('gpurun_out/r03b/b$b.json')); print('block $b:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['sigma'][:2])
P
done
