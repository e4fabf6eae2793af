#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/gpu.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
BSN_TIMING=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ingest > $O/bdef.json 2> $O/bdef.err
python - <<'P'
import json; d=json.load(open('gpurun_out/r03h/bdef.json')); print('default:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'block', d['config']['block'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, 'value', d['value'])
P
grep "host wall" $O/bdef.err | tail -2
timeout 600 python bench.py --block 8 --steps 6 --warmup 2 --no-cpu-baseline --no-ingest > $O/b8.json 2> $O/b8.err
python - <<'P'
import json; d=json.load(open('gpurun_out/r03h/b8.json')); print('block 8:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'block', d['config']['block'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, 'value', d['value'])
P
