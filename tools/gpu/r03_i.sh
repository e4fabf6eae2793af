#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/gpu.log | tail -4
one() { l=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-ingest > $O/$l.json 2> $O/$l.err
  python - <<P
import json
try:
  d=json.load(open('$O/$l.json')); print('$l:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'block', d['config']['block'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})
except Exception as e: print('$l: FAILED', e)
P
}
one b125 --variants 125000 --steps 8 --warmup 2 --no-uv
BSN_NO_SPECULATION=1 one b125_nospec --variants 125000 --steps 8 --warmup 2 --no-uv
one bdef --steps 8 --warmup 2
BSN_NO_SPECULATION=1 one bdef_nospec --steps 8 --warmup 2
one b8 --block 8 --steps 6 --warmup 2
bash tools/gpu/r03_d.sh
cp gpurun_out/r03d/* $O/
