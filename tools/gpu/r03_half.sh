#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03half; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd.py tests/test_gpu_tiled.py tests/test_gpu_fullsize.py tests/test_gpu_matvec.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
one() { l=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-ingest > $O/$l.json 2> $O/$l.err
  python - <<P
import json
try:
  d=json.load(open('$O/$l.json')); print('$l:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'block', d['config']['block'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['sigma'][:1])
except Exception as e: print('$l: FAILED', e)
P
}
one half --steps 6 --warmup 2
BSN_PROD_NO_HALF=1 one nohalf --steps 6 --warmup 2
one half2 --steps 6 --warmup 2
