#!/bin/bash
# round 3: result download inside the timed solve; block 8 / 16 at k = 10 / 20; two more k_prod<2> variants
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
one() { l=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-ingest > $O/$l.json 2> $O/$l.err
  python - <<P
import json
try:
  d=json.load(open('$O/$l.json')); print('$l:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})
except Exception as e: print('$l: FAILED', e)
P
}
one b8_nouv --steps 4 --warmup 1 --no-uv
one b8_uv --steps 4 --warmup 1
one b16_uv --block 16 --steps 4 --warmup 1
one k10_b8 --k 10 --block 8 --steps 3 --warmup 1
one k10_b16 --k 10 --block 16 --steps 3 --warmup 1
one k5_b8 --k 5 --block 8 --steps 3 --warmup 1
one k5_b16 --k 5 --block 16 --steps 3 --warmup 1
one k40_b8 --k 40 --block 8 --steps 2 --warmup 1
one k40_b16 --k 40 --block 16 --steps 2 --warmup 1
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for t in 0 97 99; do BSN_TUNE=$t one nb2_t$t --block 16 --steps 3 --warmup 1 --no-uv; done
