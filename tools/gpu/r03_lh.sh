#!/bin/bash
# k_prod<2> with lane halves (BSN_TUNE=89, ablation build): correctness against the shipped kernel, then timing
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03lh; mkdir -p $O
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
cat > /tmp/lh_check.py <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bigsnpr_amd as ba
gb = ba.bed.synthetic(5000, 300000, seed=5)
gb.tile()
r = ba.bed_randomSVD(gb, k=10, block=16)
print("TUNE", os.environ.get("BSN_TUNE"), "d", repr(r["d"][:3]), "niter", r["niter"], "usum %.12e" % np.abs(r["u"]).sum())
P
BSN_TUNE=0 python /tmp/lh_check.py 2>&1 | grep TUNE
BSN_TUNE=89 python /tmp/lh_check.py 2>&1 | grep TUNE
one() { l=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-ingest > $O/$l.json 2> $O/$l.err
  python - <<P
import json
try:
  d=json.load(open('$O/$l.json')); print('$l:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, 'sigma1 %.9f' % d['sigma'][0])
except Exception as e: print('$l: FAILED', e)
P
}
for t in 0 89 0 89; do BSN_TUNE=$t one lh_t$t --steps 3 --warmup 1 --no-uv; done
