#!/bin/bash
# one-shot calls at C2 (50 000 x 200 000): grid shapes of the two streaming kernels (ablation build)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c2s; mkdir -p $O
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
cat > /tmp/c2probe.py <<'P'
import sys, time, numpy as np
sys.path.insert(0, '.')
import bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()
n, m = 50000, 200000
gb = ba.bed.synthetic(n, m, seed=9)
sc = ba.bed_scaleBinom(gb)
rng = np.random.default_rng(0)
x, y = rng.normal(size=m), rng.normal(size=n)
def t(fn, reps=40):
    for _ in range(4): fn()
    L.bsn_device_sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    L.bsn_device_sync(); return 1e3 * (time.perf_counter() - t0) / reps
p = t(lambda: ba.bed_prodVec(gb, x, center=sc["center"], scale=sc["scale"]))
c = t(lambda: ba.bed_cprodVec(gb, y, center=sc["center"], scale=sc["scale"]))
print("%s prodVec %.3f ms  cprodVec %.3f ms" % (sys.argv[1], p, c))
P
for rep in 1 2 3; do
for tv in 0 28 29 25; do BSN_TUNE=$tv python /tmp/c2probe.py tune$tv 2>/dev/null | tee -a $O/sweep2.txt; done
done
