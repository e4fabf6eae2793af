import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()
gb = ba.bed.synthetic(50000, 200000)
sc = ba.bed_scaleBinom(gb)
rng = np.random.default_rng(1)
x, y = rng.normal(size=200000), rng.normal(size=50000)
for name, fn in (("prod", lambda: ba.bed_prodVec(gb, x, center=sc["center"], scale=sc["scale"])),
                 ("cprod", lambda: ba.bed_cprodVec(gb, y, center=sc["center"], scale=sc["scale"])),
                 ("prod", lambda: ba.bed_prodVec(gb, x, center=sc["center"], scale=sc["scale"]))):
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); fn(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(name, ts, flush=True)
