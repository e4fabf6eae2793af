#!/usr/bin/env python3
"""Where the first call of the outlier step of snp_autoSVD goes (run on the GPU box): imports, the first dist_ogk on the
device, the following ones — with BSN_ALLOC_TRACE=1 the library prints its device allocations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter(); import numpy as np; t_np = time.perf_counter() - t0
t0 = time.perf_counter(); import bigsnpr_amd as ba; from bigsnpr_amd import autosvd, _lib; L = _lib.load(); t_pkg = time.perf_counter() - t0
t0 = time.perf_counter(); from scipy.stats import chi2, norm; t_sp = time.perf_counter() - t0
print("imports: numpy %.3f s, bigsnpr_amd + library %.3f s, scipy.stats %.3f s" % (t_np, t_pkg, t_sp), flush=True)
t0 = time.perf_counter(); d = _lib.DeviceArray.from_numpy(np.zeros(8)); d.free(); print("first device allocation (runtime start): %.3f s" % (time.perf_counter() - t0), flush=True)
rng = np.random.default_rng(0)
U = np.asfortranarray(rng.normal(size=(1000000, 10)))
for rep in range(3):
    t0 = time.perf_counter(); S = autosvd.dist_ogk(U, device=True); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); S2 = autosvd.rollmean_groups(np.sqrt(S), 50, [(1, np.arange(U.shape[0]))], device=True); t2 = time.perf_counter() - t0
    t0 = time.perf_counter(); thr = autosvd.tukey_mc_up(S2, device=True); t3 = time.perf_counter() - t0
    print("call %d: dist_ogk %.3f s, rollmean %.3f s, tukey_mc_up %.3f s" % (rep + 1, t1, t2, t3), flush=True)
