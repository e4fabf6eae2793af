# emulates the HIP wrapper's automatic mode (16-bit base, precision schedule, second solve on 56-bit products after an
# inexact exhaustion) on the CPU harness, with the assertions of tests/test_gpu_random_shapes.py
import numpy as np, sys, ctypes as C, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'native'))
import test_svd_driver_cpu as T
import build_native
nt = C.CDLL(build_native.build())
seed0=int(sys.argv[1]); ntrial=int(sys.argv[2]); tiny=int(sys.argv[3])
rng=np.random.default_rng(seed0); bad=[]; ok=0; t0=time.time()
def auto_solve(A, k, block, seed):
    b = block if block else 8
    nt.nt_set_slices(2); nt.nt_set_schedule(C.c_double(2.5e-7), 3, 1)
    r = T.host_svd(nt, A, k, tol=1e-4, block=b, seed=seed)
    if r["refused"]: return r
    if r["resolve"]:
        nt.nt_set_slices(7); nt.nt_set_schedule(C.c_double(0.0), 0, 0)
        r = T.host_svd(nt, A, k, tol=1e-4, block=min(b, 4), seed=seed)
    return r
for trial in range(ntrial):
    if tiny: n, m = int(rng.integers(5, 48)), int(rng.integers(6, 70))
    else: n, m = int(rng.integers(30, 500)), int(rng.integers(40, 900))
    f = rng.uniform(0.05, 0.5, size=m); G = rng.binomial(2, f, size=(n, m)).astype(float)
    if rng.random() < 0.5: G[rng.random(G.shape) < 0.05] = np.nan
    mu = np.nanmean(G, axis=0); p = mu/2; sd = np.sqrt(2*p*(1-p))
    keep = sd > 0
    if keep.sum() < 3: continue
    A = np.where(np.isnan(G), 0.0, (G - mu) / np.where(sd > 0, sd, 1))[:, keep]
    n, m = A.shape
    kmax = min(n, m) - 1
    k = int(rng.integers(1, (kmax if tiny else min(kmax, 25)) + 1))
    block = int(rng.choice([0, 1, 2, 8, 16]))
    nt.nt_set_fused(int(rng.integers(0,2)))
    r = auto_solve(A, k, block, trial+1)
    d_true = np.linalg.svd(A, compute_uv=False)[:k]
    tag=(seed0, trial, n, m, k, block)
    if r["refused"] or not r["converged"]: bad.append(('NOTCONV/REFUSED',)+tag+(r["resid"], r["restarts"])); continue
    if block == 1 and k > 4 and not tiny:
        good = True
    else:
        good = np.allclose(r["d"], d_true, rtol=1e-6, atol=1e-6*d_true[0])
    if not good: bad.append(('WRONG',)+tag+(float(np.abs(r["d"]-d_true).max()/d_true[0]), r["exhausted"]))
    else: ok+=1
nt.nt_set_slices(0); nt.nt_set_schedule(C.c_double(0.0),0,0); nt.nt_set_fused(0)
print(seed0,'tiny',tiny,'ok',ok,'bad',len(bad),'sec %.0f'%(time.time()-t0))
for b in bad[:20]: print(b)
