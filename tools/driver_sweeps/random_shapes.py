import numpy as np, sys, ctypes as C, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'native'))
import test_svd_driver_cpu as T
import build_native
nt = C.CDLL(build_native.build())
seed0 = int(sys.argv[1]); ntrial = int(sys.argv[2])
rng = np.random.default_rng(seed0)
bad = []
stats = dict(ok=0, notconv=0, wrong=0)
t0=time.time()
for trial in range(ntrial):
    shape = rng.choice(4)
    if shape == 0: n, m = int(rng.integers(5, 80)), int(rng.integers(100, 2500))
    elif shape == 1: n, m = int(rng.integers(100, 1500)), int(rng.integers(5, 80))
    elif shape == 2: n, m = int(rng.integers(20, 400)), int(rng.integers(20, 400))
    else: n, m = int(rng.integers(330, 700)), int(rng.integers(330, 900))
    kind = rng.choice(3)
    if kind == 0:
        f = rng.uniform(0.05, 0.5, size=m); A = rng.binomial(2, f, size=(n, m)).astype(float)
        sd = A.std(axis=0); keep = sd > 0
        if keep.sum() < 4: continue
        A = (A[:, keep] - A[:, keep].mean(0)) / sd[keep]
    elif kind == 1:
        A = rng.normal(size=(n, m)) * rng.uniform(0.3, 3.0, size=m); A -= A.mean(0)
    else:  # low rank + noise: separated leading values
        r = int(rng.integers(1, 8)); A = rng.normal(size=(n, r)) @ rng.normal(size=(r, m)) * 3 + rng.normal(size=(n, m)); A -= A.mean(0)
    n, m = A.shape
    kmax = min(n, m) - 1
    if kmax < 1: continue
    k = int(rng.integers(1, min(kmax, 25) + 1))
    block = int(rng.choice([1, 2, 3, 4, 8, 16]))
    S = int(rng.choice([0, 2, 3]))
    fused = int(rng.integers(0, 2))
    maxb = int(rng.choice([0, 0, 0, k + 2 * block + 2, 2 * k + block + 8]))
    nt.nt_set_fused(fused); nt.nt_set_slices(S)
    d_true = np.linalg.svd(A, compute_uv=False)
    res = T.host_svd(nt, A, k, tol=1e-4, block=block, max_basis=maxb, seed=trial + 1)
    sig = d_true[:k] > 1e-3 * d_true[0]
    err = np.abs(res["d"][sig] / d_true[:k][sig] - 1).max() if sig.any() else 0.0
    tag = (seed0, trial, n, m, k, block, S, fused, maxb, int(kind), bool(res["converged"]), float(err), float(res["resid"]), int(res["restarts"]), int(res["niter"]))
    if not res["converged"]:
        stats['notconv'] += 1; bad.append(('NOTCONV',) + tag); np.save('/tmp/fail_%d_%d.npy' % (seed0, trial), A)
    elif err > 2e-5:
        # is every returned value a true singular value (missed member of a cluster)?
        near = np.abs(res["d"][sig][:, None] / d_true[None, :] - 1).min(axis=1).max()
        np.save('/tmp/fail_%d_%d.npy' % (seed0, trial), A); stats['wrong'] += 1; bad.append(('WRONG' if near > 2e-5 else 'SKIPPED',) + tag + (float(near),))
    else:
        stats['ok'] += 1
nt.nt_set_slices(0); nt.nt_set_fused(0)
print(seed0, stats, 'sec %.0f' % (time.time()-t0))
for b in bad: print(b)
