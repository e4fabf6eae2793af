import numpy as np, sys, ctypes as C, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'native'))
import test_svd_driver_cpu as T
import build_native
nt = C.CDLL(build_native.build())
seed0 = int(sys.argv[1]); ntrial = int(sys.argv[2])
rng = np.random.default_rng(seed0)
bad=[]; ok=0; t0=time.time()
for trial in range(ntrial):
    n, m = int(rng.integers(6, 300)), int(rng.integers(6, 300))
    kind = int(rng.integers(0, 6))
    if kind == 0:     # exact low rank
        r = int(rng.integers(1, min(n, m, 12))); A = rng.normal(size=(n, r)) @ rng.normal(size=(r, m))
    elif kind == 1:   # duplicated columns
        base = rng.normal(size=(n, max(2, m // 4))); A = base[:, rng.integers(0, base.shape[1], size=m)]
    elif kind == 2:   # repeated singular values
        U,_ = np.linalg.qr(rng.normal(size=(n, min(n,m)))); V,_ = np.linalg.qr(rng.normal(size=(m, min(n,m))))
        s = np.repeat(rng.uniform(1, 10, size=(min(n,m)+2)//3), 3)[:min(n,m)]; s = np.sort(s)[::-1]; A = (U*s) @ V.T
    elif kind == 3:   # zero rows / columns
        A = rng.normal(size=(n, m)); A[rng.random(n) < 0.3] = 0; A[:, rng.random(m) < 0.3] = 0
    elif kind == 4:   # tiny / huge scale
        A = rng.normal(size=(n, m)) * 10.0 ** int(rng.choice([-120, -30, 30, 120]))
    else:             # geometric spectrum
        U,_ = np.linalg.qr(rng.normal(size=(n, min(n,m)))); V,_ = np.linalg.qr(rng.normal(size=(m, min(n,m))))
        s = 2.0 ** -np.arange(min(n,m)); A = (U*s) @ V.T
    if not np.any(A): continue
    kmax = min(n, m) - 1
    if kmax < 1: continue
    k = int(rng.integers(1, min(kmax, 20) + 1))
    block = int(rng.choice([1, 2, 3, 4, 8, 16])); S = int(rng.choice([0, 2, 3])); fused = int(rng.integers(0, 2))
    nt.nt_set_fused(fused); nt.nt_set_slices(S)
    d_true = np.linalg.svd(A, compute_uv=False)
    np.save("/tmp/last.npy", A)
    res = T.host_svd(nt, A, k, tol=1e-4, block=block, seed=trial + 1)
    d = res["d"]
    if res.get("refused"):
        if kind == 4: ok += 1
        else: bad.append(('REFUSED', seed0, trial, kind, n, m, k, block, S, fused))
        continue
    tag = (seed0, trial, kind, n, m, k, block, S, fused)
    if not np.all(np.isfinite(d)) or not np.all(np.isfinite(res["u"])) or not np.all(np.isfinite(res["v"])):
        bad.append(('NONFINITE',) + tag); continue
    sig = d_true[:k] > 1e-3 * d_true[0]
    err = np.abs(d[sig] / d_true[:k][sig] - 1).max() if sig.any() else 0.0
    small_ok = np.all(d[~sig] <= 2e-3 * d_true[0] * (50 if S else 1) + 1e-300)
    if not res["converged"]:
        np.save('/tmp/f3_%d_%d.npy' % (seed0, trial), A)
        bad.append(('NOTCONV',) + tag + (float(err), float(res["resid"]), int(res["restarts"])))
    elif err > 1e-4 or not small_ok:
        np.save('/tmp/f3w_%d_%d.npy' % (seed0, trial), A)
        near = np.abs(d[sig][:, None] / d_true[None, :] - 1).min(axis=1).max() if sig.any() else 0
        bad.append((('SKIPPED' if near < 1e-4 else 'WRONG') if err > 1e-4 else 'SMALLBAD',) + tag + (float(err), float(res["resid"]), [float(x) for x in d[~sig][:3]], float(d_true[0])))
    else: ok += 1
nt.nt_set_slices(0); nt.nt_set_fused(0)
print(seed0, 'ok', ok, 'bad', len(bad), 'sec %.0f' % (time.time()-t0))
for b in bad[:25]: print(b)
