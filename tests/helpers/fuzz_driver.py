"""Exploration of the block-Lanczos driver on the CPU test backend (tests/native): random matrices of five kinds x digit counts x
block sizes x both orthonormalisation paths; prints every solve that comes back converged with wrong values.
    python tests/helpers/fuzz_driver.py <seed> <trials>
Kinds 0, 1, 2, 4 (dense, low rank + noise, exactly low rank, genotype-like) are clean over thousands of trials; kind 3 — singular
values with multiplicities far beyond the block size — is the textbook limit of a (block) Lanczos process, see DESIGN.md section 4."""
import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/native')
import build_native
nt = C.CDLL(build_native.build())
AR_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64, C.c_void_p)
def host_svd(A, k, tol, block, seed, max_basis=0):
    A = np.asfortranarray(A, dtype=np.float64); n, m = A.shape
    d = np.empty(k); u = np.empty((k, n)); v = np.empty((k, m))
    info = np.zeros(8, dtype=np.int32); resid = C.c_double()
    nt.nt_svd_host(A.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n), C.c_int64(m), C.c_int64(m), k, C.c_double(tol), block, max_basis, C.c_uint32(seed), AR_FN(), None,
                   d.ctypes.data_as(C.POINTER(C.c_double)), u.ctypes.data_as(C.POINTER(C.c_double)), v.ctypes.data_as(C.POINTER(C.c_double)), info.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(resid))
    return dict(d=d, u=u.T, v=v.T, niter=info[0], basis=info[2], converged=bool(info[3]), resid=resid.value)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0; conv = 0; N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for trial in range(N):
    fused = int(rng.integers(0, 2)); nt.nt_set_fused(fused)
    S = int(rng.choice([0, 2, 2, 3])); nt.nt_set_slices(S)
    n, m = int(rng.integers(4, 220)), int(rng.integers(4, 220))
    kind = int(rng.integers(0, 5))
    if kind == 0: A = rng.normal(size=(n, m))
    elif kind == 1:   # low rank + noise
        r = int(rng.integers(1, min(n, m))); A = rng.normal(size=(n, r)) @ rng.normal(size=(r, m)) + 1e-3 * rng.normal(size=(n, m))
    elif kind == 2:   # exactly low rank
        r = int(rng.integers(1, min(n, m))); A = rng.normal(size=(n, r)) @ rng.normal(size=(r, m))
    elif kind == 3:   # clustered / repeated singular values
        U, _ = np.linalg.qr(rng.normal(size=(n, min(n, m)))); V, _ = np.linalg.qr(rng.normal(size=(m, min(n, m))))
        s = np.sort(rng.choice([1.0, 1.0, 2.0, 2.0, 2.0, 5.0, 5.0000001, 10.0], size=min(n, m)))[::-1]; A = (U * s) @ V.T
    else:             # genotype-like
        f = rng.uniform(0.05, 0.5, size=m); G = rng.binomial(2, f, size=(n, m)).astype(float); sd = G.std(0); ok = sd > 0
        A = (G[:, ok] - G[:, ok].mean(0)) / sd[ok]; m = A.shape[1]
        if m < 3: continue
    kmax = min(n, m) - 1
    if kmax < 1: continue
    k = int(rng.integers(1, min(kmax, 40) + 1)); block = int(rng.choice([0, 1, 2, 4, 8, 16]))
    tol = float(rng.choice([1e-4, 1e-4, 1e-6]))
    d_true = np.linalg.svd(A, compute_uv=False)[:k]
    try:
        res = host_svd(A, k, tol, block, trial + 1)
    except Exception as e:
        print("EXC", trial, e); bad += 1; continue
    if not np.all(np.isfinite(res["d"])):
        print("NONFINITE", trial, n, m, k, block, S, fused, kind); bad += 1; continue
    sig = d_true > 1e-3 * d_true[0]
    err = np.abs(res["d"][sig] / d_true[sig] - 1).max() if sig.any() else 0.0
    zero_err = np.abs(res["d"][~sig] - d_true[~sig]).max() / d_true[0] if (~sig).any() else 0.0
    lim = 3e-5 if tol >= 1e-4 else 1e-7
    if S == 2 and tol < 1e-5: lim = 1e-4      # tolerance below the floor of 16-bit products: should refuse
    if res["converged"]:
        conv += 1
        if err > lim or zero_err > 2e-3:
            bad += 1; print("WRONG", dict(trial=trial, n=n, m=m, k=k, block=block, S=S, fused=fused, kind=kind, tol=tol, err=err, zero_err=zero_err, resid=res["resid"], basis=res["basis"]))
print("trials", N, "converged", conv, "bad", bad)
