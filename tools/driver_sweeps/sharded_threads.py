import numpy as np, sys, ctypes as C, time, threading
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'native'))
import test_svd_driver_cpu as T
import build_native
nt = C.CDLL(build_native.build())
seed0 = int(sys.argv[1]); ntrial = int(sys.argv[2])
rng = np.random.default_rng(seed0)
bad=[]; ok=0
t0=time.time()
for trial in range(ntrial):
    R = int(rng.integers(2, 5))
    shape = rng.choice(3)
    if shape == 0: n, m = int(rng.integers(8, 80)), int(rng.integers(60, 1500))
    elif shape == 1: n, m = int(rng.integers(100, 900)), int(rng.integers(R * 4, 90))
    else: n, m = int(rng.integers(60, 500)), int(rng.integers(60, 700))
    kind = rng.choice(2)
    if kind == 0:
        A = rng.normal(size=(n, m)) * rng.uniform(0.3, 3.0, size=m); A -= A.mean(0)
    else:
        r = int(rng.integers(1, 8)); A = rng.normal(size=(n, r)) @ rng.normal(size=(r, m)) * 3 + rng.normal(size=(n, m)); A -= A.mean(0)
    kmax = min(n, m) - 1
    k = int(rng.integers(1, min(kmax, 20) + 1))
    block = int(rng.choice([1, 2, 3, 4, 8, 16]))
    S = int(rng.choice([0, 2, 3])); fused = int(rng.integers(0, 2))
    nt.nt_set_fused(fused); nt.nt_set_slices(S)
    # ragged shards (a rank may get few columns)
    cuts = np.sort(rng.choice(np.arange(1, m), R - 1, replace=False)); cuts = np.concatenate([[0], cuts, [m]])
    bar = threading.Barrier(R)
    bufs = [None] * R; results = [None] * R; errs = []
    lock = threading.Lock()
    def make_ar(rank):
        def ar(buf, count, ctx):
            a = np.ctypeslib.as_array(buf, shape=(count,))
            bufs[rank] = a
            bar.wait()
            if rank == 0:
                tot = np.zeros(count)
                for r_ in range(R): tot += bufs[r_]      # fixed order: identical on all ranks
                bufs.append(tot)
            bar.wait()
            a[:] = bufs[-1]
            bar.wait()
            if rank == 0: bufs.pop()
            bar.wait()
        return ar
    def work(rank):
        try:
            results[rank] = T.host_svd(nt, A[:, cuts[rank]:cuts[rank+1]], k, tol=1e-4, block=block, m_total=m, ar=make_ar(rank), seed=trial + 1)
        except Exception as e:
            errs.append(repr(e)); bar.abort()
    th = [threading.Thread(target=work, args=(r_,)) for r_ in range(R)]
    [t.start() for t in th]; [t.join(120) for t in th]
    tag = (seed0, trial, R, n, m, k, block, S, fused, int(kind), [int(c) for c in np.diff(cuts)])
    if errs or any(r is None for r in results):
        bad.append(('ERROR',) + tag + (errs[:1],)); continue
    d_true = np.linalg.svd(A, compute_uv=False)[:k]
    res = results[0]
    same = all(np.array_equal(results[r_]["u"], res["u"]) and np.array_equal(results[r_]["d"], res["d"]) and results[r_]["converged"] == res["converged"] for r_ in range(R))
    sig = d_true > 1e-3 * d_true[0]
    err = np.abs(res["d"][sig] / d_true[sig] - 1).max() if sig.any() else 0.0
    if not same: bad.append(('RANKS DIFFER',) + tag)
    elif not res["converged"]: bad.append(('NOTCONV',) + tag + (float(err), float(res["resid"]), int(res["restarts"])))
    elif err > 1e-4: bad.append(('WRONG',) + tag + (float(err), float(res["resid"])))
    else: ok += 1
nt.nt_set_slices(0); nt.nt_set_fused(0)
print(seed0, 'ok', ok, 'bad', len(bad), 'sec %.0f' % (time.time()-t0))
for b in bad: print(b)
