"""CPU tests of the backend-independent SVD driver (bigsnpr_amd/csrc/svd_driver.hpp) with
a dense host backend, including the column-sharded world_size-2 path over gloo."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "native"))

AR_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64, C.c_void_p)


@pytest.fixture(scope="module")
def nt():
    import build_native
    lib = C.CDLL(build_native.build())
    return lib


@pytest.fixture(params=[0, 1], ids=["stepwise", "fused"], autouse=True)
def fused(request, nt):
    """every driver test runs twice: with the step-by-step orthonormalisation of svd_driver.hpp and with the
    fused two-pass block step the HIP backend takes (same small-matrix code, orth_small.hpp, host loops for
    the tall products)"""
    nt.nt_set_fused(request.param)
    yield request.param
    nt.nt_set_fused(0)


def host_svd(nt, A, k, tol=1e-10, block=8, m_total=None, ar=None, max_basis=0, seed=1):
    A = np.asfortranarray(A, dtype=np.float64)
    n, m = A.shape
    d = np.empty(k); u = np.empty((k, n)); v = np.empty((k, m))
    info = np.zeros(8, dtype=np.int32); resid = C.c_double()
    cb = AR_FN(ar) if ar is not None else AR_FN()
    nt.nt_svd_host(A.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n), C.c_int64(m),
                   C.c_int64(m_total or m), k, C.c_double(tol), block, max_basis, C.c_uint32(seed),
                   cb, None, d.ctypes.data_as(C.POINTER(C.c_double)),
                   u.ctypes.data_as(C.POINTER(C.c_double)), v.ctypes.data_as(C.POINTER(C.c_double)),
                   info.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(resid))
    return dict(d=d, u=u.T, v=v.T, niter=info[0], nops=info[1], basis=info[2],
                converged=bool(info[3]), resid=resid.value, restarts=int(info[4]), refused=bool(info[5]),
                exhausted=bool(info[6]), resolve=bool(info[7]), below_resolution=bool(info[7] & 2))


def test_eig_sym_matches_numpy(nt):
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 37, 120):
        B = rng.normal(size=(n, n)); S = B + B.T
        if n == 37:  # block tridiagonal like the Lanczos matrix, with repeated eigenvalues
            S = np.kron(np.eye(n // 2 + 1), np.ones((2, 2)))[:n, :n] + np.diag(np.ones(n - 2), 2) + np.diag(np.ones(n - 2), -2)
        A = np.asfortranarray(S.copy()); d = np.empty(n)
        nt.nt_eig_sym(n, A.ctypes.data_as(C.POINTER(C.c_double)), d.ctypes.data_as(C.POINTER(C.c_double)))
        w = np.linalg.eigvalsh(S)
        np.testing.assert_allclose(d, w, atol=1e-12 * max(1, np.abs(w).max()))
        np.testing.assert_allclose(A @ np.diag(d) @ A.T, S, atol=1e-11 * max(1, np.abs(w).max()))
        np.testing.assert_allclose(A.T @ A, np.eye(n), atol=1e-12)


def _check_svd(res, A, k, tol_d=1e-8, tol_vec=1e-6):
    U, d, Vt = np.linalg.svd(A, full_matrices=False)
    np.testing.assert_allclose(res["d"], d[:k], rtol=tol_d)
    for t in range(k):
        if t + 1 < len(d) and (d[t] - d[t + 1]) / d[0] < 1e-6:
            continue
        s = np.sign(res["u"][:, t] @ U[:, t])
        assert np.abs(s * res["u"][:, t] - U[:, t]).max() < tol_vec
        assert np.abs(s * res["v"][:, t] - Vt[t]).max() < tol_vec


@pytest.mark.parametrize("n,m,k,block", [(200, 500, 10, 8), (300, 120, 10, 8), (60, 45, 10, 4),
                                         (33, 70, 10, 8), (500, 400, 20, 8), (150, 150, 3, 1),
                                         (12, 40, 10, 8), (40, 25, 5, 5), (300, 11, 10, 5)])
def test_block_lanczos_dense(nt, n, m, k, block):
    rng = np.random.default_rng(n + m)
    r = min(n, m)
    A = rng.normal(size=(n, r)) @ np.diag(np.linspace(1, 30, r) ** 1.5) @ rng.normal(size=(r, m)) / np.sqrt(r)
    res = host_svd(nt, A, k, tol=1e-12, block=block)
    assert res["converged"]
    _check_svd(res, A, k)


def test_genotype_matrix_matches_oracle_dense_svd(nt, orc, example_bed):
    # tests/testthat/test-2-bed-clumping-SVD.R:52-54,76-79 through the oracle's dense SVD
    ic = np.arange(0, example_bed.m, 3)
    ref = orc.dense_svd(example_bed, None, ic, k=10)
    A = orc.read_bed_scaled(example_bed, None, ic, ref["center"], ref["scale"])
    res = host_svd(nt, A, 10, tol=1e-12)
    np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-9)
    assert np.abs(res["u"].mean(0)).max() < 1e-8
    # default tolerance of the reference (1e-4): d still within 1e-6
    res2 = host_svd(nt, A, 10, tol=1e-4)
    np.testing.assert_allclose(res2["d"], ref["d"], rtol=1e-6)
    assert res2["nops"] <= res["nops"]


def test_rank_deficient_and_tiny(nt):
    rng = np.random.default_rng(5)
    A = rng.normal(size=(80, 6)) @ rng.normal(size=(6, 50))   # rank 6 < k
    res = host_svd(nt, A, 10, tol=1e-10)
    d = np.linalg.svd(A, compute_uv=False)
    np.testing.assert_allclose(res["d"][:6], d[:6], rtol=1e-8)
    assert np.all(res["d"][6:] < 1e-6 * d[0])


_WORKER = r'''
import ctypes as C, os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from test_svd_driver_cpu import host_svd, AR_FN
import build_native
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
nt = C.CDLL(build_native.build())
nt.nt_set_fused(int(sys.argv[4]))
rng = np.random.default_rng(42)
n, m, k = 180, 260, 10
A = rng.normal(size=(n, 40)) @ np.diag(np.linspace(1, 20, 40)) @ rng.normal(size=(40, m)) + 0.01 * rng.normal(size=(n, m))
cols = np.arange(m)[rank::2] if False else np.arange(rank * m // 2, (rank + 1) * m // 2)
def ar(buf, count, ctx):
    a = np.ctypeslib.as_array(buf, shape=(count,))
    t = torch.from_numpy(a)
    dist.all_reduce(t)
res = host_svd(nt, A[:, cols], k, tol=1e-12, m_total=m, ar=ar)
U, d, Vt = np.linalg.svd(A, full_matrices=False)
assert res["converged"]
np.testing.assert_allclose(res["d"], d[:k], rtol=1e-9)
for t in range(k):
    s = np.sign(res["u"][:, t] @ U[:, t])
    assert np.abs(s * res["u"][:, t] - U[:, t]).max() < 1e-7
    assert np.abs(s * res["v"][:, t] - Vt[t, cols]).max() < 1e-7
# u identical on both ranks (bitwise): gather and compare
t = torch.from_numpy(np.ascontiguousarray(res["u"]))
out = [torch.empty_like(t) for _ in range(2)]
dist.all_gather(out, t)
assert torch.equal(out[0], out[1])
dist.barrier()
print("rank", rank, "ok", res["niter"], flush=True)
dist.destroy_process_group()   # without it gloo's threads are torn down at interpreter exit and may abort
sys.stdout.flush()
os._exit(0)
'''


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_sharded_two_ranks_gloo(tmp_path, nt, fused):
    """N>1 path: columns sharded over 2 ranks, W all-reduced (gloo on CPU); both ranks must
    converge to the global SVD and hold identical u.  (`nt` builds the test library once in this
    process so that the two workers only load it.)"""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests", "native"))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(fused)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_ritz_values_do_not_see_the_product_precision(nt):
    """The driver rounds finished basis blocks to the backend's grid and solves (Z'Z) s = theta (Q'Q) s,
    so 16-bit fixed-point products (slices = 2) give the same singular values as exact ones up to
    the convergence level — the old projected matrix built from the rounded expansion was off by
    ~1e-6 (numpy prototype of both variants in the round's notes)."""
    rng = np.random.default_rng(42)
    n, m, k = 600, 450, 8
    A = rng.normal(size=(n, 60)) @ np.diag(np.linspace(1, 40, 60) ** 1.3) @ rng.normal(size=(60, m)) / 8 \
        + 0.3 * rng.normal(size=(n, m))
    d_true = np.linalg.svd(A, compute_uv=False)[:k]
    try:
        errs = {}
        for S, tol in ((0, 1e-4), (2, 1e-4), (3, 1e-6), (0, 1e-6)):
            nt.nt_set_slices(S)
            res = host_svd(nt, A, k, tol=tol, block=8)
            assert res["converged"]
            errs[(S, tol)] = np.abs(res["d"] / d_true - 1).max()
            # u and v stay consistent with the returned d:  A' u = d v  exactly for the stored basis
            np.testing.assert_allclose(A.T @ res["u"], res["v"] * res["d"], atol=1e-9 * d_true[0])
        # an unattainable tolerance (below the rounding floor of 16-bit products) is reported, not faked
        nt.nt_set_slices(2)
        nt.nt_set_max_restarts(3)
        assert not host_svd(nt, A, k, tol=1e-7, block=8, max_basis=64)["converged"]
        nt.nt_set_max_restarts(100)
    finally:
        nt.nt_set_slices(0)
    # the error of d is the convergence level (~tol^2) with or without rounded products
    assert errs[(2, 1e-4)] < 5e-8 and errs[(0, 1e-4)] < 5e-8, errs
    assert errs[(3, 1e-6)] < 1e-10 and errs[(0, 1e-6)] < 1e-10, errs


def test_full_basis_is_not_mistaken_for_convergence(nt):
    """a cap on the Krylov basis that is reached before the residuals meet tol: the driver must say
    'not converged' (RSpectra warns in that case); the clipped next block must not zero the estimate"""
    rng = np.random.default_rng(5)
    A = rng.normal(size=(400, 300))
    nt.nt_set_max_restarts(-1)
    try:
        res = host_svd(nt, A, 10, tol=1e-12, block=4, max_basis=16)
    finally:
        nt.nt_set_max_restarts(100)
    assert not res["converged"] and res["basis"] == 16
    assert res["resid"] > 1e-6
    assert np.all(np.isfinite(res["d"])) and np.all(np.diff(res["d"]) <= 0)
    # with room for the whole space the same call converges
    res = host_svd(nt, A, 10, tol=1e-12, block=4, max_basis=0)
    assert res["converged"]


def test_fused_step_is_taken_and_hands_deficient_panels_to_the_careful_path(nt, fused):
    """the fused two-pass step (orth_small.hpp) must carry a normal solve on its own — also with a basis that is
    only orthonormal up to the rounding of its blocks (slices = 2: Q'Q = I + 1e-5) — and must give a panel that
    lost its rank back to the step-by-step path instead of normalising noise"""
    if not fused:
        pytest.skip("fused mode only")
    cnt = (C.c_int * 2)()
    rng = np.random.default_rng(3)
    A = rng.normal(size=(500, 60)) @ np.diag(np.linspace(1, 40, 60) ** 1.3) @ rng.normal(size=(60, 400)) / 8 \
        + 0.3 * rng.normal(size=(500, 400))
    d_true = np.linalg.svd(A, compute_uv=False)[:10]
    try:
        for S, tol in ((0, 1e-8), (2, 1e-4), (3, 1e-6)):
            nt.nt_set_slices(S)
            res = host_svd(nt, A, 10, tol=tol, block=8)
            nt.nt_fused_counts(cnt)
            assert res["converged"] and cnt[0] >= res["niter"] and cnt[1] == 0, (S, tol, cnt[0], cnt[1], res["niter"])
            np.testing.assert_allclose(res["d"], d_true, rtol=max(1e-10, 10 * tol * tol))
            np.testing.assert_allclose(res["u"].T @ res["u"], np.eye(10), atol=1e-9)
    finally:
        nt.nt_set_slices(0)
    # exact rank 6 < block: the second panel is rank deficient -> careful path, same answer as before
    A = rng.normal(size=(80, 6)) @ rng.normal(size=(6, 50))
    res = host_svd(nt, A, 10, tol=1e-10)
    nt.nt_fused_counts(cnt)
    assert cnt[1] >= 1
    np.testing.assert_allclose(res["d"][:6], np.linalg.svd(A, compute_uv=False)[:6], rtol=1e-8)


def test_thick_restart_of_a_full_basis(nt):
    """a basis that fills up before the residuals meet tol is compressed to the best Ritz vectors and the solve goes
    on (what RSpectra's implicit restart does for the reference): a hard spectrum with a small basis cap must still
    converge to the dense SVD, with exact products and with the rounded basis of the product (slices = 2)"""
    rng = np.random.default_rng(11)
    n, m, k = 500, 700, 12
    A = rng.normal(size=(n, m)) + rng.normal(size=(n, 15)) @ np.diag(np.linspace(0.16, 0.09, 15)) @ rng.normal(size=(15, m))
    U, d, Vt = np.linalg.svd(A, full_matrices=False)
    try:
        for S, tol, blk in ((0, 1e-8, 8), (2, 1e-4, 8), (2, 1e-4, 16), (3, 1e-6, 5)):
            nt.nt_set_slices(S)
            res = host_svd(nt, A, k, tol=tol, block=blk, max_basis=48)
            assert res["converged"] and res["restarts"] >= 1, (S, tol, blk, res["restarts"], res["resid"])
            assert res["basis"] <= 48
            np.testing.assert_allclose(res["d"], d[:k], rtol=max(1e-10, 20 * tol * tol))
            np.testing.assert_allclose(res["u"].T @ res["u"], np.eye(k), atol=1e-8)
            np.testing.assert_allclose(A.T @ res["u"], res["v"] * res["d"], atol=1e-8 * d[0])
    finally:
        nt.nt_set_slices(0)


def test_converged_is_never_claimed_on_faith(nt):
    """small matrices of every shape on 16-bit products (slices = 2): the Krylov space is exhausted on the way (rank +
    block <= basis), and what is left of a nearly cancelled panel is amplified rounding noise.  Whatever the driver
    returns as converged must be right to the convergence level; an exhaustion that is not is answered by a thick restart
    (round 5) and, if that does not get there, reported as such (with automatic digits the HIP wrapper also solves again
    on 56-bit products, tests/test_gpu_svd.py).  The case that started this: 39 x 17, k = 16,
    8 vectors per pass — 1.5e-3 off and "converged" before the coupling block of the last step was asked."""
    rng = np.random.default_rng(2024)
    claimed = refused = 0
    try:
        nt.nt_set_slices(2)
        for trial in range(120):
            n, m = int(rng.integers(5, 60)), int(rng.integers(5, 60))
            A = rng.normal(size=(n, m)) * rng.uniform(0.3, 3.0, size=m)
            A -= A.mean(0)
            kmax = min(n, m) - 1
            k = int(rng.integers(1, kmax + 1))
            block = int(rng.choice([1, 2, 4, 8, 16]))
            d_true = np.linalg.svd(A, compute_uv=False)[:k]
            res = host_svd(nt, A, k, tol=1e-4, block=block, seed=trial + 1)
            sig = d_true > 1e-3 * d_true[0]
            err = np.abs(res["d"][sig] / d_true[sig] - 1).max()
            if res["converged"]:
                claimed += 1
                assert err < 2e-5, (n, m, k, block, err, res["resid"])
            else:
                refused += 1
    finally:
        nt.nt_set_slices(0)
    assert claimed > 60, (claimed, refused)


def _sines(U, Ut):
    U = U / np.linalg.norm(U, axis=0)
    c = np.abs((U * Ut).sum(0))
    return np.sqrt(np.maximum(0.0, 1.0 - c * c))


def test_precision_schedule_gives_the_vectors_of_wide_panels(nt, fused):
    """Round 5 (SvdOptions::vec_floor): the early block steps on 24-bit panels, the late ones on 16 bits.  On a matrix
    with separated leading singular values (the genotype case: population structure over a noise bulk) the leading
    vectors of the scheduled solve are as close to the true ones as those of the solve that runs EVERY step on 24
    bits, and two orders of magnitude closer than the 16-bit solve's; the singular values agree in all three."""
    rng = np.random.default_rng(11)
    n, m, r, k, b = 400, 1200, 12, 8, 8
    edge = np.sqrt(n) + np.sqrt(m)
    s = edge * 12 * 0.95 ** np.arange(r)
    U0 = np.linalg.qr(rng.normal(size=(n, r)))[0]
    V0 = np.linalg.qr(rng.normal(size=(m, r)))[0]
    A = rng.normal(size=(n, m)) + (U0 * s) @ V0.T
    Ut, d, _ = np.linalg.svd(A, full_matrices=False)
    out = {}
    log = np.zeros(64, dtype=np.int32)
    try:
        for name, (S, vf, smax, gain) in dict(narrow=(2, 0.0, 0, 0.0), wide=(3, 0.0, 0, 0.0), scheduled=(2, 2.5e-7, 3, 0.0),
                                              split=(2, 2.5e-7, 3, np.sqrt(n))).items():
            nt.nt_set_slices(S)
            nt.nt_set_schedule(C.c_double(vf), smax, 0)
            nt.nt_set_noise_gain(C.c_double(gain))
            res = host_svd(nt, A, k, tol=1e-4, block=b)
            assert res["converged"]
            np.testing.assert_allclose(res["d"], d[:k], rtol=1e-6)
            cnt = nt.nt_schedule_log(log.ctypes.data_as(C.POINTER(C.c_int32)), 64)
            out[name] = (_sines(res["u"], Ut[:, :k]), res["niter"], list(log[:cnt]))
    finally:
        nt.nt_set_slices(0)
        nt.nt_set_schedule(C.c_double(0.0), 0, 0)
        nt.nt_set_noise_gain(C.c_double(0.0))
    lead = slice(0, k // 2)
    sn, sw, ss = out["narrow"][0][lead].max(), out["wide"][0][lead].max(), out["scheduled"][0][lead].max()
    sched = out["scheduled"][2]
    # (log: every set_precision call — the start block, then per block step the digits of the product pass, the grid
    # of the block it produces and, while the schedule is wide, that grid again once the step's residuals are known)
    # the schedule starts wide and ends narrow, and the solve takes the steps of the uniform ones
    assert sched[0] == 2 and sched[1] == 3 and sched[2] == 3 and sched[-1] == 2 and sched[-2] == 2, sched
    assert sorted(sched[1:], reverse=True) == sched[1:], sched      # never wider again
    assert out["scheduled"][1] == out["wide"][1] == out["narrow"][1] == out["split"][1]
    # with the amplification of a random vector known (sqrt(n) against singular values of 12 x the bulk edge) the
    # PRODUCT passes go narrow earlier than the grids, and the leading vectors stay within the floor that was asked
    # for (2.5e-7 in the residual: angles below north_star's 1e-6)
    split = out["split"][2]
    assert split.count(3) < sched.count(3), (split, sched)
    assert out["split"][0][lead].max() < 1e-6 and out["split"][0][lead].max() < sn / 8, (out["split"][0][lead].max(), sn)
    assert sn > 3e-6, sn                       # the 16-bit floor is visible on this matrix ...
    assert ss < 3.0 * sw and ss < sn / 30, (sn, sw, ss)   # ... and gone with the early steps on 24 bits


def test_fewer_rows_than_the_basis_limit_on_rounded_products(nt):
    """34 x 1701, k = 6 (tests/test_gpu_random_shapes.py at seed offset 3000): the basis can hold the whole of R^34.
    On 16-bit products the projected block at p = 32 is two genuine directions and six of rounding noise; the new
    block is bounded by dim - p, the space is then exhausted with exact Ritz pairs — no restarts, no residual made
    of noise."""
    rng = np.random.default_rng(19)
    A = rng.integers(0, 3, size=(34, 1701)).astype(float)
    A = (A - A.mean(axis=0)) / np.maximum(A.std(axis=0), 1e-9)
    dref = np.linalg.svd(A, compute_uv=False)
    try:
        for S in (2, 3, 0):
            nt.nt_set_slices(S)
            for block in (8, 4, 16, 3):
                r = host_svd(nt, A, 6, tol=1e-4, block=block)
                assert r["converged"] and r["restarts"] == 0 and r["basis"] == 34, (S, block, r["resid"], r["restarts"])
                assert np.abs(r["d"] / dref[:6] - 1).max() < 1e-8
    finally:
        nt.nt_set_slices(0)


def test_an_inexact_exhaustion_is_restarted(nt):
    """Tall matrices with few columns: the Krylov space (rank + block dimensions) is used up before the basis limit.
    In floating point it is then only NEARLY invariant — single-vector steps over a spectrum of 314 … 13 on exact
    products end with residuals of 1e-2, rounded products likewise — and the coupling block says so.  The driver
    continues from a thick restart instead of ending there: converged, right values (a random sweep of 1 600 small
    shapes: all nine exhaustions that used to end unconverged)."""
    rng = np.random.default_rng(269)
    try:
        for (n, m, r, k, block, S) in ((316, 23, 4, 18, 1, 0), (570, 27, 3, 18, 1, 3), (450, 72, 5, 25, 4, 3),
                                       (1117, 44, 2, 16, 2, 2), (243, 80, 6, 20, 16, 3)):
            A = rng.normal(size=(n, r)) @ rng.normal(size=(r, m)) * 3 + rng.normal(size=(n, m))
            A -= A.mean(0)
            d_true = np.linalg.svd(A, compute_uv=False)[:k]
            nt.nt_set_slices(S)
            res = host_svd(nt, A, k, tol=1e-4, block=block, seed=7)
            assert res["converged"], (n, m, k, block, S, res["resid"], res["restarts"])
            assert np.abs(res["d"] / d_true - 1).max() < 2e-5
    finally:
        nt.nt_set_slices(0)


def test_sharded_random_shapes_in_threads(nt):
    """2 - 4 ranks as threads of this process (the all-reduce hook sums the ranks' buffers in a fixed order behind a
    barrier), ragged column shards down to a single column, every block size, exact / 16-bit / 24-bit products: every
    rank ends with the same bits, converged, the values of a dense solver.  (2 000 further draws of this sweep were
    clean; one vector per pass may skip a member of a cluster — checked as in tests/test_gpu_random_shapes.py.)"""
    import threading
    rng = np.random.default_rng(5)
    try:
        for trial in range(40):
            R = int(rng.integers(2, 5))
            shape = trial % 3
            if shape == 0:
                n, m = int(rng.integers(8, 80)), int(rng.integers(60, 900))
            elif shape == 1:
                n, m = int(rng.integers(100, 600)), int(rng.integers(R * 4, 90))
            else:
                n, m = int(rng.integers(60, 300)), int(rng.integers(60, 400))
            r = int(rng.integers(1, 8))
            A = rng.normal(size=(n, r)) @ rng.normal(size=(r, m)) * 3 + rng.normal(size=(n, m))
            A -= A.mean(0)
            k = int(rng.integers(1, min(min(n, m) - 1, 16) + 1))
            block = int(rng.choice([1, 2, 3, 4, 8, 16]))
            nt.nt_set_slices(int(rng.choice([0, 2, 3])))
            cuts = np.concatenate([[0], np.sort(rng.choice(np.arange(1, m), R - 1, replace=False)), [m]])
            bar = threading.Barrier(R)
            bufs, results, errs = [None] * R, [None] * R, []

            def make_ar(rank):
                def ar(buf, count, ctx):
                    a = np.ctypeslib.as_array(buf, shape=(count,))
                    bufs[rank] = a
                    bar.wait()
                    tot = np.zeros(count)
                    for q in range(R):
                        tot += bufs[q]          # the same order on every rank
                    bar.wait()
                    a[:] = tot
                    bar.wait()
                return ar

            def work(rank):
                try:
                    results[rank] = host_svd(nt, A[:, cuts[rank]:cuts[rank + 1]], k, tol=1e-4, block=block, m_total=m,
                                             ar=make_ar(rank), seed=trial + 1)
                except Exception as e:      # pragma: no cover
                    errs.append(repr(e))
                    bar.abort()

            th = [threading.Thread(target=work, args=(q,)) for q in range(R)]
            [t.start() for t in th]
            [t.join(120) for t in th]
            assert not errs and all(x is not None for x in results), (trial, errs)
            res = results[0]
            for q in range(1, R):
                assert np.array_equal(results[q]["u"], res["u"]) and np.array_equal(results[q]["d"], res["d"])
            assert res["converged"], (trial, R, n, m, k, block)
            d_true = np.linalg.svd(A, compute_uv=False)
            sig = d_true[:k] > 1e-3 * d_true[0]
            if block == 1:
                assert np.abs(res["d"][sig][:, None] / d_true[None, :] - 1).min(axis=1).max() < 1e-4
            else:
                assert np.abs(res["d"][sig] / d_true[:k][sig] - 1).max() < 1e-4, (trial, R, n, m, k, block)
    finally:
        nt.nt_set_slices(0)


def test_non_finite_and_out_of_range_matrices_are_refused(nt):
    """NaN / Inf in the projected matrices used to send the eigen-solver's deflation search past the end of its arrays
    (heap overflow under ASan; found by a sweep over matrices scaled by 1e120); a matrix of the order of 1e120 came back
    "converged" with d = 0, one of 1e-100 with d wrong by a factor of two — its Gram matrices overflow / fall into the
    denormals.  The eigen-solver now returns NaN for such input and the driver refuses it with a message."""
    n = 7
    V = np.asfortranarray(np.eye(n))
    V[2, 3] = V[3, 2] = np.nan
    d = np.zeros(n)
    nt.nt_eig_sym(n, V.ctypes.data_as(C.POINTER(C.c_double)), d.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.isnan(d).all()
    rng = np.random.default_rng(8)
    A = rng.normal(size=(182, 26))
    for S in (0, 2):
        nt.nt_set_slices(S)
        try:
            for scale in (1e120, 1e-100):
                assert host_svd(nt, A * scale, 3, tol=1e-4, block=16)["refused"]
            B = A.copy()
            B[5, 7] = np.inf
            assert host_svd(nt, B, 3, tol=1e-4, block=8)["refused"]
            B[5, 7] = np.nan
            assert host_svd(nt, B, 3, tol=1e-4, block=8)["refused"]
            for scale in (1e60, 1e-60):     # inside the range: solved, and d scales with the matrix
                r = host_svd(nt, A * scale, 3, tol=1e-4, block=16)
                assert r["converged"] and not r["refused"]
                np.testing.assert_allclose(r["d"], np.linalg.svd(A, compute_uv=False)[:3] * scale, rtol=1e-6)
        finally:
            nt.nt_set_slices(0)


def test_more_triplets_than_rank_on_rounded_products(nt):
    """rank 4, k = 5 .. 7, 16-bit products (and the precision schedule): the triplets beyond the rank have Ritz values
    at the level the rounding of the stored basis leaves (theta < 1e-9 theta_1) and relative residuals that mean
    nothing; the convergence test is over the others.  Before: 100 restarts, 4 000 block steps, invented singular values
    (98 and 296 beside 207 .. 151), once under "converged"."""
    rng = np.random.default_rng(144)
    A = rng.normal(size=(223, 4)) @ rng.normal(size=(4, 133))
    d_true = np.linalg.svd(A, compute_uv=False)
    try:
        nt.nt_set_slices(2)
        for sched in (0, 1):
            nt.nt_set_schedule(C.c_double(2.5e-7 if sched else 0.0), 3 if sched else 0, 1 if sched else 0)
            for block in (16, 8, 4, 2):
                for k in (3, 4, 5, 7):
                    r = host_svd(nt, A, k, tol=1e-4, block=block, seed=145)
                    assert r["converged"] and r["restarts"] == 0 and r["niter"] <= 6, (sched, block, k, r["niter"], r["restarts"])
                    kk = min(k, 4)
                    np.testing.assert_allclose(r["d"][:kk], d_true[:kk], rtol=1e-6)
                    assert np.all(r["d"][4:] < 1e-3 * d_true[0])
                    # (round 6: triplets between the hard zero, 1e-10 theta_1, and the products' resolution raise
                    # below_resolution — here the null directions come out at either side of 1e-10; never for k <= rank)
                    assert not (r["below_resolution"] and k <= 4), (sched, block, k)
    finally:
        nt.nt_set_slices(0)
        nt.nt_set_schedule(C.c_double(0.0), 0, 0)


def test_dependent_columns_of_a_panel_keep_their_couplings(nt):
    """17 samples (some of them all zero), 16 vectors per pass, k = 1, exact products: after the first step R^17 has ONE
    direction left, and all 16 columns of the projected panel are multiples of it.  The careful orthonormalisation keeps
    one column — and must keep the couplings of the other 15 to it: they are the residuals of the step.  With only the
    kept column's coupling the solve ended "converged" after one step, up to 3.7 % off (found by the sweep over
    degenerate matrices; these seeds are ten of the 17 in 1 500 draws that did)."""
    for seed in (131, 262, 347, 460, 724, 763, 824, 832, 981, 1106):
        rng = np.random.default_rng(seed)
        m = int(rng.integers(60, 300))
        A = rng.normal(size=(17, m))
        A[rng.random(17) < 0.3] = 0
        A[:, rng.random(m) < 0.3] = 0
        d_true = np.linalg.svd(A, compute_uv=False)
        r = host_svd(nt, A, 1, tol=1e-4, block=16, seed=seed + 1)
        assert r["converged"] and r["niter"] >= 2
        np.testing.assert_allclose(r["d"], d_true[:1], rtol=2e-5)


def test_small_but_real_triplets_are_not_vouched_for(nt):
    """300 x 400 with sigma = 100, 80, 60, 40 and a tail at 1e-2, k = 6 on 16-bit products.  The tail is
    REAL (sigma = 1e-4 sigma_1) but below what 16-bit products resolve, so the convergence test leaves triplets 5 and 6 out —
    and used to return converged = 1 with d[4:6] 10 % off.  The driver now counts such triplets (below_resolution) and the
    caller acts on it: what svd.hip does in its automatic mode — the same solve on 56-bit products — gets them right."""
    rng = np.random.default_rng(5)
    U, _ = np.linalg.qr(rng.normal(size=(300, 300)))
    V, _ = np.linalg.qr(rng.normal(size=(400, 400)))
    sig = np.r_[100.0, 80.0, 60.0, 40.0, np.linspace(1e-2, 5e-3, 296)]
    A = (U * sig) @ V[:, :300].T
    k = 6
    try:
        nt.nt_set_slices(2)
        r = host_svd(nt, A, k, tol=1e-4, block=8, seed=7)
        assert r["below_resolution"], "the two tail triplets lie below the resolution of 16-bit products: the driver must say so"
        np.testing.assert_allclose(r["d"][:4], sig[:4], rtol=1e-6)       # the four it tested are right
        nt.nt_set_slices(7)                                               # the wrapper's second solve
        r7 = host_svd(nt, A, k, tol=1e-4, block=4, seed=7)
        assert r7["converged"] and not r7["below_resolution"]
        np.testing.assert_allclose(r7["d"], sig[:k], rtol=1e-6)
        nt.nt_set_slices(0)                                               # exact products never raise the flag
        r0 = host_svd(nt, A, k, tol=1e-4, block=8, seed=7)
        assert r0["converged"] and not r0["below_resolution"]
        np.testing.assert_allclose(r0["d"], sig[:k], rtol=1e-6)
    finally:
        nt.nt_set_slices(0)


def test_careful_path_projects_until_clean_on_an_eight_bit_start_block(nt):
    """Round 6, found by the fixed sweep (tests/test_svd_driver_sweep_cpu.py): 27 x 98 with singular values over eight decades,
    k = 11, block 8, the wrapper's automatic mode (8-bit start grid since round 5 + precision schedule).  The stored basis is
    orthonormal only up to the rounding of its coarsest block (|Q'Q - I| = 4e-3 with the 8-bit start block) and the panels of
    such a spectrum lose six digits in the projection: the two projection passes of the careful path left more of the OLD
    directions in the panel than there was of the new one (|Q'Q - I| = 0.63 .. 0.80 at steps 3 - 4), and the exhausted space
    (all of R^27) returned sigma_10, sigma_11 44 % and 103 % off under converged = 1.  The path now projects until a pass
    removes nothing of what is left."""
    rng = np.random.default_rng(77)
    for n, m in ((27, 98), (26, 115), (31, 60)):
        q = min(n, m)
        U, _ = np.linalg.qr(rng.normal(size=(n, q)))
        V, _ = np.linalg.qr(rng.normal(size=(m, q)))
        A = (U * np.logspace(0, -8, q)) @ V.T
        d_true = np.linalg.svd(A, compute_uv=False)
        try:
            nt.nt_set_slices(2)
            nt.nt_set_schedule(C.c_double(1e-7), 3, 1)
            for k in (9, 11):
                r = host_svd(nt, A, k, tol=1e-4, block=8, seed=57)
                assert r["converged"]
                np.testing.assert_allclose(r["d"], d_true[:k], rtol=1e-6)
        finally:
            nt.nt_set_slices(0)
            nt.nt_set_schedule(C.c_double(0.0), 0, 0)
