"""Seeded random shapes through the C ABI against the oracle: the fixed-size tests pin the reference's own cases, these
sweep what lies between them — ragged sizes around every tile boundary (4 samples per byte, 16 / 64 / 128-row MFMA tiles,
256-variant panels), index lists with repeats and in any order, missing-value rates from none to heavy, every block
size and digit count the solver accepts.  Same tolerances as the fixed tests: integer and index outputs bit-exact,
products 1e-9, correlations 1e-12, singular values 1e-6."""
import os

import numpy as np
import pytest

# BSN_TEST_SEED_OFFSET=<k> moves every case to another draw (exploration runs; the committed suite runs offset 0)
_OFF = 100000 * int(os.environ.get("BSN_TEST_SEED_OFFSET", "0"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def _pair(ba, orc, n, m, seed, na16):
    return ba.bed.synthetic(n, m, seed=seed, na16=na16), orc.fake_bed(n, m, seed=seed, na16=na16)


def _indices(rng, limit, style):
    if style == 0:
        return None
    if style == 1:                                    # sorted subset
        return np.sort(rng.choice(limit, max(1, int(limit * rng.uniform(0.2, 0.9))), replace=False))
    if style == 2:                                    # any order, with repeats
        return rng.integers(0, limit, size=max(1, int(limit * rng.uniform(0.3, 1.4))))
    return np.arange(limit)[::-1].copy()              # reversed


_SIZES = [1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 127, 129, 255, 257, 511, 1023, 1025, 2049]


@pytest.mark.parametrize("case", range(24))
def test_products_and_statistics(ba, orc, case):
    rng = np.random.default_rng(_OFF + 1000 + case)
    n = int(rng.choice(_SIZES)) + int(rng.integers(0, 3))
    m = int(rng.choice(_SIZES)) + int(rng.integers(0, 3))
    na16 = int(rng.choice([0, 0, 655, 6000, 30000]))
    gb, ob = _pair(ba, orc, n, m, 77 + case, na16)
    ir = _indices(rng, n, case % 4)
    ic = _indices(rng, m, (case // 4) % 4)
    nr = n if ir is None else ir.size
    nc = m if ic is None else ic.size
    # counts and column statistics: bit-exact
    np.testing.assert_array_equal(ba.bed_counts(gb, ir, ic), orc.bed_col_counts(ob, ir, ic))
    st, st_o = ba.bed_colstats(gb, ir, ic), orc.bed_colstats(ob, ir, ic)
    for f in ("sumX", "denoX", "nb_nona_col"):
        np.testing.assert_array_equal(st[f], st_o[f])
    np.testing.assert_array_equal(ba.read_bed(gb, np.arange(n) if ir is None else ir,
                                              np.arange(m) if ic is None else ic),
                                  orc.read_bed(ob, ir, ic))
    center = rng.normal(size=nc) if case % 3 else None
    scale = rng.uniform(0.5, 2.0, size=nc) if case % 3 else None
    x = rng.normal(size=nc) * 10.0 ** rng.integers(-6, 6)
    y = rng.normal(size=nr) * 10.0 ** rng.integers(-6, 6)
    got, want = ba.bed_prodVec(gb, x, ir, ic, center, scale), orc.bed_prodVec(ob, x, ir, ic, center, scale)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()))
    got, want = ba.bed_cprodVec(gb, y, ir, ic, center, scale), orc.bed_cprodVec(ob, y, ir, ic, center, scale)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()))


@pytest.mark.parametrize("case", range(20))
def test_partial_svd(ba, orc, case):
    rng = np.random.default_rng(_OFF + 2000 + case)
    n = int(rng.integers(30, 2200))
    m = int(rng.integers(40, 4200))
    na16 = int(rng.choice([0, 655, 6000]))
    gb, ob = _pair(ba, orc, n, m, 177 + case, na16)
    ir = _indices(rng, n, 1) if case % 2 else None
    nr = n if ir is None else ir.size
    sc = orc.bed_scaleBinom(ob, ir, None)
    ic = np.nonzero(sc["scale"] > 0)[0]
    if case % 3 == 0 and ic.size > 60:
        ic = np.sort(rng.choice(ic, int(ic.size * 0.7), replace=False))
    k = int(rng.integers(1, max(2, min(25, min(nr, ic.size) // 3))))
    block = int(rng.choice([0, 0, 1, 3, 4, 8, 16]))
    slices = int(rng.choice([0, 0, 2, 3, 4, 7]))
    if block * max(slices, 2) > 32:                   # the library's limit on digit columns per pass
        block = 4
    ref = orc.dense_svd(ob, ir, ic, k=k)
    res = ba.bed_randomSVD(gb, ind_row=ir, ind_col=ic, k=k, block=block, slices=slices, seed=case + 1)
    assert res["converged"], (n, m, k, block, slices)
    if block == 1 and k > 4:
        # one vector per pass IS single-vector Lanczos (what RSpectra runs): in the flat spectrum of a small random
        # matrix it can stop with every one of its k Ritz pairs converged and one value of a cluster not found yet —
        # offset 4000, case 14: 344 x 3360, k = 16, the 17th value (73.2257) returned in place of the 16th (73.2489).
        # What must hold: every returned value IS a singular value, and the leading ones are the leading ones.
        more = orc.dense_svd(ob, ir, ic, k=min(k + 4, nr, ic.size))["d"]
        assert np.abs(res["d"][:, None] / more[None, :] - 1).min(axis=1).max() < 1e-6
        np.testing.assert_allclose(res["d"][:k - 4], ref["d"][:k - 4], rtol=1e-6)
    else:
        np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-6)
    np.testing.assert_array_equal(res["center"], ref["center"])
    np.testing.assert_array_equal(res["scale"], ref["scale"])
    # the triplets are consistent with the matrix itself: A~ v = u d through the oracle's products
    j = int(rng.integers(0, k))
    av = orc.bed_prodVec(ob, res["v"][:, j].copy(), ir, ic, ref["center"], ref["scale"])
    assert np.abs(av - res["u"][:, j] * res["d"][j]).max() <= 2e-4 * res["d"][0]
    np.testing.assert_allclose(res["u"].T @ res["u"], np.eye(k), atol=1e-6)
    np.testing.assert_allclose(res["v"].T @ res["v"], np.eye(k), atol=1e-6)


def _same_cor(res, ref, tol=1e-12):
    i, p, x = ref
    np.testing.assert_array_equal(res.p, p)
    np.testing.assert_array_equal(res.i, i)
    assert np.all(np.isnan(res.x) == np.isnan(x))
    ok = ~np.isnan(x)
    np.testing.assert_allclose(res.x[ok], x[ok], rtol=0, atol=tol)


@pytest.mark.filterwarnings("ignore:.*NA or NaN values.*")
@pytest.mark.parametrize("case", range(16))
def test_correlations_scores_clumping(ba, orc, case):
    rng = np.random.default_rng(_OFF + 3000 + case)
    n = int(rng.integers(20, 1500))
    m = int(rng.integers(2, 900))
    na16 = int(rng.choice([0, 655, 12000]))
    gb, ob = _pair(ba, orc, n, m, 277 + case, na16)
    ir = _indices(rng, n, case % 3)
    ic = _indices(rng, m, 1) if case % 2 else None
    mc = m if ic is None else ic.size
    size = float(rng.choice([1, 7, 33, 150, 1e4]))
    kw = dict(size=size, alpha=float(rng.choice([1.0, 0.2])), thr_r2=float(rng.choice([0.0, 0.1])),
              fill_diag=bool(case % 2))
    pos = None
    if case % 4 == 3:
        pos = np.cumsum(rng.integers(1, 3000, size=mc)).astype(np.float64)
        kw["size"] = float(rng.choice([2.0, 20.0]))
    _same_cor(ba.bed_cor(gb, ir, ic, infos_pos=pos, **kw), orc.snp_cor(ob, ir, ic, infos_pos=pos, ncores=8, **kw))
    got = ba.bed_ld_scores(gb, ir, ic, size=kw["size"], infos_pos=pos)
    want = orc.ld_scores(ob, ir, ic, size=kw["size"], infos_pos=pos)
    assert np.all(np.isnan(got) == np.isnan(want))
    np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-12)
    # clumping over the whole matrix in two "chromosomes": kept indices identical
    chrom = np.r_[np.ones(m // 2, dtype=np.int64), np.full(m - m // 2, 2, dtype=np.int64)]
    bp = np.cumsum(rng.integers(1, 5000, size=m))
    S = rng.uniform(size=m) if case % 2 else None
    # (thresholds that no r2 can hit: bed_clumping_chr sums scaled products in fp64, src/clumping-bed.cpp:62-76, while the
    # GPU works from exact integer sums — with a few dozen samples r2 = 1 / 5 EXACTLY happens, the reference then sees
    # 0.20000000000000004 > 0.2 and the two sides of the tie differ legitimately; tests/test_gpu_ld.py treats that case)
    thr = float(rng.choice([0.0517, 0.2013, 0.4989]))
    win = float(rng.choice([50, 500]))
    with np.errstate(all="ignore"):
        want = orc.bed_clumping(ob, chrom, bp, ind_row=ir, S=S, thr_r2=thr, size=win)
        got = ba.bed_clumping(gb, ind_row=ir, S=S, thr_r2=thr, size=win, infos_chr=chrom, infos_pos=bp)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("case", range(10))
def test_kinship_projection_regression_prs(ba, orc, case):
    rng = np.random.default_rng(_OFF + 4000 + case)
    n = int(rng.choice([9, 31, 130, 257, 640, 1001]))
    m = int(rng.integers(20, 1800))
    na16 = int(rng.choice([0, 655, 9000]))
    gb, ob = _pair(ba, orc, n, m, 377 + case, na16)
    ir = _indices(rng, n, (case % 3))
    nr = n if ir is None else ir.size
    sc = orc.bed_scaleBinom(ob, ir, None)
    ic = np.nonzero(sc["scale"] > 0)[0]
    if case % 2:
        ic = ic[rng.permutation(ic.size)[:max(1, ic.size // 2)]]
    # bed_tcrossprodSelf
    fs = lambda obj, ind_row, ind_col, ncores=1: dict(center=sc["center"][ind_col], scale=sc["scale"][ind_col])
    K, _ = ba.bed_tcrossprodSelf(gb, fun_scaling=fs, ind_row=ir, ind_col=ic)
    A = orc.read_bed_scaled(ob, ob.rows() if ir is None else ir, ic, sc["center"][ic], sc["scale"][ic])
    Kref = A @ A.T
    np.testing.assert_allclose(K, Kref, rtol=0, atol=1e-10 * max(1.0, np.abs(Kref).max()))
    np.testing.assert_array_equal(K, K.T)
    # prod_and_rowSumsSq
    kk = int(rng.integers(1, 12))
    V = rng.normal(size=(ic.size, kk))
    XV, rs = ba.prod_and_rowSumsSq(gb, np.arange(n) if ir is None else ir, ic, sc["center"][ic], sc["scale"][ic], V)
    np.testing.assert_allclose(XV, A @ V, rtol=0, atol=1e-9 * max(1.0, np.abs(A @ V).max()))
    np.testing.assert_allclose(rs, (A * A).sum(1), rtol=1e-9, atol=1e-9)
    # multLinReg
    if nr > 8:
        ku = int(rng.integers(1, min(6, nr - 3)))
        U = rng.normal(size=(nr, ku))
        ref = orc.multLinReg(ob, ir, ic, U)
        got = ba.multLinReg(gb, ir, ic, U)
        ok = ~np.isnan(ref)
        assert np.array_equal(np.isnan(got), ~ok)
        np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-7, atol=1e-7)
    # snp_PRS on the FBM image of the same calls (complete data only, as the reference requires)
    if na16 == 0:
        Go = orc.fbm_from_bed(ob)
        G = ba.FBM_code256(Go.bytes)
        keep = np.sort(rng.choice(m, max(1, m // 2), replace=False))
        betas = rng.normal(size=keep.size)
        lp = rng.uniform(0, 6, size=keep.size)
        same = rng.random(keep.size) < 0.7
        thr = np.sort(rng.uniform(0, 6, size=int(rng.integers(1, 9))))
        got = ba.snp_PRS(G, betas, ir, keep, same, lp, thr)
        ref = orc.snp_PRS(Go, betas, ir, keep, same, lp, thr)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))


@pytest.mark.filterwarnings("ignore:.*NA or NaN values.*")
@pytest.mark.parametrize("case", range(10))
def test_dosage_matrices(ba, orc, case):
    rng = np.random.default_rng(_OFF + 5000 + case)
    n = int(rng.integers(10, 1300))
    m = int(rng.integers(3, 700))
    f = rng.uniform(0.05, 0.95, size=m)
    dos = np.clip(np.round((2 * f + 0.4 * rng.normal(size=(n, m))) * 100), 0, 200).astype(np.int64)
    raw = (dos + 7).astype(np.uint8)
    with_na = case % 2 == 1
    if with_na:
        raw[rng.random(raw.shape) < rng.choice([0.002, 0.05])] = 3
    Go, G = orc.FBM256(raw, ba.CODE_DOSAGE), ba.FBM_code256(raw, ba.CODE_DOSAGE)
    ir = _indices(rng, n, case % 3)
    ic = _indices(rng, m, 1) if case % 4 >= 2 else None
    st, ref = ba.snp_colstats(G, ir, ic), orc.snp_colstats(Go, ir, ic)
    assert np.array_equal(np.isnan(st["sumX"]), np.isnan(ref["sumX"]))
    ok = ~np.isnan(ref["sumX"])
    np.testing.assert_allclose(st["sumX"][ok], ref["sumX"][ok], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(st["denoX"][ok], ref["denoX"][ok], rtol=1e-9, atol=1e-9)
    if not with_na:
        nr = n if ir is None else ir.size
        nc = m if ic is None else ic.size
        x, y = rng.normal(size=nc), rng.normal(size=nr)
        want = orc.fbm_prodVec(Go, x, ir, ic)
        np.testing.assert_allclose(ba.big_prodVec(G, x, ir, ic), want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()))
        want = orc.fbm_cprodVec(Go, y, ir, ic)
        np.testing.assert_allclose(ba.big_cprodVec(G, y, ir, ic), want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()))
    # windowed correlations on dosages, with and without missing values (src/corr.cpp:113-118)
    kw = dict(size=float(rng.choice([3, 40, 1e4])), thr_r2=float(rng.choice([0.0, 0.05])), fill_diag=bool(case % 2))
    if (n if ir is None else ir.size) < 8:
        # A dosage column that is CONSTANT over the selected samples has r = 0 / 0.  The library's sums are exact
        # integers: NaN, kept.  The reference's are sums of decimals (0.57 + 0.57): its "zero" variance is 1e-16 and
        # r is rounding noise that the threshold drops (offset 11000, case 1: 2 samples, 20 such columns of 405, 2 355
        # entries against 2 192).  With a handful of samples such columns are common, so the comparison starts at 8.
        return
    _same_cor(ba.snp_cor(G, ir, ic, **kw), orc.snp_cor(Go, ir, ic, **kw), tol=1e-9)
    got = ba.snp_ld_scores(G, ir, ic, size=kw["size"])
    want = orc.ld_scores(Go, ir, ic, size=kw["size"])
    assert np.all(np.isnan(got) == np.isnan(want))
    np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-9)


@pytest.mark.parametrize("case", range(16))
def test_partial_svd_of_tiny_matrices(ba, orc, case):
    """dimensions below one tile, k up to the smaller dimension minus one: the Krylov space is exhausted on the way
    (the centred matrix has rank <= n - 1), singular values beyond the rank come out as (numerical) zeros"""
    rng = np.random.default_rng(_OFF + 6000 + case)
    n = int(rng.integers(5, 48))
    m = int(rng.integers(6, 70))
    gb, ob = _pair(ba, orc, n, m, 477 + case, int(rng.choice([0, 3000])))
    sc = orc.bed_scaleBinom(ob)
    ic = np.nonzero(sc["scale"] > 0)[0]
    if ic.size < 3:
        pytest.skip("degenerate draw")
    kmax = min(n, ic.size) - 1
    k = int(rng.integers(1, kmax + 1))
    block = int(rng.choice([0, 1, 2, 8, 16]))
    ref = orc.dense_svd(ob, None, ic, k=k)
    res = ba.bed_randomSVD(gb, ind_col=ic, k=k, block=block, seed=case + 1)
    # an exhausted space whose last directions carry the noise of the 16-bit products is solved again on 56-bit
    # products (svd_driver.hpp): converged, and the values of a dense solver
    assert res["converged"]
    np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-6, atol=1e-6 * ref["d"][0])
    big = ref["d"] > 1e-3 * ref["d"][0]
    np.testing.assert_allclose(res["d"][big], ref["d"][big], rtol=1e-6)
    kk = int(big.sum())
    np.testing.assert_allclose((res["u"].T @ res["u"])[:kk, :kk], np.eye(kk), atol=1e-6)


@pytest.mark.parametrize("n,m,k,block", [(1500, 2500, 40, 0), (1200, 5000, 80, 8), (900, 1400, 120, 16), (2500, 1800, 64, 4)])
def test_many_triplets(ba, orc, n, m, k, block):
    """k well beyond one block: many block steps, thick restarts when the basis fills up, same singular values"""
    gb, ob = _pair(ba, orc, n, m, 577 + k, 655)
    sc = orc.bed_scaleBinom(ob)
    ic = np.nonzero(sc["scale"] > 0)[0]
    ref = orc.dense_svd(ob, None, ic, k=k)
    res = ba.bed_randomSVD(gb, ind_col=ic, k=k, block=block)
    assert res["converged"]
    np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-6)
    np.testing.assert_allclose(res["u"].T @ res["u"], np.eye(k), atol=1e-6)
    j = k - 1
    av = orc.bed_prodVec(ob, res["v"][:, j].copy(), None, ic, ref["center"], ref["scale"])
    assert np.abs(av - res["u"][:, j] * res["d"][j]).max() <= 2e-4 * res["d"][0]


@pytest.mark.parametrize("case", range(8))
def test_grid_clumping_and_scores(ba, orc, case):
    """snp_grid_clumping / snp_grid_PRS (R/SCT.R) on random shapes: kept indices identical, scores 1e-9"""
    rng = np.random.default_rng(_OFF + 7000 + case)
    n = int(rng.integers(30, 900))
    m = int(rng.integers(20, 700))
    gb, ob = _pair(ba, orc, n, m, 677 + case, 0)
    Go = orc.fbm_from_bed(ob)
    G = ba.FBM_code256(Go.bytes)
    nchr = int(rng.integers(1, 4))
    chrom = np.sort(rng.integers(1, nchr + 1, size=m))
    pos = np.zeros(m, dtype=np.int64)
    for c in np.unique(chrom):
        idx = np.nonzero(chrom == c)[0]
        pos[idx] = np.cumsum(rng.integers(1, 4000, size=idx.size))
    lpval = -np.log10(rng.uniform(size=m))
    ir = _indices(rng, n, case % 2)                      # all rows / a sorted subset
    kw = dict(grid_thr_r2=tuple(sorted(rng.choice([0.0513, 0.1007, 0.2011, 0.5003, 0.8009], size=2, replace=False))),
              grid_base_size=tuple(sorted(rng.choice([37, 50, 100, 211], size=2, replace=False).tolist())))
    if case % 3 == 0:
        kw["exclude"] = rng.choice(m, size=max(1, m // 10), replace=False)
    with np.errstate(all="ignore"):
        got = ba.snp_grid_clumping(G, chrom, pos, lpval, ind_row=ir, **kw)
        want, grid, _ = orc.snp_grid_clumping(Go, chrom, pos, lpval, ind_row=ir, **kw)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert len(a) == len(b)
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)
    betas = rng.normal(0, 0.1, m)
    thr = orc.seq_log(0.1, 0.9999 * lpval.max(), 6)
    S = ba.snp_grid_PRS(G, got, betas, lpval, ind_row=ir, grid_lpS_thr=thr, type="double")
    Sref = orc.snp_grid_PRS(Go, want, betas, lpval, thr, ind_row=ir)
    np.testing.assert_allclose(np.asarray(S), Sref, rtol=0, atol=1e-9 * max(1.0, np.abs(Sref).max()))


@pytest.mark.filterwarnings("ignore:.*NA or NaN values.*")
@pytest.mark.parametrize("case", range(6))
def test_correlations_on_longer_chromosomes(ba, orc, case):
    """several thousand variants: many 128 x 32 blocks per launch, windows that cut blocks anywhere, the lazy clumping
    path, rows as a subset / with repeats"""
    rng = np.random.default_rng(_OFF + 8000 + case)
    n = int(rng.integers(150, 1600))
    m = int(rng.integers(2500, 7000))
    gb, ob = _pair(ba, orc, n, m, 777 + case, int(rng.choice([0, 655, 9000])))
    ir = _indices(rng, n, case % 3)
    ic = _indices(rng, m, 1) if case % 2 else None
    mc = m if ic is None else ic.size
    pos = np.cumsum(rng.integers(1, 2000, size=mc)).astype(np.float64)
    size = float(rng.choice([15.0, 90.0, 400.0]))           # kb: windows of a few to a few hundred variants
    _same_cor(ba.bed_cor(gb, ir, ic, size=size, infos_pos=pos, thr_r2=0.0507), orc.snp_cor(ob, ir, ic, size=size, infos_pos=pos, thr_r2=0.0507, ncores=8))
    got = ba.bed_ld_scores(gb, ir, ic, size=size, infos_pos=pos)
    want = orc.ld_scores(ob, ir, ic, size=size, infos_pos=pos)
    assert np.all(np.isnan(got) == np.isnan(want))
    np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-11)
    chrom = np.ones(m, dtype=np.int64)
    bp = np.cumsum(rng.integers(1, 3000, size=m))
    win = float(rng.choice([100, 1000]))
    with np.errstate(all="ignore"):
        want = orc.bed_clumping(ob, chrom, bp, ind_row=ir, thr_r2=0.2013, size=win)
        got = ba.bed_clumping(gb, ind_row=ir, thr_r2=0.2013, size=win, infos_chr=chrom, infos_pos=bp)
    np.testing.assert_array_equal(got, want)
