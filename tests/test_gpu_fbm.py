"""FBM.code256 objects on the device: the byte-per-genotype matrix with any of
the reference's decode tables (R/bigSNP-class.R:7-13).

* CODE_012 / CODE_IMPUTE_PRED decode to genotype calls: 2-bit image, every snp_* function.
* CODE_DOSAGE decodes to a grid of dosages: byte image (one int8 grid index per genotype, exact integer
  sums) for snp_colstats, big_prodVec / big_cprodVec (bigstatsr, external: restated as the plain products
  on decoded values, oracle/bsn_oracle.c:orc_fbm_*), snp_PRS and big_randomSVD.
Parity bar: 1e-9 relative (north_star: 1e-6), indices sampled with replacement as the reference's tests do."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def _close(got, ref, tol=1e-9):
    scale = np.nanmax(np.abs(ref)) if np.size(ref) else 1.0
    np.testing.assert_allclose(got, ref, rtol=0, atol=tol * max(scale, 1e-300))


def _dosage_bytes(rng, n, m, with_na):
    """bytes the way bigsnpr writes imputed data: calls 0-2, imputed calls 4-6, dosages 7-207; 3 = missing"""
    a = rng.integers(7, 208, size=(n, m), dtype=np.int64)
    calls = rng.random((n, m)) < 0.3
    a[calls] = rng.integers(0, 3, size=int(calls.sum()))
    imp = rng.random((n, m)) < 0.1
    a[imp] = rng.integers(4, 7, size=int(imp.sum()))
    if with_na:
        a[rng.random((n, m)) < 0.02] = 3
        a[:, 5] = 3                      # an all-missing variant
    return a.astype(np.uint8)


@pytest.mark.parametrize("n,m", [(517, 300), (1003, 777), (64, 130)])
def test_dosage_colstats_and_products(ba, orc, n, m):
    rng = np.random.default_rng(n + m)
    raw = _dosage_bytes(rng, n, m, with_na=False)
    Go = orc.FBM256(raw, ba.CODE_DOSAGE)
    G = ba.FBM_code256(raw, ba.CODE_DOSAGE)
    assert G.bits == 8 and not G._has_na
    ir = rng.choice(n, n // 2, replace=True)
    ic = rng.choice(m, m // 2, replace=True)
    for rows, cols in ((None, None), (ir, ic)):
        st, ref = ba.snp_colstats(G, rows, cols), orc.snp_colstats(Go, rows, cols)
        _close(st["sumX"], ref["sumX"], 1e-12)
        _close(st["denoX"], ref["denoX"], 1e-9)
        nr, nc = (n if rows is None else rows.size), (m if cols is None else cols.size)
        x, y = rng.normal(size=nc), rng.normal(size=nr)
        rr = np.arange(n) if rows is None else rows
        cc = np.arange(m) if cols is None else cols
        _close(ba.big_prodVec(G, x, rows, cols), orc.fbm_prodVec(Go, x, rr, cc))
        _close(ba.big_cprodVec(G, y, rows, cols), orc.fbm_cprodVec(Go, y, rr, cc))
        # centre / scale: ((X - c) / s) x  ==  X (x / s) - sum(c x / s)
        c, s = rng.normal(size=nc), rng.uniform(0.5, 2.0, size=nc)
        _close(ba.big_prodVec(G, x, rows, cols, c, s), orc.fbm_prodVec(Go, x / s, rr, cc) - np.sum(c * x / s))
        _close(ba.big_cprodVec(G, y, rows, cols, c, s), (orc.fbm_cprodVec(Go, y, rr, cc) - c * y.sum()) / s)
    maf, af = ba.snp_MAF(G), orc.snp_colstats(Go)["sumX"] / (2.0 * n)
    _close(maf, np.minimum(af, 1 - af), 1e-12)


def test_dosage_with_missing_values(ba, orc):
    """the accessor of the reference has no missing-value handling (src/colstats.cpp:14-27): a column
    with a missing code has NA statistics, and the bigstatsr products would be NA: the GPU path says so"""
    rng = np.random.default_rng(3)
    n, m = 400, 90
    raw = _dosage_bytes(rng, n, m, with_na=True)
    Go, G = orc.FBM256(raw, ba.CODE_DOSAGE), ba.FBM_code256(raw, ba.CODE_DOSAGE)
    assert G.bits == 8 and G._has_na
    st, ref = ba.snp_colstats(G), orc.snp_colstats(Go)
    bad = (raw == 3).any(axis=0)
    assert np.isnan(st["sumX"][bad]).all() and np.isnan(ref["sumX"][bad]).all()
    _close(st["sumX"][~bad], ref["sumX"][~bad], 1e-12)
    _close(st["denoX"][~bad], ref["denoX"][~bad], 1e-9)
    with pytest.raises(ValueError, match="missing values"):
        ba.big_prodVec(G, np.ones(m))
    # ... for the SELECTION: a complete sub-matrix of an FBM with missing values elsewhere gives finite products,
    # as bigstatsr does (callers exclude such columns through ind.col)
    raw_c = raw.copy()
    raw_c[:, :10][raw_c[:, :10] == 3] = 107          # ten complete variants (a dosage of 1.00) next to the others
    Gc, Goc = ba.FBM_code256(raw_c, ba.CODE_DOSAGE), orc.FBM256(raw_c, ba.CODE_DOSAGE)
    good = np.arange(10)
    xg = rng.normal(size=good.size)
    assert Gc._has_na
    _close(ba.big_prodVec(Gc, xg, None, good), orc.fbm_prodVec(Goc, xg, None, good))
    _close(ba.big_cprodVec(Gc, np.ones(n), None, good), orc.fbm_cprodVec(Goc, np.ones(n), None, good))
    with pytest.raises(ValueError, match="missing values"):
        ba.big_prodVec(Gc, np.ones(11), None, np.arange(11))
    # the library itself treats a missing value as "contributes nothing" (mean-imputed after centring),
    # like bedAccScaled (src/bed-acc.h:98-111): checked on every variant that has a value at all
    ok = np.nonzero(~(raw == 3).all(axis=0))[0]
    from bigsnpr_amd.bed import bed_prodVec, bed_cprodVec
    x, y = rng.normal(size=ok.size), rng.normal(size=n)
    dec = ba.CODE_DOSAGE[raw[:, ok]]
    c = np.nanmean(dec, axis=0)
    A = np.where(np.isnan(dec), 0.0, dec - c)
    _close(bed_prodVec(G._bed, x, None, ok, c, None), A @ x)
    _close(bed_cprodVec(G._bed, y, None, ok, c, None), A.T @ y)


def test_dosage_prs_and_svd(ba, orc):
    rng = np.random.default_rng(11)
    n, m = 700, 1500
    # dosages around planted population structure so that the spectrum has a gap
    pop = rng.integers(0, 3, size=n)
    f = rng.uniform(0.1, 0.9, size=(3, m))
    dos = np.clip(np.round((2 * f[pop] + 0.15 * rng.normal(size=(n, m))) * 100), 0, 200).astype(np.int64)
    raw = (dos + 7).astype(np.uint8)
    Go, G = orc.FBM256(raw, ba.CODE_DOSAGE), ba.FBM_code256(raw, ba.CODE_DOSAGE)
    betas = rng.normal(size=m)
    same = rng.random(m) < 0.7
    lp = rng.uniform(0, 6, size=m)
    got = ba.snp_PRS(G, betas, same_keep=same, lpS_keep=lp, thr_list=[1, 3, 5])
    ref = orc.snp_PRS(Go, betas, same_keep=same, lpS_keep=lp, thr_list=[1, 3, 5])
    _close(got, ref)
    res = ba.big_randomSVD(G, ba.snp_scaleBinom(), k=5, tol=1e-10, slices=7)
    dec = ba.CODE_DOSAGE[raw]
    af = dec.sum(0) / (2.0 * n)
    A = (dec - 2 * af) / np.sqrt(2 * af * (1 - af))
    d = np.linalg.svd(A, compute_uv=False)[:5]
    np.testing.assert_allclose(res["d"], d, rtol=1e-9)
    # default settings (16-bit panels): singular values within the north_star bar
    res = ba.big_randomSVD(G, ba.snp_scaleBinom(), k=5)
    np.testing.assert_allclose(res["d"], d, rtol=1e-6)


def test_code_tables(ba, orc, golden_dir, example_bed):
    """CODE_IMPUTE_PRED (bytes 4-6 are imputed calls) shares the 2-bit image with CODE_012"""
    G012 = orc.fbm_from_bed(example_bed)
    raw = G012.bytes.copy()
    flip = np.random.default_rng(0).random(raw.shape) < 0.3
    raw[flip] += 4                                    # 0,1,2 -> 4,5,6: same decoded values
    G = ba.FBM_code256(raw, ba.CODE_IMPUTE_PRED)
    assert G.bits == 2
    ref = orc.snp_colstats(G012)
    st = ba.snp_colstats(G)
    np.testing.assert_array_equal(st["sumX"], ref["sumX"])
    np.testing.assert_array_equal(st["denoX"], ref["denoX"])
    # what the byte image cannot do says so
    Gd = ba.FBM_code256((raw[:60, :80] % 3 + 7).astype(np.uint8), ba.CODE_DOSAGE)
    with pytest.raises(ba.BsnError, match="2-bit genotype image"):
        ba.bed_counts(Gd._bed)


def test_any_decode_table_is_served(ba, orc):
    """SubBMCode256Acc decodes through ANY 256 doubles (src/colstats.cpp:13-14, R/bigSNP-class.R:7-13).  A table that
    is neither calls nor a grid has no exact integer image: it takes the fp64 look-up kernels — snp_colstats,
    big_prodVec, big_cprodVec within 1e-9 of the oracle, rows / columns with replacement, centre / scale — and
    everything else says by name what it cannot do."""
    rng = np.random.default_rng(31)
    n, m = 700, 450
    code = rng.normal(size=256) * np.exp(rng.normal(size=256))          # an arbitrary table, no missing code
    raw = rng.integers(0, 256, size=(n, m)).astype(np.uint8)
    Go, G = orc.FBM256(raw, code), ba.FBM_code256(raw, code)
    assert G.bits == 8 and not G._has_na
    st, ref = ba.snp_colstats(G), orc.snp_colstats(Go)
    np.testing.assert_allclose(st["sumX"], ref["sumX"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(st["denoX"], ref["denoX"], rtol=1e-9)
    ir, ic = rng.choice(n, 500, replace=True), rng.choice(m, 600, replace=True)
    st, ref = ba.snp_colstats(G, ir, ic), orc.snp_colstats(Go, ir, ic)
    np.testing.assert_allclose(st["sumX"], ref["sumX"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(st["denoX"], ref["denoX"], rtol=1e-9)
    x, y = rng.normal(size=ic.size), rng.normal(size=ir.size)
    _close(ba.big_prodVec(G, x, ir, ic), orc.fbm_prodVec(Go, x, ir, ic))
    _close(ba.big_cprodVec(G, y, ir, ic), orc.fbm_cprodVec(Go, y, ir, ic))
    xf, yf = rng.normal(size=m), rng.normal(size=n)
    _close(ba.big_prodVec(G, xf), orc.fbm_prodVec(Go, xf))
    _close(ba.big_cprodVec(G, yf), orc.fbm_cprodVec(Go, yf))
    c, s_ = rng.normal(size=ic.size), rng.uniform(0.5, 2, size=ic.size)
    _close(ba.big_prodVec(G, x, ir, ic, c, s_), orc.fbm_prodVec(Go, x / s_, ir, ic) - np.sum(c * x / s_))
    _close(ba.big_cprodVec(G, y, ir, ic, c, s_), (orc.fbm_cprodVec(Go, y, ir, ic) - c * y.sum()) / s_)
    # run-to-run identical (fixed summation order)
    np.testing.assert_array_equal(ba.big_prodVec(G, xf), ba.big_prodVec(G, xf))
    # a table with a missing code: NA_real poisons the sums of the columns that hold it, like the reference's accessor
    code2 = code.copy()
    code2[200:] = np.nan
    G2, Go2 = ba.FBM_code256(raw, code2), orc.FBM256(raw, code2)
    st, ref = ba.snp_colstats(G2), orc.snp_colstats(Go2)
    assert np.array_equal(np.isnan(st["sumX"]), np.isnan(ref["sumX"])) and np.isnan(ref["sumX"]).any()
    ok = ~np.isnan(ref["sumX"])
    np.testing.assert_allclose(st["sumX"][ok], ref["sumX"][ok], rtol=1e-10, atol=1e-9)
    # what the look-up image cannot do says so
    for fn in (lambda: ba.snp_cor(G, size=10), lambda: ba.big_randomSVD(G, k=3),
               lambda: ba.snp_ld_scores(G, size=10)):
        with pytest.raises(ba.BsnError, match="neither genotype calls"):
            fn()


def test_fbm_products_on_calls_with_replacement(ba, orc, example_bed):
    """a9 on the CODE_012 image: big_prodVec / big_cprodVec == the plain products on decoded values
    (oracle), rows and columns sampled WITH replacement, with and without centre / scale"""
    Go = orc.fbm_from_bed(example_bed)
    G = ba.FBM_code256(Go.bytes)
    rng = np.random.default_rng(8)
    n, m = Go.n, Go.m
    ir, ic = rng.choice(n, 300, replace=True), rng.choice(m, 2000, replace=True)
    x, y = rng.normal(size=ic.size), rng.normal(size=ir.size)
    _close(ba.big_prodVec(G, x, ir, ic), orc.fbm_prodVec(Go, x, ir, ic))
    _close(ba.big_cprodVec(G, y, ir, ic), orc.fbm_cprodVec(Go, y, ir, ic))
    c, s = rng.normal(size=ic.size), rng.uniform(0.5, 2, size=ic.size)
    _close(ba.big_prodVec(G, x, ir, ic, c, s), orc.fbm_prodVec(Go, x / s, ir, ic) - np.sum(c * x / s))
    _close(ba.big_cprodVec(G, y, ir, ic, c, s), (orc.fbm_cprodVec(Go, y, ir, ic) - c * y.sum()) / s)
    _close(ba.big_cprodVec(G, np.ones(n)), orc.snp_colstats(Go)["sumX"], 1e-12)


def test_c2_fbm_ingest_at_full_size(ba):
    """config C2: a 50 000 x 200 000 FBM.code256 (10 GB at one byte per genotype) really goes through
    bsn_fbm_open — pinned double-buffered upload, repack to the 2-bit image on the device — and the image
    equals the one the generator writes directly (the bytes are its decoded genotypes)."""
    n, m = 50000, 200000
    ref = ba.bed.synthetic(n, m, seed=77)
    chunk = 20000
    raw = np.empty((n, m), dtype=np.uint8, order="F")
    for j in range(0, m, chunk):
        raw[:, j:j + chunk] = ba.bed_to_bytes(ref, None, np.arange(j, min(m, j + chunk)))
    import time
    t0 = time.perf_counter()
    G = ba.FBM_code256(raw)
    dt = time.perf_counter() - t0
    assert G.bits == 2
    np.testing.assert_array_equal(ba.bed_counts(G._bed), ba.bed_counts(ref))
    rng = np.random.default_rng(0)
    y = rng.normal(size=n)
    np.testing.assert_array_equal(ba.bed_cprodVec(G._bed, y), ba.bed_cprodVec(ref, y))
    cols = rng.choice(m, 300, replace=False)
    np.testing.assert_array_equal(ba.read_bed(G._bed, np.arange(n), cols), ba.read_bed(ref, np.arange(n), cols))
    print("C2 FBM ingest: %.2f s for %.1f GB (%.1f GB/s)" % (dt, n * m / 1e9, n * m / 1e9 / dt))


def _dosage_panel(rng, n, m):
    """dosages with local correlation (an AR(1) latent along the variants) so that windows hold real LD"""
    z = rng.normal(size=(n, m))
    for j in range(1, m):
        z[:, j] = 0.8 * z[:, j - 1] + 0.6 * z[:, j]
    f = rng.uniform(0.1, 0.9, size=m)
    from scipy.stats import norm
    dos = np.clip(np.round(200 * norm.cdf(z + norm.ppf(f))), 0, 200).astype(np.int64)
    return (dos + 7).astype(np.uint8)


def test_dosage_ld_matches_oracle(ba, orc):
    """snp_cor / snp_ld_scores / snp_clumping on a CODE_DOSAGE FBM without missing values: the byte image's
    cross products are exact integers and r is affine-invariant, so the oracle's corMat (kind = 1: the FBM
    accessor with the dosage table, src/corr.cpp:113-118) is met to 1e-9 with the identical sparsity
    pattern, the LD scores to 1e-9, and the clumping keeps the identical variants."""
    rng = np.random.default_rng(21)
    n, m = 900, 700
    raw = _dosage_panel(rng, n, m)
    Go, G = orc.FBM256(raw, ba.CODE_DOSAGE), ba.FBM_code256(raw, ba.CODE_DOSAGE)
    assert G.bits == 8
    pos = np.cumsum(rng.integers(1, 3000, size=m)).astype(np.float64)
    ir = np.sort(rng.choice(n, 600, replace=False))
    for rows, kw in ((None, dict(size=40, infos_pos=pos)), (ir, dict(size=25, infos_pos=pos, alpha=0.05)),
                     (None, dict(size=30, thr_r2=0.1))):
        got = ba.snp_cor(G, ind_row=rows, **kw)
        ri, rp, rx = orc.snp_cor(Go, ind_row=rows, **kw)           # CSC slots i, p, x (R/corr.R:43-47)
        np.testing.assert_array_equal(got.p, rp)
        np.testing.assert_array_equal(got.i, ri)
        np.testing.assert_allclose(got.x, rx, rtol=0, atol=1e-9)
    np.testing.assert_allclose(ba.snp_ld_scores(G, size=40, infos_pos=pos),
                               orc.ld_scores(Go, size=40, infos_pos=pos), rtol=1e-9)
    chrom = np.repeat([1, 2], [400, 300])
    for kw in (dict(thr_r2=0.2, infos_pos=pos), dict(thr_r2=0.05, size=20), dict(thr_r2=0.3, infos_pos=pos, ind_row=ir)):
        np.testing.assert_array_equal(ba.snp_clumping(G, chrom, **kw), orc.snp_clumping(Go, chrom, **kw))


def test_dosage_ld_with_missing_values_matches_oracle(ba, orc):
    """snp_cor / snp_ld_scores on a CODE_DOSAGE FBM WITH missing values: the reference recodes NA to 3 and runs
    the pairwise-complete loop of corMat0 on any code256 (src/corr.cpp:113-118, src/ld-scores.cpp:92-97); the
    byte image does it with eight exact int8 products (the marker -128 is the mask plane, k^2 two digit planes)
    — identical sparsity pattern, x to 1e-9, LD scores to 1e-9.  snp_clumping: a pair with a missing dosage has
    r2 = NA in the reference (clumping_chr has no missing-value handling) and never prunes."""
    rng = np.random.default_rng(22)
    n, m = 1100, 520
    raw = _dosage_panel(rng, n, m)
    miss = rng.random((n, m)) < 0.03
    miss[:, 11] = rng.random(n) < 0.6          # a variant that is mostly missing
    miss[:, 40] = False                         # and complete ones
    miss[3, :] = True                           # a sample without data
    raw[miss] = 3
    Go, G = orc.FBM256(raw, ba.CODE_DOSAGE), ba.FBM_code256(raw, ba.CODE_DOSAGE)
    assert G.bits == 8 and G._has_na
    pos = np.cumsum(rng.integers(1, 3000, size=m)).astype(np.float64)
    ir = np.sort(rng.choice(n, 700, replace=False))
    for rows, kw in ((None, dict(size=40, infos_pos=pos)), (ir, dict(size=25, infos_pos=pos, alpha=0.05)),
                     (None, dict(size=30, thr_r2=0.1))):
        got = ba.snp_cor(G, ind_row=rows, **kw)
        ri, rp, rx = orc.snp_cor(Go, ind_row=rows, **kw)
        np.testing.assert_array_equal(got.p, rp)
        np.testing.assert_array_equal(got.i, ri)
        np.testing.assert_allclose(got.x, rx, rtol=0, atol=1e-9, equal_nan=True)
    assert "8 products" in ba.ld.last_stats()["kernel"]
    np.testing.assert_allclose(ba.snp_ld_scores(G, size=40, infos_pos=pos),
                               orc.ld_scores(Go, size=40, infos_pos=pos), rtol=1e-9)
    np.testing.assert_allclose(ba.snp_ld_scores(G, ind_row=ir, size=25, infos_pos=pos),
                               orc.ld_scores(Go, ind_row=ir, size=25, infos_pos=pos), rtol=1e-9)


def test_default_scaling_of_big_randomsvd_rides_along_the_first_pass(ba, orc, example_bed):
    """big_randomSVD(G) with the default snp_scaleBinom() == the explicit scaling function, field by field
    (the default is evaluated inside the solve: R/binom-scaling.R:62-77 on complete data is bed_scaleBinom's
    formula with nb_nona = n)"""
    G = ba.FBM_code256(orc.fbm_from_bed(example_bed).bytes)
    a = ba.big_randomSVD(G, k=6, tol=1e-10, slices=7)
    b = ba.big_randomSVD(G, ba.snp_scaleBinom(), k=6, tol=1e-10, slices=7)
    assert a["fused_stats"] and not b["fused_stats"]
    np.testing.assert_array_equal(a["center"], b["center"])
    np.testing.assert_array_equal(a["scale"], b["scale"])
    np.testing.assert_allclose(a["d"], b["d"], rtol=1e-12)
