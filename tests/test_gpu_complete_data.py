"""Operators over variants that are known to have no missing genotype skip the missing-value
plane (one look-up set and one MFMA per 1 024 genotypes instead of two).  Knowledge comes from
any full-sample count (bed_counts / bed_colstats / bed_scaleBinom), from FBM creation, or from
the generator.  Results must equal the oracle and be BIT-identical to the general kernels
(integer sums: the skipped plane contributes exact zeros)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def _both_paths(fn):
    os.environ.pop("BSN_FORCE_NA_PLANE", None)
    fast = fn()
    os.environ["BSN_FORCE_NA_PLANE"] = "1"
    try:
        general = fn()
    finally:
        os.environ.pop("BSN_FORCE_NA_PLANE", None)
    return fast, general


def test_products_on_complete_synthetic(ba, orc):
    n, m = 2999, 1500
    ob = orc.fake_bed(n, m, na16=0)
    gb = ba.bed.synthetic(n, m, na16=0)
    assert orc.bed_col_counts(ob)[3].sum() == 0
    sc = orc.bed_scaleBinom(ob)
    rng = np.random.default_rng(3)
    x, y = rng.normal(size=m), rng.normal(size=n)
    ir = rng.choice(n, 1000, replace=True)
    ic = rng.choice(m, 700, replace=True)
    for rows, cols in ((None, None), (ir, ic)):
        c = sc["center"] if cols is None else sc["center"][cols]
        s = sc["scale"] if cols is None else sc["scale"][cols]
        xx = x if cols is None else x[cols]
        yy = y if rows is None else y[rows]
        f, g = _both_paths(lambda: ba.bed_prodVec(gb, xx, rows, cols, center=c, scale=s))
        np.testing.assert_array_equal(f, g)
        ref = orc.bed_prodVec(ob, xx, rows, cols, c, s)
        np.testing.assert_allclose(f, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        f, g = _both_paths(lambda: ba.bed_cprodVec(gb, yy, rows, cols, center=c, scale=s))
        np.testing.assert_array_equal(f, g)
        ref = orc.bed_cprodVec(ob, yy, rows, cols, c, s)
        np.testing.assert_allclose(f, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def test_knowledge_comes_from_counts_and_is_per_variant(ba, orc, golden_dir):
    """example-missing.bed: variants without missing values exist next to variants with; an
    operator over the complete ones takes the short path only after a count has seen them."""
    path = os.path.join(golden_dir, "example-missing.bed")
    ob, gb = orc.BedFile(path), ba.bed(path)
    na = orc.bed_col_counts(ob)[3]
    full, holes = np.nonzero(na == 0)[0], np.nonzero(na > 0)[0]
    if full.size < 8:
        pytest.skip("fixture has too few complete variants")
    rng = np.random.default_rng(4)
    y = rng.normal(size=ob.n)
    for cols in (full, np.r_[full[:5], holes[:5]]):
        sc = orc.bed_scaleBinom(ob, None, cols)
        before = ba.bed_cprodVec(gb, y, None, cols, center=sc["center"], scale=sc["scale"])
        ba.bed_counts(gb)                                       # now the image knows
        after, general = _both_paths(
            lambda: ba.bed_cprodVec(gb, y, None, cols, center=sc["center"], scale=sc["scale"]))
        np.testing.assert_array_equal(before, after)
        np.testing.assert_array_equal(after, general)
        ref = orc.bed_cprodVec(ob, y, None, cols, sc["center"], sc["scale"])
        np.testing.assert_allclose(after, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def test_svd_and_fbm_on_complete_data(ba, orc):
    n, m = 1203, 800
    ob = orc.fake_bed(n, m, na16=0, seed=11)
    gb = ba.bed.synthetic(n, m, na16=0, seed=11)
    f, g = _both_paths(lambda: ba.bed_randomSVD(gb, k=6, tol=1e-8))
    np.testing.assert_array_equal(f["d"], g["d"])
    np.testing.assert_array_equal(f["u"], g["u"])
    ref = orc.dense_svd(ob, k=6)
    np.testing.assert_allclose(f["d"], ref["d"], rtol=1e-6)   # the tolerance of north_star
    # FBM.code256 images learn it at creation
    G = ba.FBM_code256(orc.fbm_from_bed(ob).bytes)
    beta = np.random.default_rng(5).normal(size=m)
    f, g = _both_paths(lambda: ba.snp_PRS(G, beta))
    np.testing.assert_array_equal(f, g)


def test_ld_on_complete_data_uses_one_product_and_is_identical(ba, orc):
    """Without missing values among the selected samples only the cross-product GEMM is run (the other
    five pairwise sums are per-variant totals); r, LD scores and clumping must be bit-identical to
    the six-product path and match the oracle."""
    n, m = 700, 320
    ob = orc.fake_bed(n, m, na16=0, seed=21)
    gb = ba.bed.synthetic(n, m, na16=0, seed=21)
    rng = np.random.default_rng(21)
    ir = np.sort(rng.choice(n, 500, replace=False))
    ic = np.sort(rng.choice(m, 250, replace=False))
    pos = np.cumsum(rng.uniform(0.5, 3.0, ic.size))
    for rows in (None, ir):
        f, g = _both_paths(lambda: ba.bed_cor(gb, rows, ic, size=0.03, infos_pos=pos, thr_r2=0.001))
        for a, b in ((f.p, g.p), (f.i, g.i), (f.x, g.x)):
            np.testing.assert_array_equal(a, b)
        ref = orc.snp_cor(ob, rows, ic, size=0.03, infos_pos=pos, thr_r2=0.001)
        np.testing.assert_array_equal(f.i, ref[0]); np.testing.assert_array_equal(f.p, ref[1])
        np.testing.assert_allclose(f.x, ref[2], rtol=0, atol=1e-12)
        f, g = _both_paths(lambda: ba.bed_ld_scores(gb, rows, ic, size=0.03, infos_pos=pos))
        np.testing.assert_array_equal(f, g)
        np.testing.assert_allclose(f, orc.ld_scores(ob, rows, ic, size=0.03, infos_pos=pos), rtol=1e-12)
    chr_ = np.repeat([1, 2], [m // 2, m - m // 2])
    bp = 1000.0 * np.arange(m)
    f, g = _both_paths(lambda: ba.bed_clumping(gb, thr_r2=0.05, size=40, infos_chr=chr_, infos_pos=bp))
    np.testing.assert_array_equal(f, g)
    np.testing.assert_array_equal(f, orc.bed_clumping(ob, chr_, bp, thr_r2=0.05, size=40))
    G = ba.FBM_code256(orc.fbm_from_bed(ob).bytes)
    f, g = _both_paths(lambda: ba.snp_clumping(G, chr_, thr_r2=0.05, size=40, infos_pos=bp))
    np.testing.assert_array_equal(f, g)
    np.testing.assert_array_equal(f, orc.snp_clumping(orc.fbm_from_bed(ob), chr_, thr_r2=0.05, size=40, infos_pos=bp))


def test_cross_product_kernel_in_blocks_of_four_tile_pairs(ba, orc, monkeypatch):
    """round 6: k_quad_xy_f4 — four tile pairs per workgroup, their tiles brought once through LDS — against the
    one-pair-per-wave kernel (BSN_LD_NO_QUAD=1: bit-identical band), the six-product kernels and the oracle: band widths
    from under one tile to many, an odd number of tiles, row subsets (the keep-mask applied by the loading wave), a
    variant list that is not contiguous, clumping on an FBM"""
    from bigsnpr_amd import ld as ldm
    n, m = 1300, 2250                                   # 36 tiles of 64 variants (the last one ragged)
    ob = orc.fake_bed(n, m, na16=0, seed=33)
    gb = ba.bed.synthetic(n, m, na16=0, seed=33)
    rng = np.random.default_rng(33)
    ir = np.sort(rng.choice(n, 900, replace=False))
    ic = np.sort(rng.choice(m, 2100, replace=False))
    for rows, cols, size in ((None, None, 0.3), (ir, None, 0.07), (None, ic, 0.5), (ir, ic, 0.15)):
        mm = m if cols is None else cols.size
        pos = np.cumsum(rng.uniform(0.5, 1.5, mm))
        quad = ba.bed_ld_scores(gb, rows, cols, size=size, infos_pos=pos)
        assert "k_quad_xy_f4" in ldm.last_stats()["kernel"]
        monkeypatch.setenv("BSN_LD_NO_QUAD", "1")
        single = ba.bed_ld_scores(gb, rows, cols, size=size, infos_pos=pos)
        assert "k_pair_xy_f4" in ldm.last_stats()["kernel"]
        monkeypatch.delenv("BSN_LD_NO_QUAD")
        np.testing.assert_array_equal(quad, single)
        monkeypatch.setenv("BSN_FORCE_NA_PLANE", "1")
        np.testing.assert_array_equal(quad, ba.bed_ld_scores(gb, rows, cols, size=size, infos_pos=pos))
        monkeypatch.delenv("BSN_FORCE_NA_PLANE")
        np.testing.assert_allclose(quad, orc.ld_scores(ob, rows, cols, size=size, infos_pos=pos), rtol=1e-12)
    c1 = ba.bed_cor(gb, ir, None, size=0.1, infos_pos=np.arange(m, dtype=float), thr_r2=0.01)
    monkeypatch.setenv("BSN_LD_NO_QUAD", "1")
    c2 = ba.bed_cor(gb, ir, None, size=0.1, infos_pos=np.arange(m, dtype=float), thr_r2=0.01)
    monkeypatch.delenv("BSN_LD_NO_QUAD")
    for a, b in ((c1.p, c2.p), (c1.i, c2.i), (c1.x, c2.x)):
        np.testing.assert_array_equal(a, b)
    Go = orc.fbm_from_bed(ob)
    G = ba.FBM_code256(Go.bytes)
    chr_ = np.repeat([1, 2, 3], [900, 900, 450])
    keep = ba.snp_clumping(G, chr_, thr_r2=0.1, size=200)
    np.testing.assert_array_equal(keep, orc.snp_clumping(Go, chr_, thr_r2=0.1, size=200))
    monkeypatch.setenv("BSN_LD_NO_QUAD", "1")
    np.testing.assert_array_equal(ba.snp_clumping(G, chr_, thr_r2=0.1, size=200), keep)
