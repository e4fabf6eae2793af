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
