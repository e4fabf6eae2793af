"""bed_randomSVD with the default fun.scaling (bed_scaleBinom, R/binom-scaling.R:133-142) evaluates the
scaling INSIDE the solve: the code counts ride along the first crossproduct pass.  The values must be
bit-identical to bed_scaleBinom (and to the oracle's restatement of src/bed-fun.cpp:9-46), and the solve
must agree with the one that gets the same scaling from a separate statistics pass."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def test_fused_scaling_equals_separate_pass(ba, orc, golden_dir, missing_bed):
    gb = ba.bed(os.path.join(golden_dir, "example-missing.bed"))
    sc = orc.bed_scaleBinom(missing_bed)
    ic = np.nonzero(sc["scale"] > 0)[0]
    separate = ba.bed_randomSVD(gb, fun_scaling=lambda *a, **k: ba.bed_scaleBinom(*a, **k), ind_col=ic, k=5,
                                tol=1e-10, slices=7)
    gb2 = ba.bed(os.path.join(golden_dir, "example-missing.bed"))  # a handle no count has touched
    fused = ba.bed_randomSVD(gb2, ind_col=ic, k=5, tol=1e-10, slices=7)
    assert fused["fused_stats"] and not separate["fused_stats"]
    assert fused["nops"] == separate["nops"]
    for key in ("center", "scale"):
        np.testing.assert_array_equal(fused[key], sc[key][ic])
        np.testing.assert_array_equal(separate[key], sc[key][ic])
    np.testing.assert_allclose(fused["d"], separate["d"], rtol=1e-12)
    ref = orc.dense_svd(missing_bed, None, ic, k=5)
    np.testing.assert_allclose(fused["d"], ref["d"], rtol=1e-9)


def test_fused_counts_teach_the_handle_which_variants_are_complete(ba, orc):
    """complete data: the counting pass finds no missing value, the rest of the solve runs on the
    kernels without the missing-value plane and later operators know it too; the result is the one of
    the general kernels bit for bit (the skipped plane would add exact zeros)."""
    n, m = 3001, 2000
    ob = orc.fake_bed(n, m, seed=5, na16=0)
    payload = ob.payload
    gb = ba.bed.from_payload(payload, n, m)      # nothing known about missing values
    res = ba.bed_randomSVD(gb, k=8)
    os.environ["BSN_FORCE_NA_PLANE"] = "1"
    try:
        gen = ba.bed_randomSVD(ba.bed.from_payload(payload, n, m), k=8)
    finally:
        os.environ.pop("BSN_FORCE_NA_PLANE", None)
    np.testing.assert_array_equal(res["d"], gen["d"])
    np.testing.assert_array_equal(res["u"], gen["u"])
    ref = orc.dense_svd(ob, k=8)
    np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-6)
    sc = orc.bed_scaleBinom(ob)
    np.testing.assert_array_equal(res["center"], sc["center"])
    np.testing.assert_array_equal(res["scale"], sc["scale"])


def test_row_subset_takes_the_separate_pass(ba, orc, golden_dir, missing_bed):
    gb = ba.bed(os.path.join(golden_dir, "example-missing.bed"))
    rng = np.random.default_rng(2)
    ir = np.sort(rng.choice(missing_bed.n, 160, replace=False))
    sc = orc.bed_scaleBinom(missing_bed, ir, None)
    ic = np.nonzero(sc["scale"] > 0)[0]
    res = ba.bed_randomSVD(gb, ind_row=ir, ind_col=ic, k=4, tol=1e-10, slices=7)
    assert not res["fused_stats"]
    np.testing.assert_array_equal(res["center"], sc["center"][ic])
    np.testing.assert_array_equal(res["scale"], sc["scale"][ic])
    ref = orc.dense_svd(missing_bed, ir, ic, k=4)
    np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-9)


def test_more_than_half_missing_warns_like_bed_colstats(ba, orc):
    """src/bed-fun.cpp:40-41: '%d variants have >50% missing values.'"""
    n, m = 400, 300
    ob = orc.fake_bed(n, m, seed=9)
    payload = ob.payload.copy().reshape(m, -1)
    payload[7, : payload.shape[1] * 3 // 4] = 0x55        # code 01 = missing for 3/4 of variant 7
    gb = ba.bed.from_payload(payload.reshape(-1), n, m)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ba.bed_randomSVD(gb, k=3)
    assert any("1 variants have >50% missing values." in str(x.message) for x in w)


def test_unconverged_solve_warns(ba, orc):
    """a basis too small for the request: results are returned with a warning (RSpectra warns too)"""
    n, m = 600, 500
    gb = ba.bed.synthetic(n, m, seed=4)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = ba.bed_randomSVD(gb, k=10, tol=1e-12, slices=7, max_basis=16, block=4, max_restarts=-1)
    assert not res["converged"]
    assert any("did not converge" in str(x.message) for x in w)
    assert np.all(np.isfinite(res["d"]))
    # with thick restarts (the default) the same small basis gets there
    res = ba.bed_randomSVD(gb, k=10, tol=1e-8, slices=7, max_basis=48, block=4, verbose=1)
    assert res["converged"] and res["basis"] <= 48
    ref = orc.dense_svd(orc.fake_bed(n, m, seed=4), k=10)
    np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-8)
