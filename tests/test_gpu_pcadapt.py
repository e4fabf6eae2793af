"""multLinReg (src/multLinReg.cpp) against the oracle; snp_pcadapt / bed_pcadapt sanity
(tests/testthat/test-4-pcadapt.R: genomic control gives lambda_GC = 1, bed == FBM)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def test_multlinreg_matches_oracle(ba, orc, golden_dir, example_bed, missing_bed):
    rng = np.random.default_rng(0)
    for name, ob in (("example.bed", example_bed), ("example-missing.bed", missing_bed)):
        gb = ba.bed(os.path.join(golden_dir, name))
        U = np.linalg.qr(rng.normal(size=(ob.n, 4)))[0]
        ref = orc.multLinReg(ob, None, None, U, ncores=4)
        got = ba.multLinReg(gb, None, None, U)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-7, atol=1e-7)
        # row / column subsets, rows sampled with replacement
        ir = rng.choice(ob.n, ob.n // 2, replace=True)
        ic = rng.choice(ob.m, 200, replace=False)
        U2 = rng.normal(size=(ir.size, 3))
        ref = orc.multLinReg(ob, ir, ic, U2)
        got = ba.multLinReg(gb, ir, ic, U2)
        ok = ~np.isnan(ref)
        assert np.array_equal(np.isnan(got), ~ok)
        np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-7, atol=1e-7)
    with pytest.raises(ValueError, match="Incompatibility between dimensions"):
        ba.multLinReg(gb, None, None, U[:-1])


def test_pcadapt_sanity(ba, orc, golden_dir, example_bed):
    gb = ba.bed(os.path.join(golden_dir, "example.bed"))
    G = ba.FBM_code256(orc.fbm_from_bed(example_bed).bytes)
    svd = ba.big_randomSVD(G, ba.snp_scaleBinom(), k=10)
    a = ba.snp_pcadapt(G, svd["u"][:, :3])
    b = ba.bed_pcadapt(gb, svd["u"][:, :3])
    np.testing.assert_array_equal(a["score"], b["score"])
    assert a["score"].shape == (example_bed.m,) and a["lamGC"] > 0
    p = a["predict"](log10=False)
    assert np.nanmin(p) >= 0 and np.nanmax(p) <= 1
    assert abs(np.nanmedian(p) - 0.5) < 1e-6                # always genomic-controlled (test-4-pcadapt.R:39)
    one = ba.snp_pcadapt(G, svd["u"][:, 0])                  # K = 1 path
    t = one["tscores"][:, 0]
    np.testing.assert_allclose(one["score"], (t - np.median(t)) ** 2)
