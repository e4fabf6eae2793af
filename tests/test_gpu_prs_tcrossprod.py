"""GPU parity for snp_PRS (R/PRS.R) and bed_tcrossprodSelf (R/bed-tcrossprodSelf.R)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def test_snp_prs_matches_oracle_and_reference_properties(ba, orc, example_bed):
    """test-6-PRS.R:33-70: dims, threshold-order invariance, message without thresholds; values
    against the oracle's restatement of R/PRS.R (itself checked against dense G %*% beta)"""
    Go = orc.fbm_from_bed(example_bed)
    G = ba.FBM_code256(Go.bytes)
    rng = np.random.default_rng(0)
    keep = np.sort(rng.choice(Go.m, 1500, replace=False))
    betas = rng.normal(size=keep.size)
    lpS = rng.uniform(0, 6, size=keep.size)
    same = rng.uniform(size=keep.size) > 0.2
    test = rng.choice(Go.n, 400, replace=False)
    thrs = np.arange(0, 5.5, 0.5)
    prs = ba.snp_PRS(G, betas, test, keep, same, lpS, thrs)
    assert prs.shape == (test.size, thrs.size)
    ref = orc.snp_PRS(Go, betas, test, keep, same, lpS, thrs)
    np.testing.assert_allclose(prs, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    perm = rng.permutation(thrs.size)
    prs2 = ba.snp_PRS(G, betas, test, keep, same, lpS, thrs[perm])
    np.testing.assert_array_equal(prs[:, perm], prs2)        # bit-reproducible accumulation
    with pytest.warns(UserWarning, match="Thresholding disabled"):
        p0 = ba.snp_PRS(G, betas, test, keep, same)
    np.testing.assert_allclose(p0[:, 0], ref[:, 0] * 0 + orc.snp_PRS(Go, betas, test, keep, same)[:, 0],
                               atol=1e-9 * np.abs(ref).max())
    with pytest.raises(ValueError, match="Incompatibility between dimensions"):
        ba.snp_PRS(G, betas[:-1], test, keep, same, lpS, thrs)


def test_tcrossprod_self(ba, orc, golden_dir, example_bed, missing_bed):
    """test-2-bed-clumping-SVD.R:76-79: sqrt(eigen(K)) == svd$d; K against the dense oracle"""
    gb = ba.bed(os.path.join(golden_dir, "example.bed"))
    ic = np.arange(0, example_bed.m, 3)
    K, ms = ba.bed_tcrossprodSelf(gb, ind_col=ic, block_size=500)
    Kref, msref = orc.bed_tcrossprodSelf(example_bed, None, ic)
    np.testing.assert_array_equal(ms["center"], msref["center"])
    np.testing.assert_allclose(K, Kref, rtol=0, atol=1e-10 * np.abs(Kref).max())
    np.testing.assert_allclose(K, K.T, atol=1e-12 * np.abs(K).max())
    svd = ba.bed_randomSVD(gb, ind_col=ic, k=10, tol=1e-10, slices=7)
    ev = np.linalg.eigvalsh(K)[::-1][:10]
    np.testing.assert_allclose(np.sqrt(ev), svd["d"], rtol=1e-8)
    # with missing values and a row subset (n not a multiple of the tile)
    gm = ba.bed(os.path.join(golden_dir, "example-missing.bed"))
    rng = np.random.default_rng(1)
    ir = rng.choice(missing_bed.n, 77, replace=False)
    sc = orc.bed_scaleBinom(missing_bed, ir)
    icm = np.nonzero(sc["scale"] > 0)[0]
    K2, _ = ba.bed_tcrossprodSelf(gm, ind_row=ir, ind_col=icm)
    K2ref, _ = orc.bed_tcrossprodSelf(missing_bed, ir, icm)
    np.testing.assert_allclose(K2, K2ref, rtol=0, atol=1e-10 * np.abs(K2ref).max())


def test_tcrossprod_self_tiles_and_slabs(ba, orc):
    """the tiled kernel beyond one 128-sample tile: n not a multiple of 16, missing values, all rows (word
    decode) against the oracle, a permuted / repeated row selection (byte gather) against the same matrix
    re-indexed, a scattered variant selection, and exact symmetry (only the upper tiles are computed)"""
    bo = orc.fake_bed(1003, 2500, seed=77, na16=1310)
    gb = ba.bed.from_payload(bo.payload, bo.n, bo.m)
    sc = orc.bed_scaleBinom(bo)
    ic = np.nonzero(sc["scale"] > 0)[0]
    K, _ = ba.bed_tcrossprodSelf(gb, ind_col=ic)
    Kref, _ = orc.bed_tcrossprodSelf(bo, None, ic)
    scale_ = np.abs(Kref).max()
    np.testing.assert_allclose(K, Kref, rtol=0, atol=1e-10 * scale_)
    np.testing.assert_array_equal(K, K.T)
    rng = np.random.default_rng(5)
    ir = np.concatenate([rng.permutation(bo.n)[:700], [3, 3, 1002]])
    fs = lambda obj, ind_row, ind_col, ncores=1: dict(center=sc["center"][ind_col], scale=sc["scale"][ind_col])
    Kall, _ = ba.bed_tcrossprodSelf(gb, fun_scaling=fs, ind_col=ic)
    Ksub, _ = ba.bed_tcrossprodSelf(gb, fun_scaling=fs, ind_row=ir, ind_col=ic)
    np.testing.assert_allclose(Ksub, Kall[np.ix_(ir, ir)], rtol=0, atol=1e-11 * scale_)
    ic2 = ic[rng.permutation(ic.size)[:900]]
    K3, _ = ba.bed_tcrossprodSelf(gb, fun_scaling=fs, ind_col=ic2)
    K3ref, _ = orc.bed_tcrossprodSelf(bo, None, np.sort(ic2))
    np.testing.assert_allclose(K3, K3ref, rtol=0, atol=1e-10 * scale_)


def test_prod_and_rowsumssq_and_self_projection(ba, orc, golden_dir, missing_bed, example_bed):
    """src/bed-fun.cpp:103-133 + tests/testthat/test-2-pca-project.R (simple projection):
    projecting the SVD's own rows gives u d, and rowSumsSq == rowSums(X^2)"""
    gm = ba.bed(os.path.join(golden_dir, "example-missing.bed"))
    rng = np.random.default_rng(3)
    ir = rng.choice(missing_bed.n, 120, replace=True)
    sc = orc.bed_scaleBinom(missing_bed)
    ic = np.nonzero(sc["scale"] > 0)[0][::2]
    V = rng.normal(size=(ic.size, 5))
    XV, rs = ba.prod_and_rowSumsSq(gm, ir, ic, sc["center"][ic], sc["scale"][ic], V)
    XVr, rsr = orc.prod_and_rowSumsSq(missing_bed, ir, ic, sc["center"][ic], sc["scale"][ic], V)
    np.testing.assert_allclose(XV, XVr, rtol=0, atol=1e-9 * np.abs(XVr).max())
    np.testing.assert_allclose(rs, rsr, rtol=1e-9)
    with pytest.raises(ValueError, match="Incompatibility between dimensions"):
        ba.prod_and_rowSumsSq(gm, ir, ic, sc["center"][ic], sc["scale"][ic], V[:-1])
    ge = ba.bed(os.path.join(golden_dir, "example.bed"))
    svd = ba.bed_randomSVD(ge, k=5, tol=1e-10, slices=7)
    svd["subset"] = np.arange(example_bed.m)
    proj = ba.bed_projectSelfPCA(svd, ge, ind_row=np.arange(example_bed.n))
    np.testing.assert_allclose(proj["simple_proj"], svd["u"] * svd["d"], rtol=0, atol=1e-6 * svd["d"][0])


def test_fbm_twin_of_the_projection(ba, orc, golden_dir, example_bed, missing_bed):
    """src/project-utils.cpp:12-43 + test-2-pca-project.R:65-70: the FBM path gives the same
    simple projection as the bed path; on an FBM with missing codes the affected rows are NA."""
    ge = ba.bed(os.path.join(golden_dir, "example.bed"))
    Go = orc.fbm_from_bed(example_bed)
    G = ba.FBM_code256(Go.bytes)
    rng = np.random.default_rng(8)
    ir = np.sort(rng.choice(example_bed.n, 400, replace=False))
    svd = ba.bed_randomSVD(ge, ind_row=ir, k=5, tol=1e-8)
    test = np.setdiff1d(np.arange(example_bed.n), ir)
    with pytest.raises(ValueError, match="'ind.col' can't be `NULL`."):
        ba.snp_projectSelfPCA(svd, G, ind_row=test)
    with pytest.raises(ValueError, match="Incompatibility between dimensions"):
        ba.snp_projectSelfPCA(svd, G, ind_row=test, ind_col=np.arange(5))
    p_bed = ba.bed_projectSelfPCA(svd, ge, ind_row=np.arange(example_bed.n), ind_col=np.arange(example_bed.m))
    p_fbm = ba.snp_projectSelfPCA(svd, G, ind_row=test, ind_col=np.arange(example_bed.m))
    np.testing.assert_array_equal(p_fbm["simple_proj"], p_bed["simple_proj"][test])
    # test-2-pca-project.R:49-57, 66: simple projections of left-out individuals are shrunk towards 0, the OADP
    # projections sit closer to the population centres of the reference PCs; FBM and bed paths agree
    np.testing.assert_allclose(p_fbm["OADP_proj"], p_bed["OADP_proj"][test], rtol=1e-9, atol=1e-9 * svd["d"][0])
    pop = np.repeat([1, 2, 3], [143, 167, 207])
    med = lambda X, who: np.array([np.median(X[pop[who] == c][:, 1:3], axis=0) for c in (1, 2, 3)])
    refm = med(svd["u"] * svd["d"], ir)
    pred1, pred2 = med(p_bed["simple_proj"][test], test), med(p_bed["OADP_proj"][test], test)
    assert (refm ** 2).sum() > (pred1 ** 2).sum()
    assert ((refm - pred2) ** 2).sum() < ((refm - pred1) ** 2).sum()
    ref, ref_rs = orc.prod_and_rowSumsSq2(Go, test, None, svd["center"], svd["scale"], svd["v"])
    np.testing.assert_allclose(p_fbm["simple_proj"], ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    # missing codes: NA rows exactly where the reference's accessor would produce them
    Gm_o = orc.fbm_from_bed(missing_bed)
    Gm = ba.FBM_code256(Gm_o.bytes)
    sc = orc.bed_scaleBinom(missing_bed)
    ic = np.nonzero(sc["scale"] > 0)[0][:60]
    V = rng.normal(size=(ic.size, 3))
    XV, rs = ba.prod_and_rowSumsSq2(Gm, None, ic, sc["center"][ic], sc["scale"][ic], V)
    XVr, rsr = orc.prod_and_rowSumsSq2(Gm_o, None, ic, sc["center"][ic], sc["scale"][ic], V)
    assert np.array_equal(np.isnan(rs), np.isnan(rsr)) and np.isnan(rsr).any() and not np.isnan(rsr).all()
    ok = ~np.isnan(rsr)
    np.testing.assert_allclose(XV[ok], XVr[ok], rtol=0, atol=1e-9 * np.abs(XVr[ok]).max())
    np.testing.assert_allclose(rs[ok], rsr[ok], rtol=1e-9)
    assert np.isnan(XV[~ok]).all()
