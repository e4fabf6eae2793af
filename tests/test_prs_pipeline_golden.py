"""The reference's PRS pipeline test (tests/testthat/test-6-PRS.R:13-44) against its three golden files —
pval.rds, clumping.rds (PLINK-derived) and scores-PRS.rds — on the shipped example data:

    snp_autoSVD  ->  big_univLogReg with the PCs as covariates  ->  snp_clumping on |score|  ->  snp_PRS

once with every step from the CPU oracle (pins the oracle's autoSVD loop, clumping and snp_PRS to reference-held
data; `not gpu`) and once with the genotype-touching steps on the GPU (`gpu`).  big_univLogReg is external
(bigstatsr); its restatement (oracle.univ_logreg: the per-variant logistic MLE) is pinned by pval.rds."""
import os

import numpy as np
import pytest

THRS = np.arange(0, 5.5, 0.5)


def _fam_y01(golden_dir):
    aff = np.array([int(line.split()[5]) for line in open(os.path.join(golden_dir, "example.fam"))])
    return aff - 1.0


def _check(orc, golden_dir, u, clump, prs):
    """u -> GWAS -> clumping -> PRS, each stage against its golden file as test-6-PRS.R asserts it"""
    ob = orc.BedFile(os.path.join(golden_dir, "example.bed"))
    Go = orc.fbm_from_bed(ob)
    gwas = orc.univ_logreg(Go, _fam_y01(golden_dir), u)
    pval2 = orc.read_rds(os.path.join(golden_dir, "pval.rds"))
    # expect_equal(pval, pval2, tolerance = 1e-4): mean relative difference
    assert np.mean(np.abs(gwas["pval"] - pval2)) / np.mean(np.abs(pval2)) < 1e-4
    keep = clump(np.abs(gwas["score"]))
    keep2 = orc.read_rds(os.path.join(golden_dir, "clumping.rds")) - 1
    assert np.isin(keep, keep2).mean() > 0.98
    lp = -np.log10(gwas["pval"])
    scores = prs(gwas["estim"][keep], keep, lp[keep], THRS)
    assert scores.shape == (ob.n, THRS.size)
    prs2 = np.asarray(orc.read_rds(os.path.join(golden_dir, "scores-PRS.rds"))["value"]).reshape((ob.n, THRS.size), order="F")
    cors = np.array([np.corrcoef(scores[:, j], prs2[:, j])[0, 1] for j in range(THRS.size)])
    np.testing.assert_allclose(cors, 1.0, atol=1e-3)
    # no ordering in `thrs` (test-6-PRS.R:46-57)
    perm = np.random.default_rng(0).permutation(THRS.size)
    scores_p = prs(gwas["estim"][keep], keep, lp[keep], THRS[perm])
    np.testing.assert_allclose(scores_p[:, np.argsort(perm)], scores, rtol=1e-12, atol=1e-12)


def test_pipeline_with_the_oracle_only(orc, golden_dir, example_bed):
    from oracle import autosvd_oracle as ao
    Go = orc.fbm_from_bed(example_bed)
    chrom, pos = orc.read_bim(os.path.join(golden_dir, "example.bed"))
    n, m = Go.n, Go.m
    st = orc.snp_colstats(Go)
    maf = np.minimum(st["sumX"] / (2.0 * n), 1 - st["sumX"] / (2.0 * n))
    svd, subset, lrldr = ao.auto_svd_loop(
        lambda keep: orc.dense_svd(example_bed, None, keep, k=10),
        lambda excl: orc.snp_clumping(Go, chrom, exclude=excl, thr_r2=0.2, size=500.0, infos_pos=pos),
        maf, n, np.arange(m), chrom, infos_pos=pos, n_all_cols=m)
    _check(orc, golden_dir, svd["u"],
           lambda S: orc.snp_clumping(Go, chrom, S=S, size=250, infos_pos=pos),
           lambda b, keep, lp, thrs: orc.snp_PRS(Go, b, ind_keep=keep, lpS_keep=lp, thr_list=thrs))


@pytest.mark.gpu
def test_pipeline_on_the_gpu(orc, golden_dir, example_bed):
    import bigsnpr_amd as ba
    G = ba.FBM_code256(orc.fbm_from_bed(example_bed).bytes)
    chrom, pos = orc.read_bim(os.path.join(golden_dir, "example.bed"))
    svd = ba.snp_autoSVD(G, chrom, pos, verbose=False)
    _check(orc, golden_dir, svd["u"],
           lambda S: ba.snp_clumping(G, chrom, S=S, size=250, infos_pos=pos),
           lambda b, keep, lp, thrs: np.asarray(ba.snp_PRS(G, b, ind_keep=keep, lpS_keep=lp, thr_list=thrs)))
