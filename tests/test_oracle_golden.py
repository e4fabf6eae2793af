"""Pins the CPU oracle (oracle/bsn_oracle.c) against the reference's own golden
data and known-answer tests.  CPU only."""
import os

import numpy as np
import pytest


def _plink_pairs(golden_dir):
    rows = [l.split() for l in open(os.path.join(golden_dir, "example.ld"))][1:]
    a = np.array([int(r[2][3:]) for r in rows])
    b = np.array([int(r[5][3:]) for r in rows])
    r2 = np.array([float(r[6]) for r in rows])
    return a, b, r2


def test_bed_header_validation(orc, golden_dir):
    # src/bed-acc-xptr.cpp:14-35
    raw = np.fromfile(os.path.join(golden_dir, "example.bed"), dtype=np.uint8)
    orc.BedFile(raw=raw, n=517, m=4542)
    with pytest.raises(ValueError, match="n or p does not match"):
        orc.BedFile(raw=raw, n=516, m=4542)
    bad = raw.copy(); bad[0] = 0
    with pytest.raises(ValueError, match="not a binary PED"):
        orc.BedFile(raw=bad, n=517, m=4542)
    bad = raw.copy(); bad[2] = 0
    with pytest.raises(ValueError, match="Variant-major"):
        orc.BedFile(raw=bad, n=517, m=4542)


def test_decode_lut(orc):
    # src/bed-acc.h:22-37: 2-bit value 0,1,2,3 -> 2, NA, 1, 0
    lut = np.empty(4 * 256, dtype=np.int32)
    orc.lib().orc_decode_lut(lut.ctypes.data_as(orc.i32p), 3)
    lut = lut.reshape(256, 4)
    num = np.array([2, 3, 1, 0])
    for k in (0, 1, 0b11100100, 255, 0b01101100):
        for i in range(4):
            assert lut[k, i] == num[(k >> (2 * i)) & 3]


@pytest.mark.parametrize("size", [37, 200])
def test_cormat_vs_plink(orc, example_bed, golden_dir, size):
    """tests/testthat/test-2-corr.R:21-58: same sparsity pattern and r2 as PLINK."""
    a, b, r2 = _plink_pairs(golden_dir)
    bed = example_bed
    thr = np.full(bed.n, np.sqrt(0.2))
    i, p, x = orc.corMat(bed, None, None, size, thr, np.arange(1, bed.m + 1.0),
                         fill_diag=False, ncores=8)
    j = np.repeat(np.arange(bed.m), np.diff(p))
    sel = (b - a) <= size
    want = dict(zip(zip(a[sel], b[sel]), r2[sel]))
    got = dict(zip(zip(i, j), x * x))
    assert set(want) == set(got)                      # @i / @p identical
    assert max(abs(want[k] - got[k]) for k in want) < 1e-6
    # rows ascending inside each column (CSC order of R/corr.R:43-47)
    for c in range(bed.m):
        assert np.all(np.diff(i[p[c]:p[c + 1]]) > 0)


@pytest.mark.slow
def test_cormat_vs_plink_all_pairs(orc, example_bed, golden_dir):
    a, b, r2 = _plink_pairs(golden_dir)
    bed = example_bed
    thr = np.full(bed.n, np.sqrt(0.2))
    i, p, x = orc.corMat(bed, None, None, 1e9, thr, np.arange(1, bed.m + 1.0),
                         fill_diag=False, ncores=8)
    j = np.repeat(np.arange(bed.m), np.diff(p))
    want = dict(zip(zip(a, b), r2))
    got = dict(zip(zip(i, j), x * x))
    assert set(want) == set(got) and len(want) == 1431
    assert max(abs(want[k] - got[k]) for k in want) < 1e-6


def test_fbm_and_bed_cor_agree(orc, missing_bed):
    # tests/testthat/test-2-corr.R:148-159 (bed_cor == snp_cor)
    G = orc.fbm_from_bed(missing_bed)
    r1 = orc.snp_cor(missing_bed, size=30, ncores=4)
    r2 = orc.snp_cor(G, size=30, ncores=4)
    for u, v in zip(r1, r2):
        np.testing.assert_array_equal(u, v)
    i, p, x = r1
    # diagonal is last in each column and equals 1 (src/corr.cpp:36-39)
    assert np.all(i[p[1:] - 1] == np.arange(missing_bed.m))
    assert np.all(x[p[1:] - 1] == 1.0)


def test_cor_matches_numpy_pairwise_complete(orc, missing_bed):
    # tests/testthat/test-2-corr.R:77-116 uses Hmisc::rcorr (pairwise complete)
    rng = np.random.default_rng(1)
    ir = rng.choice(missing_bed.n, 120, replace=False)
    ic = np.sort(rng.choice(missing_bed.m, 40, replace=False))
    g = orc.read_bed(missing_bed, ir, ic, na_val=-1).astype(float)
    g[g < 0] = np.nan
    i, p, x = orc.snp_cor(missing_bed, ir, ic, size=1e6, fill_diag=False)
    j = np.repeat(np.arange(ic.size), np.diff(p))
    for ii, jj, xx in zip(i, j, x):
        ok = ~np.isnan(g[:, ii]) & ~np.isnan(g[:, jj])
        ref = np.corrcoef(g[ok, ii], g[ok, jj])[0, 1]
        if np.isnan(ref):
            assert np.isnan(xx)
        else:
            assert abs(ref - xx) < 1e-12


def test_ld_scores_identity(orc, missing_bed):
    # tests/testthat/test-2-ld-scores.R:15-30: ld == colSums(cor^2), symmetric
    m = missing_bed.m
    i, p, x = orc.snp_cor(missing_bed, size=25, fill_diag=True)
    j = np.repeat(np.arange(m), np.diff(p))
    ld_ref = np.zeros(m)
    off = i != j
    xx = np.where(np.isnan(x), 0, x) ** 2
    np.add.at(ld_ref, j, xx)
    np.add.at(ld_ref, i[off], xx[off])
    ld = orc.ld_scores(missing_bed, size=25)
    np.testing.assert_allclose(ld, ld_ref, rtol=1e-12)
    # size = 0.5 (SNP units / 1000) -> all ones, test-2-ld-scores.R:46-50
    np.testing.assert_array_equal(orc.ld_scores(missing_bed, size=0.5), np.ones(m))


def test_prodvec_dense_identity(orc, missing_bed):
    """tests/testthat/test-5-bed-prod-vec.R:18-41 and test-7-OpenMP.R:27-63: random
    unsorted / with-replacement subsets, random centre/scale, vs dense %*%."""
    rng = np.random.default_rng(2)
    bed = missing_bed
    for rep in range(6):
        replace = rep % 2 == 1
        ir = rng.choice(bed.n, 150, replace=replace)
        ic = rng.choice(bed.m, 333, replace=replace)
        center = rng.normal(size=ic.size) if rep > 1 else np.zeros(ic.size)
        scale = rng.uniform(0.5, 2, size=ic.size) if rep > 1 else np.ones(ic.size)
        A = orc.read_bed_scaled(bed, ir, ic, center, scale)
        x = rng.normal(size=ic.size)
        y = rng.normal(size=ir.size)
        for nc in (1, 3):
            np.testing.assert_allclose(orc.bed_prodVec(bed, x, ir, ic, center, scale, nc),
                                       A @ x, rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(orc.bed_cprodVec(bed, y, ir, ic, center, scale, nc),
                                       A.T @ y, rtol=1e-10, atol=1e-10)
    # cprodVec is bit-deterministic across ncores
    a = orc.bed_cprodVec(bed, y, ir, ic, center, scale, 1)
    b = orc.bed_cprodVec(bed, y, ir, ic, center, scale, 4)
    np.testing.assert_array_equal(a, b)


def test_counts_colstats_scaling(orc, missing_bed, example_bed):
    # tests/testthat/test-2-bed-clumping-SVD.R:99-136
    for bed in (missing_bed, example_bed):
        g = orc.read_bed(bed, na_val=3)
        counts = orc.bed_col_counts(bed, ncores=2)
        for c in range(4):
            np.testing.assert_array_equal(counts[c], (g == c).sum(0))
        st = orc.bed_colstats(bed)
        gg = np.where(g == 3, 0, g).astype(float)
        np.testing.assert_array_equal(st["sumX"], gg.sum(0))
        np.testing.assert_array_equal(st["nb_nona_col"], (g != 3).sum(0))
        maf = orc.bed_MAF(bed)
        sc = orc.bed_scaleBinom(bed)
        np.testing.assert_allclose(sc["center"], 2 * maf["af"], rtol=1e-15)
        np.testing.assert_array_equal(maf["N"], (g != 3).sum(0))


def test_snp_colstats_matches_bed_without_na(orc, example_bed):
    G = orc.fbm_from_bed(example_bed)
    a = orc.snp_colstats(G, ncores=2)
    b = orc.bed_colstats(example_bed)
    np.testing.assert_array_equal(a["sumX"], b["sumX"])
    np.testing.assert_allclose(a["denoX"], b["denoX"], rtol=1e-14)


def test_clumping_bed_equals_fbm_and_unit_invariance(orc, example_bed, golden_dir):
    """tests/testthat/test-2-bed-clumping-SVD.R:33-39,47-48: bed_clumping(obj.bed)
    identical to snp_clumping(G, CHR, POS); position-unit invariance."""
    chrom, pos = orc.read_bim(os.path.join(golden_dir, "example.bed"))
    G = orc.fbm_from_bed(example_bed)
    k1 = orc.snp_clumping(G, chrom, infos_pos=pos, thr_r2=0.2)
    k2 = orc.bed_clumping(example_bed, chrom, pos, thr_r2=0.2)
    np.testing.assert_array_equal(k1, k2)
    k3 = orc.snp_clumping(G, chrom, infos_pos=pos / 1000, size=500 / 1000, thr_r2=0.2)
    np.testing.assert_array_equal(k1, k3)
    assert 0 < k1.size < example_bed.m


def test_clumping_vs_plink_golden(orc, example_bed, golden_dir):
    """tests/testthat/test-6-PRS.R:24-31: >98 % of kept SNPs are in PLINK's clumps.
    The reference prioritises by |GWAS score|; pval.rds (same pipeline) is a
    monotone transform of it, so S = -log10(p) gives the same order."""
    chrom, pos = orc.read_bim(os.path.join(golden_dir, "example.bed"))
    G = orc.fbm_from_bed(example_bed)
    pval = orc.read_rds(os.path.join(golden_dir, "pval.rds"))
    keep2 = orc.read_rds(os.path.join(golden_dir, "clumping.rds")) - 1
    keep = orc.snp_clumping(G, chrom, S=-np.log10(pval), size=250, infos_pos=pos)
    assert np.isin(keep, keep2).mean() > 0.98


def test_dense_svd_matches_tcrossprod_eigen(orc, example_bed):
    # tests/testthat/test-2-bed-clumping-SVD.R:76-79
    ic = np.arange(0, example_bed.m, 7)
    K, ms = orc.bed_tcrossprodSelf(example_bed, None, ic)
    ev = np.linalg.eigvalsh(K)[::-1][:10]
    svd = orc.dense_svd(example_bed, None, ic, k=10)
    np.testing.assert_allclose(np.sqrt(ev), svd["d"], rtol=1e-10)
    assert np.abs(svd["u"].mean(0)).max() < 1e-10   # colMeans(u) ~ 0 (:53)


def test_prs_threshold_accumulation(orc, example_bed):
    # R/PRS.R:61-69 + test-6-PRS.R:46-57: threshold order invariance
    G = orc.fbm_from_bed(example_bed)
    rng = np.random.default_rng(3)
    keep = np.sort(rng.choice(G.m, 300, replace=False))
    betas = rng.normal(size=keep.size)
    lpS = rng.uniform(0, 6, size=keep.size)
    same = rng.uniform(size=keep.size) > 0.2
    thrs = np.arange(0, 5.5, 0.5)
    prs = orc.snp_PRS(G, betas, None, keep, same, lpS, thrs)
    perm = rng.permutation(thrs.size)
    prs2 = orc.snp_PRS(G, betas, None, keep, same, lpS, thrs[perm])
    np.testing.assert_allclose(prs[:, perm], prs2, rtol=1e-12)
    g = orc.read_bed(example_bed, None, keep).astype(float)
    for t, thr in enumerate(thrs):
        sel = lpS > thr
        ref = np.where(same[sel], g[:, sel], 2 - g[:, sel]) @ betas[sel]
        np.testing.assert_allclose(prs[:, t], ref, rtol=1e-10, atol=1e-10)


def test_fake_bed_is_deterministic_and_structured(orc):
    b1 = orc.fake_bed(403, 257, seed=7)
    b2 = orc.fake_bed(403, 257, seed=7)
    np.testing.assert_array_equal(b1.raw, b2.raw)
    # column-offset invariance: shard [j0, j1) equals the slice of the full draw
    b3 = orc.fake_bed(403, 100, seed=7, j_begin=57)
    np.testing.assert_array_equal(b3.payload, b1.payload[57 * b1.n_byte:157 * b1.n_byte])
    counts = orc.bed_col_counts(b1)
    assert 0.001 < counts[3].sum() / (403 * 257) < 0.03   # ~1 % missing
    assert counts[:3].min(1).min() >= 0 and (counts[1] > 0).mean() > 0.9


def test_fbm_products_are_the_plain_definition(orc, example_bed):
    """orc_fbm_prodVec / orc_fbm_cprodVec (bigstatsr's products are external: restated as y = G x, z = G' x
    on decoded values) against numpy on the decoded matrix, for CODE_012 and a dosage table, with
    indices sampled with replacement"""
    rng = np.random.default_rng(1)
    G = orc.fbm_from_bed(example_bed)
    dosage = np.array([0, 1, 2, np.nan, 0, 1, 2] + list(np.round(np.arange(201) * 0.01, 2)) + [np.nan] * 48)
    raw = rng.integers(7, 208, size=(120, 90)).astype(np.uint8)
    for obj, dec in ((G, G.bytes.astype(np.float64)), (orc.FBM256(raw, dosage), dosage[raw])):
        ir, ic = rng.choice(obj.n, 70, replace=True), rng.choice(obj.m, 60, replace=True)
        x, y = rng.normal(size=ic.size), rng.normal(size=ir.size)
        A = dec[np.ix_(ir, ic)]
        np.testing.assert_allclose(orc.fbm_prodVec(obj, x, ir, ic), A @ x, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(orc.fbm_cprodVec(obj, y, ir, ic), A.T @ y, rtol=1e-12, atol=1e-12)
        st = orc.snp_colstats(obj, ir, ic)
        np.testing.assert_allclose(st["sumX"], A.sum(0), rtol=1e-12)
        np.testing.assert_allclose(st["denoX"], (A * A).sum(0) - A.sum(0) ** 2 / ir.size, rtol=1e-9, atol=1e-9)


def test_readbina_restatement_against_the_decoder(orc, golden_dir):
    """readbina + getCode() (src/read-plink.cpp:13-56, R/utils.R:21-31) give what snp_readBed stores: the decoded
    calls with 3 for a missing value (tests/testthat/test-1-readBed.R: `G[]` of the FBM equals the bed accessor)."""
    tab = orc.get_code()
    assert tab.shape == (4, 256) and list(tab[:, 0]) == [2, 2, 2, 2] and list(tab[:, 255]) == [0, 0, 0, 0]
    assert list(tab[:, 0b00011011]) == [0, 1, 3, 2]          # bit pairs 11, 10, 01, 00, lowest first
    for name in ("example.bed", "example-missing.bed"):
        path = os.path.join(golden_dir, name)
        ob = orc.BedFile(path)
        got, eof = orc.readbina(path, ob.n, ob.m, tab)
        np.testing.assert_array_equal(got, orc.read_bed(ob, na_val=3).astype(np.uint8))
        assert eof
