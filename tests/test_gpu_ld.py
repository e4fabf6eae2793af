"""GPU parity of snp_cor / bed_cor / ld_scores / clumping against the oracle (which is
itself pinned to PLINK's golden files).  Mirrors tests/testthat/test-2-corr.R,
test-2-ld-scores.R and test-2-bed-clumping-SVD.R:33-48.  Bars: sparsity pattern (@i, @p) and
clumping indices bit-exact; @x and LD scores within 1e-6 relative (asserted much tighter)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def _plink_pairs(golden_dir):
    rows = [l.split() for l in open(os.path.join(golden_dir, "example.ld"))][1:]
    a = np.array([int(r[2][3:]) for r in rows]); b = np.array([int(r[5][3:]) for r in rows])
    return a, b, np.array([float(r[6]) for r in rows])


def _same_cor(res, ref, tol=1e-12):
    i, p, x = ref
    np.testing.assert_array_equal(res.p, p)
    np.testing.assert_array_equal(res.i, i)
    both_nan = np.isnan(res.x) & np.isnan(x)
    assert np.all(np.isnan(res.x) == np.isnan(x))
    np.testing.assert_allclose(res.x[~both_nan], x[~both_nan], rtol=0, atol=tol)


def test_cor_vs_plink_golden(ba, golden_dir):
    """test-2-corr.R:21-58 directly against PLINK's file, all 1431 pairs"""
    gb = ba.bed(os.path.join(golden_dir, "example.bed"))
    a, b, r2 = _plink_pairs(golden_dir)
    res = ba.bed_cor(gb, size=1e6, thr_r2=0.2, fill_diag=False)
    j = np.repeat(np.arange(gb.ncol), np.diff(res.p))
    want = dict(zip(zip(a, b), r2)); got = dict(zip(zip(res.i, j), res.x ** 2))
    assert set(want) == set(got)
    assert max(abs(want[k] - got[k]) for k in want) < 1e-6


@pytest.mark.parametrize("size", [0.037, 0.2, 2.5])
def test_cor_bed_matches_oracle(ba, orc, golden_dir, example_bed, size):
    gb = ba.bed(os.path.join(golden_dir, "example.bed"))
    res = ba.bed_cor(gb, size=size, thr_r2=0.2, fill_diag=False)
    _same_cor(res, orc.snp_cor(example_bed, size=size, thr_r2=0.2, fill_diag=False, ncores=8))
    res = ba.bed_cor(gb, size=size, ind_col=np.arange(300, 1500))
    _same_cor(res, orc.snp_cor(example_bed, size=size, ind_col=np.arange(300, 1500), ncores=8))


def test_cor_missing_values_subsets_alpha(ba, orc, golden_dir, missing_bed):
    """test-2-corr.R:77-116,148-159: NA, row/column subsets, alpha thresholds, FBM == bed"""
    gb = ba.bed(os.path.join(golden_dir, "example-missing.bed"))
    G_o = orc.fbm_from_bed(missing_bed)
    G = ba.FBM_code256(G_o.bytes)
    rng = np.random.default_rng(0)
    ir = rng.choice(missing_bed.n, 100, replace=False)
    ic = np.sort(rng.choice(missing_bed.m, 250, replace=False))
    for kw in (dict(size=30), dict(size=0.05, alpha=0.1), dict(size=1e3, thr_r2=0.05, fill_diag=False),
               dict(size=3, alpha=0.05, thr_r2=0.01)):
        ref = orc.snp_cor(missing_bed, ir, ic, ncores=8, **kw)
        _same_cor(ba.bed_cor(gb, ir, ic, **kw), ref)
        _same_cor(ba.snp_cor(G, ir, ic, **kw), ref)
    # positions in other units (test-2-corr.R:120-144)
    pos = np.cumsum(rng.uniform(0.5, 3, ic.size))
    _same_cor(ba.snp_cor(G, ir, ic, size=0.004, infos_pos=pos),
              orc.snp_cor(missing_bed, ir, ic, size=0.004, infos_pos=pos))


def test_cor_nan_for_constant_column(ba, orc):
    """test-2-corr.R:163-171: zero-variance column -> NaN + warning"""
    rng = np.random.default_rng(1)
    g = rng.integers(0, 4, size=(200, 40)).astype(np.uint8)
    g[:, 7] = 1
    G = ba.FBM_code256(g); Go = orc.FBM256(g)
    with pytest.warns(UserWarning, match="NA or NaN values"):
        res = ba.snp_cor(G, size=1e3)
    _same_cor(res, orc.snp_cor(Go, size=1e3))
    assert np.isnan(res.x).any()


@pytest.mark.parametrize("budget_cols", [128, 384])
@pytest.mark.parametrize("which", ["missing", "complete"])
def test_band_in_blocks_of_columns(ba, orc, golden_dir, missing_bed, example_bed, monkeypatch, budget_cols, which):
    """The reference walks any window (src/corr.cpp:52-53, src/ld-scores.cpp same loop); a dense m x W band that does
    not fit HBM is processed in blocks of columns (BSN_LD_BAND_BUDGET forces the path here with a budget of 128 / 384
    columns, so that blocks, a ragged last block and windows that reach across two blocks all occur): @p, @i, @x and the
    LD scores are IDENTICAL to the one-block result — same kernels, same order of every sum — and match the oracle.
    `missing`: 200 x 500 with missing values (six-product kernels); `complete`: 517 samples x 1111 variants without
    (cross-product kernel)."""
    if which == "missing":
        ob, name, ic = missing_bed, "example-missing.bed", np.arange(7, missing_bed.m - 6)
    else:
        ob, name, ic = example_bed, "example.bed", np.arange(100, 100 + 1111)
    gb = ba.bed(os.path.join(golden_dir, name))
    G = ba.FBM_code256(orc.fbm_from_bed(ob).bytes)
    rng = np.random.default_rng(3)
    ir = np.sort(rng.choice(ob.n, ob.n - 17, replace=False))
    pos = np.cumsum(rng.uniform(0.5, 3, ic.size))
    cases = [dict(size=0.3, infos_pos=pos), dict(size=0.12, infos_pos=pos, alpha=0.2, thr_r2=0.02, fill_diag=False)]
    dense = [(ba.bed_cor(gb, ir, ic, **kw), ba.snp_cor(G, ir, ic, **kw)) for kw in cases]
    ld_dense = ba.bed_ld_scores(gb, ir, ic, size=0.3, infos_pos=pos)
    width = int(np.max(np.arange(ic.size) - np.searchsorted(pos, pos - 300.0, side="left")))
    assert width > 140                                                # windows wider than a block of 128 columns
    monkeypatch.setenv("BSN_LD_BAND_BUDGET", str(budget_cols * width * 8))
    for kw, (d_bed, d_fbm) in zip(cases, dense):
        for res, ref in ((ba.bed_cor(gb, ir, ic, **kw), d_bed), (ba.snp_cor(G, ir, ic, **kw), d_fbm)):
            np.testing.assert_array_equal(res.p, ref.p)
            np.testing.assert_array_equal(res.i, ref.i)
            np.testing.assert_array_equal(res.x, ref.x)
        _same_cor(ba.bed_cor(gb, ir, ic, **kw), orc.snp_cor(ob, ir, ic, ncores=8, **kw))
    ld = ba.bed_ld_scores(gb, ir, ic, size=0.3, infos_pos=pos)
    np.testing.assert_array_equal(ld, ld_dense)
    np.testing.assert_allclose(ld, orc.ld_scores(ob, ir, ic, size=0.3, infos_pos=pos), rtol=1e-12)
    if which == "complete":
        # a dosage FBM (byte image, other statistics kernels) through the same blocks
        g8 = rng.integers(7, 208, size=(150, 700)).astype(np.uint8)       # dosages 0.00 .. 2.00, none missing
        D = ba.FBM_code256(g8, code=ba.CODE_DOSAGE)
        monkeypatch.delenv("BSN_LD_BAND_BUDGET")
        ref8, ld8 = ba.snp_cor(D, size=200), ba.snp_ld_scores(D, size=200)
        monkeypatch.setenv("BSN_LD_BAND_BUDGET", str(budget_cols * 200 * 8))
        res8 = ba.snp_cor(D, size=200)
        np.testing.assert_array_equal(res8.p, ref8.p)
        np.testing.assert_array_equal(res8.i, ref8.i)
        np.testing.assert_array_equal(res8.x, ref8.x)
        np.testing.assert_array_equal(ba.snp_ld_scores(D, size=200), ld8)


def test_ld_scores(ba, orc, golden_dir, missing_bed, example_bed):
    """test-2-ld-scores.R:15-64"""
    gb = ba.bed(os.path.join(golden_dir, "example-missing.bed"))
    for size in (25, 0.004, 300):
        np.testing.assert_allclose(ba.bed_ld_scores(gb, size=size), orc.ld_scores(missing_bed, size=size),
                                   rtol=1e-12)
    np.testing.assert_array_equal(ba.bed_ld_scores(gb, size=0.0005), np.ones(missing_bed.m))
    ge = ba.bed(os.path.join(golden_dir, "example.bed"))
    ic = np.arange(1000, 2200)
    ld = ba.bed_ld_scores(ge, ind_col=ic, size=50)
    np.testing.assert_allclose(ld, orc.ld_scores(example_bed, ind_col=ic, size=50), rtol=1e-12)
    # == colSums(cor^2)
    c = ba.bed_cor(ge, ind_col=ic, size=50).tocsc()
    full = c + c.T - __import__("scipy.sparse", fromlist=["x"]).diags(c.diagonal())
    np.testing.assert_allclose(ld, np.asarray(full.multiply(full).sum(0)).ravel(), rtol=1e-10)


def test_clumping_identical_to_oracle(ba, orc, golden_dir, example_bed):
    """test-2-bed-clumping-SVD.R:33-48,70 and test-6-PRS.R:24-31: indices bit-exact"""
    path = os.path.join(golden_dir, "example.bed")
    gb = ba.bed(path)
    chrom, pos = orc.read_bim(path)
    Go = orc.fbm_from_bed(example_bed)
    G = ba.FBM_code256(Go.bytes)
    ref = orc.snp_clumping(Go, chrom, infos_pos=pos, thr_r2=0.2)
    np.testing.assert_array_equal(ba.snp_clumping(G, chrom, infos_pos=pos, thr_r2=0.2), ref)
    np.testing.assert_array_equal(ba.bed_clumping(gb, thr_r2=0.2), ref)          # bed == FBM
    np.testing.assert_array_equal(ba.snp_clumping(G, chrom, infos_pos=pos / 1000, size=0.5), ref)
    # other thresholds, window in SNP units, exclusions, row subset, custom statistic
    rng = np.random.default_rng(2)
    ir = np.sort(rng.choice(example_bed.n, 300, replace=False))
    excl = rng.choice(example_bed.m, 500, replace=False)
    S = rng.uniform(size=example_bed.m)
    for kw in (dict(thr_r2=0.05, size=200), dict(thr_r2=0.5, infos_pos=pos, size=100),
               dict(thr_r2=0.2, infos_pos=pos, ind_row=ir, exclude=excl),
               dict(thr_r2=0.1, infos_pos=pos, S=S)):
        np.testing.assert_array_equal(ba.snp_clumping(G, chrom, **kw), orc.snp_clumping(Go, chrom, **kw))
    np.testing.assert_array_equal(ba.bed_clumping(gb, ind_row=ir, exclude=excl, thr_r2=0.3),
                                  orc.bed_clumping(example_bed, chrom, pos, ind_row=ir, exclude=excl, thr_r2=0.3))
    # PLINK golden: > 98 % overlap when prioritising by the stored p-values
    pval = orc.read_rds(os.path.join(golden_dir, "pval.rds"))
    keep2 = orc.read_rds(os.path.join(golden_dir, "clumping.rds")) - 1
    keep = ba.snp_clumping(G, chrom, S=-np.log10(pval), size=250, infos_pos=pos)
    assert np.isin(keep, keep2).mean() > 0.98


def test_clumping_all_chromosomes_in_one_call(ba, orc, monkeypatch):
    """round 6: on a resident handle with whole-number positions the chromosomes of a clumping go to the device laid end to
    end in ONE call (ld.py: _clump_jobs) — the same indices as the chromosome-by-chromosome loop (BSN_CLUMP_PER_CHR=1) and
    as the oracle's loop: uneven chromosomes, tied statistics and tied positions, windows wider than a chromosome, a row
    subset, exclusions; fractional positions and unsorted labels keep to the loop"""
    n, m = 260, 2400
    ob = orc.fake_bed(n, m, seed=31, na16=0)
    gb = ba.bed.from_payload(ob.payload, n, m)
    Go = orc.fbm_from_bed(ob)
    G = ba.FBM_code256(Go.bytes)
    rng = np.random.default_rng(4)
    chrom = np.repeat([1, 2, 3, 4, 5, 6], [700, 40, 900, 3, 500, 257])
    pos = np.concatenate([np.sort(rng.integers(1, 2_000_000, size=c)) for c in (700, 40, 900, 3, 500, 257)]).astype(float)
    pos[100:110] = pos[100]                                            # tied positions
    ir = np.sort(rng.choice(n, 200, replace=False))
    excl = rng.choice(m, 300, replace=False)
    S_tied = rng.integers(0, 6, size=m).astype(float)
    for kw in (dict(thr_r2=0.2, infos_pos=pos), dict(thr_r2=0.05, size=150), dict(thr_r2=0.1, infos_pos=pos, size=5000),
               dict(thr_r2=0.2, infos_pos=pos, ind_row=ir, exclude=excl), dict(thr_r2=0.3, infos_pos=pos, S=S_tied, size=300)):
        one = ba.snp_clumping(G, chrom, **kw)
        np.testing.assert_array_equal(one, orc.snp_clumping(Go, chrom, **kw))
        monkeypatch.setenv("BSN_CLUMP_PER_CHR", "1")
        np.testing.assert_array_equal(ba.snp_clumping(G, chrom, **kw), one)
        monkeypatch.delenv("BSN_CLUMP_PER_CHR")
    for kw in (dict(thr_r2=0.2), dict(thr_r2=0.2, ind_row=ir, exclude=excl, size=300)):
        one = ba.bed_clumping(gb, infos_chr=chrom, infos_pos=pos, **kw)
        np.testing.assert_array_equal(one, orc.bed_clumping(ob, chrom, pos, **kw))
        monkeypatch.setenv("BSN_CLUMP_PER_CHR", "1")
        np.testing.assert_array_equal(ba.bed_clumping(gb, infos_chr=chrom, infos_pos=pos, **kw), one)
        monkeypatch.delenv("BSN_CLUMP_PER_CHR")
    # fractional positions (kb) and labels that are not in runs: the loop, same answer as the oracle
    np.testing.assert_array_equal(ba.snp_clumping(G, chrom, infos_pos=pos / 1000 + 0.25, size=0.5),
                                  orc.snp_clumping(Go, chrom, infos_pos=pos / 1000 + 0.25, size=0.5))
    mixed = rng.permutation(chrom)
    np.testing.assert_array_equal(ba.snp_clumping(G, mixed, thr_r2=0.2, size=100), orc.snp_clumping(Go, mixed, thr_r2=0.2, size=100))
    with pytest.raises(ValueError, match="not sorted"):
        ba.snp_clumping(G, chrom, infos_pos=pos[::-1].copy())


@pytest.mark.parametrize("batch", [37, 400])
def test_clumping_wide_windows_lazy_path(ba, orc, golden_dir, example_bed, monkeypatch, batch):
    """Windows whose dense r2 band would not fit the device take the batched candidate-vs-kept path
    (clump_lazy, ld.hip).  Forced here with a tiny band budget: the kept indices must stay bit-identical to
    the oracle (and hence to the dense path) for FBM and bed formulas, wide and narrow windows, a row subset
    with exclusions, and through the grid entry of snp_grid_clumping."""
    path = os.path.join(golden_dir, "example.bed")
    chrom, pos = orc.read_bim(path)
    Go = orc.fbm_from_bed(example_bed)
    rng = np.random.default_rng(12)
    ir = np.sort(rng.choice(example_bed.n, 350, replace=False))
    excl = rng.choice(example_bed.m, 300, replace=False)
    S = rng.uniform(size=example_bed.m)
    cases = (dict(thr_r2=0.01, infos_pos=pos, size=20000), dict(thr_r2=0.2, infos_pos=pos, size=500),
             dict(thr_r2=0.05, size=3000), dict(thr_r2=0.3, infos_pos=pos, ind_row=ir, exclude=excl, size=5000),
             dict(thr_r2=0.1, infos_pos=pos, S=S, size=2000))
    refs = [orc.snp_clumping(Go, chrom, **kw) for kw in cases]
    bed_ref = orc.bed_clumping(example_bed, chrom, pos, ind_row=ir, thr_r2=0.02, size=10000)
    monkeypatch.setenv("BSN_CLUMP_BAND_BUDGET", "20000")      # bytes: no chromosome's band fits
    monkeypatch.setenv("BSN_CLUMP_LAZY_BATCH", str(batch))
    G = ba.FBM_code256(Go.bytes)
    gb = ba.bed(path)
    for kw, ref in zip(cases, refs):
        np.testing.assert_array_equal(ba.snp_clumping(G, chrom, **kw), ref)
    np.testing.assert_array_equal(ba.bed_clumping(gb, ind_row=ir, thr_r2=0.02, size=10000), bed_ref)
    lpval = -np.log10(rng.uniform(size=Go.m))
    kw = dict(grid_thr_r2=(0.02, 0.5), grid_base_size=(100, 400))
    res = ba.snp_grid_clumping(G, chrom, pos, lpval, **kw)
    ref, _, _ = orc.snp_grid_clumping(Go, chrom, pos, lpval, **kw)
    for a, b in zip(res, ref):
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)


def test_clumping_with_missing_values(ba, orc, golden_dir, missing_bed):
    path = os.path.join(golden_dir, "example-missing.bed")
    gb = ba.bed(path)
    chrom, pos = orc.read_bim(path)
    np.testing.assert_array_equal(ba.bed_clumping(gb, thr_r2=0.1),
                                  orc.bed_clumping(missing_bed, chrom, pos, thr_r2=0.1))


def test_ld_errors_and_repeated_samples(ba, orc, golden_dir, missing_bed):
    gb = ba.bed(os.path.join(golden_dir, "example-missing.bed"))
    with pytest.raises(ValueError, match="not sorted"):
        ba.bed_cor(gb, infos_pos=np.arange(gb.ncol)[::-1])
    # a row list with repeats (a bootstrap draw): the reference's accessor takes any list (src/bed-acc.h:64-65); here
    # the sub-matrix is gathered once and the band runs on the copy
    rng = np.random.default_rng(4)
    ir = rng.integers(0, missing_bed.n, size=300)
    ic = np.arange(50, min(missing_bed.m, 450))
    with np.errstate(all="ignore"):
        _same_cor(ba.bed_cor(gb, ir, ic, size=40), orc.snp_cor(missing_bed, ir, ic, size=40))
    np.testing.assert_allclose(ba.bed_ld_scores(gb, ir, ic, size=40), orc.ld_scores(missing_bed, ir, ic, size=40), rtol=1e-12)


def test_clumping_at_a_threshold_that_sits_on_a_pair(ba, orc, golden_dir, missing_bed):
    """bed_clumping_chr decides `r2 > thr` per pair (src/clumping-bed.cpp:69-76).  The reference
    accumulates r over the samples in fp64 (error ~1e-13 relative); the GPU evaluates it from exact
    integer plane sums.  With the threshold placed ON the r2 of a pair that decides a variant's fate, the
    two sides of it must both agree with the oracle at a margin a thousand times that rounding noise;
    exactly on it, either side's answer is legitimate (it is a rounding-level tie)."""
    path = os.path.join(golden_dir, "example-missing.bed")
    gb = ba.bed(path)
    chrom, pos = orc.read_bim(path)
    n, m = missing_bed.n, missing_bed.m
    # the scaled matrix of the reference (missing -> 0) and r2 of all pairs in its window, as it computes them
    st = orc.bed_colstats(missing_bed)
    center, scale = st["sumX"] / st["nb_nona_col"], np.sqrt(st["denoX"])
    g = orc.read_bed(missing_bed, na_val=-1).astype(np.float64)
    A = np.where(g < 0, 0.0, (g - center) / scale)
    base = orc.bed_clumping(missing_bed, chrom, pos, thr_r2=0.2)
    # pairs (kept variant, pruned neighbour): their r2 exceeded 0.2 — pick a few and put the threshold on them
    pruned = np.setdiff1d(np.arange(m), base)
    rng = np.random.default_rng(0)
    tested = 0
    for j0 in rng.permutation(pruned)[:40]:
        near = base[(chrom[base] == chrom[j0]) & (np.abs(pos[base] - pos[j0]) <= 500e3)]
        if near.size == 0:
            continue
        r2 = np.array([np.sum(A[:, j] * A[:, j0]) ** 2 for j in near])
        t = float(r2.max())
        if not (0.2 < t < 0.95):
            continue
        for thr in (t * (1 - 1e-10), t * (1 + 1e-10)):
            np.testing.assert_array_equal(ba.bed_clumping(gb, thr_r2=thr, size=500),
                                          orc.bed_clumping(missing_bed, chrom, pos, thr_r2=thr, size=500))
        on = ba.bed_clumping(gb, thr_r2=t, size=500)
        sides = [orc.bed_clumping(missing_bed, chrom, pos, thr_r2=x, size=500) for x in (t * (1 - 1e-10), t * (1 + 1e-10))]
        assert any(np.array_equal(on, s_) for s_ in sides)
        tested += 1
        if tested == 4:
            break
    assert tested >= 2


def test_bed_clumping_on_the_four_product_kernel(ba, orc, monkeypatch):
    """round 6: with missing values and enough blocks to fill the chip the bed clumping band runs k_pair_stats_f4<SQ = false>
    — the four sums src/clumping-bed.cpp:69-73 reads, the two with squares neither decoded nor multiplied — against the
    oracle's loop (indices bit-exact), chromosome by chromosome and in one call; bed_ld_scores on the same handle keeps
    the six-product kernel and still matches"""
    from bigsnpr_amd import ld as ldm
    n, m = 420, 12032
    ob = orc.fake_bed(n, m, seed=77, na16=2000)                       # 3 % missing values
    gb = ba.bed.from_payload(ob.payload, n, m)
    chrom = np.repeat([1, 2], [7000, m - 7000])
    pos = (np.arange(m) * 1000.0)
    keep = ba.bed_clumping(gb, thr_r2=0.02, size=400, infos_chr=chrom, infos_pos=pos)
    assert "k_pair_stats_f4" in ldm.last_stats()["kernel"]
    np.testing.assert_array_equal(keep, orc.bed_clumping(ob, chrom, pos, thr_r2=0.02, size=400))
    assert 1000 < keep.size < m - 1000                                   # (sampling noise of 420 samples prunes most variants at 0.02)
    monkeypatch.setenv("BSN_CLUMP_PER_CHR", "1")
    np.testing.assert_array_equal(ba.bed_clumping(gb, thr_r2=0.02, size=400, infos_chr=chrom, infos_pos=pos), keep)
    monkeypatch.delenv("BSN_CLUMP_PER_CHR")
    ld = ba.bed_ld_scores(gb, size=0.4, infos_pos=pos / 1000.0)
    assert "k_pair_stats_f4" in ldm.last_stats()["kernel"]
    np.testing.assert_allclose(ld, orc.ld_scores(ob, size=0.4, infos_pos=pos / 1000.0), rtol=1e-12)


def test_raw_plane_kernel_equals_the_look_up_kernel_bit_for_bit(ba, orc, monkeypatch):
    """round 6, second session: k_pair_stats_f4<., RAW> multiplies planes that cost no look-up — the raw code c, its high
    bit H = [c >= 2] and M = [c = 3] — and recombines the six pairwise-complete sums of src/corr.cpp:54-75 from the six
    products and the per-variant totals (x = c - 3 M, x^2 = c + 2 H - 5 M, present = 1 - M): integer arithmetic, so
    every band entry is the one the look-up kernel (BSN_LD_LUT=1) writes — scores, @p / @i / @x and clumping indices
    bit for bit; row subsets (dropped samples are ORed to missing and counted in the totals' way), an increasing column
    subset, heavy and light missingness, ragged sizes; and against the oracle."""
    from bigsnpr_amd import ld as ldm
    rng = np.random.default_rng(41)
    for n, m, na16 in ((420, 12032, 2000), (1003, 9000, 655), (259, 10250, 20000), (64, 8200, 100)):
        ob = orc.fake_bed(n, m, seed=5 + n, na16=na16)
        gb = ba.bed.from_payload(ob.payload, n, m)
        pos = np.cumsum(rng.integers(1, 2000, size=m)).astype(np.float64)
        chrom = np.repeat([1, 2], [m // 2, m - m // 2])
        ir = np.sort(rng.choice(n, n - n // 5, replace=False))
        ic = np.sort(rng.choice(m, m - 300, replace=False))
        got = {}
        for tag in ("raw", "lut"):
            if tag == "lut":
                monkeypatch.setenv("BSN_LD_LUT", "1")
            # (the switch is read once per process: the A/B runs in children)
            code = (
                "import numpy as np, sys, pickle, bigsnpr_amd as ba\n"
                "from bigsnpr_amd import ld as ldm\n"
                "d = pickle.load(open(sys.argv[1], 'rb'))\n"
                "gb = ba.bed.from_payload(d['payload'], d['n'], d['m'])\n"
                "out = {}\n"
                "out['ld'] = ba.bed_ld_scores(gb, size=500, infos_pos=d['pos']); out['k_ld'] = ldm.last_stats()['kernel']\n"
                "out['ld_sub'] = ba.bed_ld_scores(gb, ind_row=d['ir'], ind_col=d['ic'], size=500, infos_pos=d['pos'][d['ic']])\n"
                "c = ba.bed_cor(gb, size=400, infos_pos=d['pos'], alpha=0.5); out['cor'] = (c.p, c.i, c.x)\n"
                "c = ba.bed_cor(gb, ind_row=d['ir'], size=400, infos_pos=d['pos'], thr_r2=0.01); out['cor_sub'] = (c.p, c.i, c.x)\n"
                "out['clump'] = ba.bed_clumping(gb, thr_r2=0.02, size=500, infos_chr=d['chrom'], infos_pos=d['pos']); out['k_cl'] = ldm.last_stats()['kernel']\n"
                "out['clump_sub'] = ba.bed_clumping(gb, ind_row=d['ir'], thr_r2=0.1, size=500, infos_chr=d['chrom'], infos_pos=d['pos'])\n"
                "pickle.dump(out, open(sys.argv[2], 'wb'))\n")
            import pickle, subprocess, sys, tempfile, os
            with tempfile.TemporaryDirectory() as tmp:
                fin, fout = os.path.join(tmp, "in.pkl"), os.path.join(tmp, "out.pkl")
                pickle.dump(dict(payload=ob.payload, n=n, m=m, pos=pos, chrom=chrom, ir=ir, ic=ic), open(fin, "wb"))
                r = subprocess.run([sys.executable, "-c", code, fin, fout], capture_output=True, text=True, timeout=600,
                                   cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=dict(os.environ))
                assert r.returncode == 0, r.stderr[-2000:]
                got[tag] = pickle.load(open(fout, "rb"))
            monkeypatch.delenv("BSN_LD_LUT", raising=False)
        assert "RAW" in got["raw"]["k_ld"] and "RAW" in got["raw"]["k_cl"], (got["raw"]["k_ld"], got["raw"]["k_cl"])
        assert "RAW" not in got["lut"]["k_ld"] and "k_pair_stats_f4" in got["lut"]["k_ld"] and "RAW" not in got["lut"]["k_cl"]
        for key in ("ld", "ld_sub", "clump", "clump_sub"):
            np.testing.assert_array_equal(got["raw"][key], got["lut"][key], err_msg="%s, n=%d m=%d" % (key, n, m))
        for key in ("cor", "cor_sub"):
            for a, b in zip(got["raw"][key], got["lut"][key]):
                np.testing.assert_array_equal(a, b, err_msg="%s, n=%d m=%d" % (key, n, m))
        if n <= 420:      # (the oracle's scalar loops: seconds at these sizes)
            np.testing.assert_allclose(got["raw"]["ld"], orc.ld_scores(ob, size=500, infos_pos=pos), rtol=1e-12)
            np.testing.assert_array_equal(got["raw"]["clump"], orc.bed_clumping(ob, chrom, pos, thr_r2=0.02, size=500))
