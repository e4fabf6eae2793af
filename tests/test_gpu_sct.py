"""GPU parity of snp_grid_clumping / snp_grid_PRS against the oracle's
restatement of R/SCT.R + src/clumping-cached.cpp, plus the properties of
tests/testthat/test-6-SCT.R.  Bars: kept indices bit-exact (the reference test uses
`identical`); scores within 1e-7 relative for type = "double" (test-6-SCT.R:104-113), float32
rounding for type = "float"."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


@pytest.fixture(scope="module")
def setup(ba, orc, example_bed, golden_dir):
    Go = orc.fbm_from_bed(example_bed)
    G = ba.FBM_code256(Go.bytes)
    rng = np.random.default_rng(6)
    CHR = np.repeat([1, 2], [2542, 2000])
    POS = orc.read_bim(os.path.join(golden_dir, "example.bed"))[1]
    lpval = -np.log10(rng.uniform(size=Go.m))
    betas = rng.normal(0, 0.1, Go.m)
    return G, Go, CHR, POS, lpval, betas, rng


KW = dict(grid_thr_r2=(0.05, 0.2, 0.8), grid_base_size=(100, 200))


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for u, v in zip(x, y):
            np.testing.assert_array_equal(u, v)


def test_grid_clumping_matches_oracle_and_plain_clumping(ba, orc, setup):
    """test-6-SCT.R:32-48"""
    G, Go, CHR, POS, lpval, _, _ = setup
    with pytest.raises(ValueError, match="'pos.chr' is not sorted."):
        ba.snp_grid_clumping(G, CHR, POS[::-1], lpval)
    res = ba.snp_grid_clumping(G, CHR, POS, lpval, **KW)
    ref, grid, _ = orc.snp_grid_clumping(Go, CHR, POS, lpval, **KW)
    _same(res, ref)
    for k in grid:
        np.testing.assert_array_equal(res.grid[k], grid[k])
    assert list(res.names) == [1, 2]
    for i in range(6):
        plain = ba.snp_clumping(G, CHR, S=lpval, thr_r2=grid["thr_r2"][i], size=grid["size"][i],
                                infos_pos=POS)
        np.testing.assert_array_equal(np.concatenate([res[0][i], res[1][i]]), plain)
    # a chromosome with a single variant at the end (test-6-SCT.R:34-35)
    r3 = ba.snp_grid_clumping(G, np.r_[CHR[1:], 22], np.r_[POS[1:], 1], lpval, grid_thr_r2=0.2,
                              grid_base_size=50)
    assert len(r3) == 3 and list(r3[2][0]) == [G.ncol - 1]


def test_grid_clumping_groups_imputation_rows_exclude(ba, orc, setup):
    """test-6-SCT.R:50-86 + row subsets / exclude against the oracle"""
    G, Go, CHR, POS, lpval, _, rng = setup
    infos = rng.uniform(0.2, 1, Go.m)
    k3 = ba.snp_grid_clumping(G, CHR, POS, lpval, infos_imp=infos, grid_thr_imp=(0.3, 0.8, 0.95), **KW)
    assert k3.grid["size"].size == 18
    np.testing.assert_array_equal(k3.grid["thr_imp"], np.repeat([0.3, 0.8, 0.95], 6))
    _same(k3, orc.snp_grid_clumping(Go, CHR, POS, lpval, infos_imp=infos, grid_thr_imp=(0.3, 0.8, 0.95), **KW)[0])
    groups = [np.nonzero(infos >= t)[0] for t in (0.3, 0.8, 0.95)]
    k4 = ba.snp_grid_clumping(G, CHR, POS, lpval, groups=groups, **KW)
    _same(k4, k3)
    np.testing.assert_array_equal(k4.grid["grp_num"], np.repeat([0, 1, 2], 6))
    base = ba.snp_grid_clumping(G, CHR, POS, lpval, **KW)
    k5 = ba.snp_grid_clumping(G, CHR, POS, lpval, groups=[None, [0], np.arange(Go.m)], **KW)
    assert all(x.size == 0 for x in k5[0][:6]) and all(list(x) == [0] for x in k5[0][6:12])
    assert all(x.size == 0 for x in k5[1][:12])
    _same([k5[0][12:], k5[1][12:]], base)
    ir = np.sort(rng.choice(Go.n, 300, replace=False))
    excl = rng.choice(Go.m, 500, replace=False)
    lp_na = lpval.copy(); lp_na[rng.choice(Go.m, 50, replace=False)] = np.nan
    res = ba.snp_grid_clumping(G, CHR, POS, lp_na, ind_row=ir, exclude=excl, **KW)
    _same(res, orc.snp_grid_clumping(Go, CHR, POS, lp_na, ind_row=ir, exclude=excl, **KW)[0])


def test_grid_prs(ba, orc, setup):
    """test-6-SCT.R:90-123"""
    G, Go, CHR, POS, lpval, betas, rng = setup
    all_keep = ba.snp_grid_clumping(G, CHR, POS, lpval, **KW)
    with pytest.raises(ValueError):
        ba.snp_grid_PRS(G, all_keep, betas, lpval, type="integer")
    n_thr = int(rng.integers(10, 31))
    mp = ba.snp_grid_PRS(G, all_keep, betas, lpval, type="double", n_thr_lpS=n_thr)
    assert mp.dtype == np.float64 and mp.shape == (Go.n, n_thr * 12)
    ref = orc.snp_grid_PRS(Go, all_keep, betas, lpval, mp.grid_lpS_thr)
    np.testing.assert_allclose(mp[:], ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    # unsorted thresholds with a duplicate, float output, row subset
    thr = np.array([3.0, 0.0, 5.0, 1.0, 1.0, 2.0, 4.0])
    ir = np.sort(rng.choice(Go.n, 200, replace=False))
    mf = ba.snp_grid_PRS(G, all_keep, betas, lpval, grid_lpS_thr=thr, ind_row=ir)
    assert mf.dtype == np.float32 and mf.shape == (200, 7 * 12)
    ref = orc.snp_grid_PRS(Go, all_keep, betas, lpval, thr, ind_row=ir)
    np.testing.assert_allclose(mf[:], ref, rtol=0, atol=2e-7 * np.abs(ref).max())
    # the same numbers as one snp_PRS per set (test-6-SCT.R:115-120), and chromosome blocks add up
    # to the whole-genome C+T score of the same grid row (test-6-SCT.R:104-113)
    md = ba.snp_grid_PRS(G, all_keep, betas, lpval, grid_lpS_thr=np.arange(6.0), type="double")
    sets = all_keep[0] + all_keep[1]
    one = np.hstack([ba.snp_PRS(G, betas[k], ind_keep=k, lpS_keep=lpval[k], thr_list=np.arange(6.0))
                     for k in sets])
    np.testing.assert_allclose(md[:], one, rtol=0, atol=1e-7 * np.abs(one).max())
    for i in range(6):
        k = np.concatenate([all_keep[0][i], all_keep[1][i]])
        whole = ba.snp_PRS(G, betas[k], ind_keep=k, lpS_keep=lpval[k], thr_list=np.arange(6.0))
        np.testing.assert_allclose(md[:, 6 * i:6 * i + 6] + md[:, 36 + 6 * i:36 + 6 * i + 6], whole,
                                   rtol=0, atol=1e-7 * np.abs(whole).max())
    # NA in lpS for variants that are in no set (test-6-SCT.R:135-140)
    lp2 = lpval.copy(); lp2[:100] = np.nan
    ak = [[np.setdiff1d(k, np.arange(100)) for k in all_keep[0]], all_keep[1]]
    m2 = ba.snp_grid_PRS(G, ak, betas, lp2)
    assert np.all(np.isfinite(m2[:]))
    # empty sets give zero columns
    m3 = ba.snp_grid_PRS(G, [[np.zeros(0, dtype=np.int64), all_keep[0][0]]], betas, lpval,
                         grid_lpS_thr=[1.0, 2.0], type="double")
    assert np.all(m3[:, :2] == 0) and np.any(m3[:, 2:] != 0)
