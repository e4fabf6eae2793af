"""The oracle's restatement of R/SCT.R + src/clumping-cached.cpp, pinned by the properties the
reference's own test asserts (tests/testthat/test-6-SCT.R:7-15,37-86): every grid row equals a
plain snp_clumping with the same thr.r2 / size, imputation thresholds and groups are
interchangeable, empty groups give empty sets, and the r2 cache is actually reused."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def setup(orc, example_bed, golden_dir):
    G = orc.fbm_from_bed(example_bed)
    rng = np.random.default_rng(6)
    CHR = np.repeat([1, 2], [2542, 2000])
    import os
    POS = orc.read_bim(os.path.join(golden_dir, "example.bed"))[1]
    lpval = -np.log10(rng.uniform(size=G.m))
    return G, CHR, POS, lpval, rng


def test_seq_log(orc):
    """test-6-SCT.R:7-15"""
    np.testing.assert_allclose(orc.seq_log(1, 1000, 4), 10.0 ** np.arange(4))
    np.testing.assert_allclose(orc.seq_log(1, 100, 5), 10.0 ** (np.arange(5) / 2))
    np.testing.assert_allclose(orc.seq_log(1000, 1, 4), orc.seq_log(1, 1000, 4)[::-1])
    np.testing.assert_allclose(orc.seq_log(1, 1, 5), np.ones(5))
    with pytest.raises(ValueError, match="'length.out' must be a non-negative number"):
        orc.seq_log(1, 1000, -4)


def test_grid_rows_equal_plain_clumping(orc, setup):
    """test-6-SCT.R:37-48"""
    G, CHR, POS, lpval, _ = setup
    with pytest.raises(ValueError, match="'pos.chr' is not sorted."):
        orc.snp_grid_clumping(G, CHR, POS[::-1], lpval)
    all_keep, grid, computed = orc.snp_grid_clumping(G, CHR, POS, lpval, grid_thr_r2=(0.05, 0.2, 0.8),
                                                     grid_base_size=(100, 200))
    assert len(all_keep) == 2 and all(len(k) == 6 for k in all_keep)
    np.testing.assert_array_equal(grid["size"], [2000, 4000, 500, 1000, 125, 250])
    total_pairs = 0
    for i in range(6):
        ref = orc.snp_clumping(G, CHR, S=lpval, thr_r2=grid["thr_r2"][i], size=grid["size"][i],
                               infos_pos=POS)
        np.testing.assert_array_equal(np.concatenate([all_keep[0][i], all_keep[1][i]]), ref)
    # the cache: a second identical grid recomputes nothing new beyond exact zeros
    _, _, computed_one = orc.snp_grid_clumping(G, CHR, POS, lpval, grid_thr_r2=(0.05,),
                                               grid_base_size=(200,))
    assert computed < 6 * computed_one


def test_groups_and_imputation_thresholds(orc, setup):
    """test-6-SCT.R:50-86"""
    G, CHR, POS, lpval, rng = setup
    kw = dict(grid_thr_r2=(0.05, 0.2, 0.8), grid_base_size=(100, 200))
    all_keep, _, _ = orc.snp_grid_clumping(G, CHR, POS, lpval, **kw)
    infos = rng.uniform(0.2, 1, G.m)
    k3, g3, _ = orc.snp_grid_clumping(G, CHR, POS, lpval, infos_imp=infos, grid_thr_imp=(0.3, 0.8, 0.95), **kw)
    assert len(g3["size"]) == 18
    np.testing.assert_array_equal(g3["thr_imp"], np.repeat([0.3, 0.8, 0.95], 6))
    groups = [np.nonzero(infos >= t)[0] for t in (0.3, 0.8, 0.95)]
    k4, g4, _ = orc.snp_grid_clumping(G, CHR, POS, lpval, groups=groups, **kw)
    np.testing.assert_array_equal(g4["grp_num"], np.repeat([0, 1, 2], 6))
    for a, b in zip(k3, k4):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    k5, _, _ = orc.snp_grid_clumping(G, CHR, POS, lpval, groups=[[], [0], np.arange(G.m)], **kw)
    assert all(x.size == 0 for x in k5[0][:6]) and all(list(x) == [0] for x in k5[0][6:12])
    assert all(x.size == 0 for x in k5[1][:12])
    for x, y in zip(k5[0][12:] + k5[1][12:], all_keep[0] + all_keep[1]):
        np.testing.assert_array_equal(x, y)
