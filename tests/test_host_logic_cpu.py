"""Host-side pieces of the mirror that need no GPU: R/corr.R:18-29 thresholds, R/SCT.R:150-154
seq_log, the grid of snp_grid_clumping (expand.grid order, R/SCT.R:54-60)."""
import numpy as np
import pytest
from scipy import stats


def test_cor_thresholds_match_qt_formula():
    from bigsnpr_amd.ld import _cor_thresholds
    n = 500
    df = np.arange(1, n + 1, dtype=np.float64) - 2
    for alpha, thr_r2 in ((1.0, 0.0), (0.05, 0.0), (0.3, 0.04), (1e-4, 0.0)):
        with np.errstate(all="ignore"):
            q = stats.t.isf(alpha / 2, df=np.where(df > 0, df, np.nan))
            ref = np.maximum(q / np.sqrt(df + q * q), np.sqrt(thr_r2))
        got = _cor_thresholds(n, alpha, thr_r2)
        np.testing.assert_array_equal(got, ref)
        # only reachable pair counts are evaluated when the missing counts allow it
        part = _cor_thresholds(n, alpha, thr_r2, n_min=n - 37)
        np.testing.assert_array_equal(part[n - 38:], ref[n - 38:])
        assert np.isnan(part[:n - 38]).all()
    # alpha = 1 keeps everything: thresholds are 0 (R: qt(0.5, df) == 0)
    assert np.all(_cor_thresholds(50, 1.0, 0.0)[2:] == 0.0)


def test_seq_log():
    """tests/testthat/test-6-SCT.R:7-15"""
    from bigsnpr_amd.sct import seq_log
    np.testing.assert_allclose(seq_log(1, 1000, 4), 10.0 ** np.arange(4))
    np.testing.assert_allclose(seq_log(1, 100, 5), 10.0 ** (np.arange(5) / 2))
    np.testing.assert_allclose(seq_log(1000, 1, 4), seq_log(1, 1000, 4)[::-1])
    np.testing.assert_allclose(seq_log(1, 1, 5), np.ones(5))
    with pytest.raises(ValueError, match="'length.out' must be a non-negative number"):
        seq_log(1, 1000, -4)


def test_chr_groups_equals_unique_and_compare():
    """the chromosome groups of snp_clumping / bed_clumping / snp_autoSVD (R/clumping.R:83-88 splits by chromosome): the
    run-boundary fast path and the general path give what `for chrom in unique(chr[keep]): which(chr == chrom & keep)` gives —
    sorted files, unsorted labels, string labels, excluded variants, a chromosome whose variants are all excluded"""
    from bigsnpr_amd.ld import chr_groups
    rng = np.random.default_rng(3)
    cases = [np.repeat(np.arange(1, 23), rng.integers(1, 400, size=22)),                    # a sorted file
             np.repeat([3, 1, 2, 10], [50, 70, 1, 40]),                                     # runs, labels not in order
             rng.integers(1, 6, size=500),                                                  # no runs at all
             np.repeat(np.array(["1", "10", "2", "X"]), [30, 20, 10, 5]),                   # strings sort as strings, like R
             np.array([7])]
    for chrs in cases:
        for keep in (None, rng.random(chrs.size) < 0.7, chrs != chrs[0]):
            sel = np.ones(chrs.size, dtype=bool) if keep is None else keep
            want = [(c, np.nonzero((chrs == c) & sel)[0]) for c in np.unique(chrs[sel])]
            got = chr_groups(chrs, keep)
            assert [g[0] for g in got] == [w[0] for w in want]
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g[1], w[1])
                assert g[1].dtype == np.int64
    assert chr_groups(np.array([], dtype=int)) == []
