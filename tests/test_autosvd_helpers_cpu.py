"""CPU tests of the host-side statistics that snp_autoSVD / bed_autoSVD need (restatements of
bigutilsr::dist_ogk / rollmean / tukey_mc_up, see bigsnpr_amd/autosvd.py — unpinned against
the reference, so they are checked against independent definitions)."""
import numpy as np
import pytest

from bigsnpr_amd import autosvd as A


def test_scale_tau2_is_consistent_for_normal_data():
    rng = np.random.default_rng(0)
    x = rng.normal(3.0, 2.5, size=200000)
    mu, s = A.scale_tau2(x, mu_too=True)
    assert abs(mu - 3.0) < 0.03 and abs(s / 2.5 - 1) < 0.02
    x[:2000] = 1e6                       # 1 % gross outliers barely move it
    assert abs(A.scale_tau2(x) / 2.5 - 1) < 0.05


def test_dist_ogk_matches_mahalanobis_on_clean_data_and_flags_outliers():
    rng = np.random.default_rng(1)
    L = np.array([[2.0, 0, 0], [0.8, 1.0, 0], [-0.5, 0.3, 0.7]])
    X = rng.normal(size=(20000, 3)) @ L.T + np.array([1.0, -2.0, 0.5])
    d = A.dist_ogk(X)
    Xc = X - X.mean(0)
    dm = np.einsum("ij,ij->i", Xc @ np.linalg.inv(np.cov(X, rowvar=False)), Xc)
    # the reweighted covariance (hard rejection at beta = 0.9) is not rescaled, as in
    # robustbase::covOGK: distances are a constant multiple of the classical ones
    ratio = d / dm
    assert np.corrcoef(d, dm)[0, 1] > 0.999 and ratio.std() / ratio.mean() < 0.05 and 1.0 < ratio.mean() < 1.5
    X[:50] += 25.0
    d2 = A.dist_ogk(X)
    assert d2[:50].min() > np.quantile(d2[50:], 0.999)


def test_rollmean_matches_naive_definition():
    rng = np.random.default_rng(2)
    x = rng.normal(size=300)
    for size in (0, 1, 7, 50):
        got = A.rollmean(x, size)
        if size == 0:
            assert got is x or np.array_equal(got, x)
            continue
        from scipy.stats import norm
        L = 2 * size + 1
        a = 3 / 8 if L <= 10 else 0.5
        pp = (np.arange(1, L + 1) - a) / (L + 1 - 2 * a)
        w = norm.pdf(np.linspace(norm.ppf(pp[0]), norm.ppf(pp[-1]), L))
        ref = np.empty_like(x)
        for i in range(x.size):
            lo, hi = max(0, i - size), min(x.size, i + size + 1)
            ww = w[lo - (i - size):hi - (i - size)]
            ref[i] = np.sum(x[lo:hi] * ww) / ww.sum()
        np.testing.assert_allclose(got, ref, rtol=1e-12)
    with pytest.raises(ValueError, match="too large"):
        A.rollmean(x, 150)


def test_medcouple_matches_naive():
    rng = np.random.default_rng(3)
    for n, gen in ((201, rng.normal), (500, lambda size: rng.exponential(size=size)),
                   (333, lambda size: -rng.lognormal(size=size))):
        x = gen(size=n)
        med = np.median(x)
        up, lo = x[x >= med], x[x <= med]
        with np.errstate(all="ignore"):
            H = ((up[:, None] - med) - (med - lo[None, :])) / (up[:, None] - lo[None, :])
        H[np.isnan(H)] = 0.0                       # the (med, med) pair
        assert abs(A.medcouple(x) - np.median(H)) < 1e-9
    assert A.medcouple(rng.exponential(size=5000)) > 0.2 > 0 > A.medcouple(-rng.exponential(size=5000))


def test_tukey_mc_up_behaviour():
    rng = np.random.default_rng(4)
    x = rng.normal(size=20000)
    # family-wise calibration under normality: the fence sits near qnorm(1 - alpha/m)
    from scipy.stats import norm
    thr = A.tukey_mc_up(x, alpha=0.05)
    assert abs(thr - norm.ppf(1 - 0.05 / x.size)) < 0.35
    assert (x > thr).sum() <= 1
    assert A.tukey_mc_up(x, alpha=0.999) < thr                 # larger alpha -> lower fence
    assert abs(A.tukey_mc_up(x, coef=1.5) - (np.quantile(x, .75) + 1.5 * np.subtract(*np.quantile(x, [.75, .25])))) < 0.15
    skew = rng.exponential(size=20000)
    q1, q3 = np.quantile(skew, [.25, .75])
    assert A.tukey_mc_up(skew, coef=1.5) > q3 + 1.5 * (q3 - q1)  # right-skew widens the upper fence


def test_get_intervals():
    # R: getIntervals(c(1,2,3, 7, 9,10, 20,21,22,23), n = 3) -> rows (1,3), (20,23)
    np.testing.assert_array_equal(A.getIntervals(np.array([1, 2, 3, 7, 9, 10, 20, 21, 22, 23]), n=3),
                                  [[1, 3], [20, 23]])
    np.testing.assert_array_equal(A.getIntervals(np.array([1, 2, 3, 7, 9, 10, 20, 21, 22, 23]), n=2),
                                  [[1, 3], [9, 10], [20, 23]])
    assert A.getIntervals(np.array([5]), n=2).shape == (0, 2)
    assert A.getIntervals(np.array([1, 3, 5]), n=2).shape == (0, 2)


# ---- worked values from the publications / manuals the restatements follow -------------------------
def test_medcouple_worked_examples():
    """robustbase::mc help page (the implementation bigutilsr calls): mc(1:5) is 0 for a symmetric
    sample and mc(c(1, 2, 7, 9, 10)) = -1/3 — by the kernel of Brys, Hubert & Struyf (2004), J. Comput.
    Graph. Statist. 13(4), eq. (2.2): the nine values -1, -1, -1/2, -3/7, -1/3, -1/4, 0, 1, 1 have median
    -1/3.  Ties at the median follow the paper's rule h = sign(m - 1 - i - j) (its section 2.1): on
    (1, 2, 2, 2, 3, 5, 9) the brute-force median of all 4 x 5 kernel values is 0.875."""
    assert A.medcouple(np.arange(1, 6)) == 0.0
    assert abs(A.medcouple([1, 2, 7, 9, 10]) + 1.0 / 3.0) < 1e-12
    assert abs(A.medcouple([1, 2, 2, 2, 3, 5, 9]) - 0.875) < 1e-12
    # a sample and its mirror image have opposite medcouples (the paper's property 2)
    x = np.array([0.3, 0.9, 1.4, 2.2, 2.3, 4.0, 7.5, 11.0])
    assert abs(A.medcouple(x) + A.medcouple(-x)) < 1e-12
    # location and scale invariance (property 1)
    assert abs(A.medcouple(3.0 + 2.5 * x) - A.medcouple(x)) < 1e-12


def test_adjusted_boxplot_fence_worked_example():
    """Hubert & Vandervieren (2008), Comput. Statist. Data Anal. 52, eq. (5): upper fence
    Q3 + 1.5 exp(3 MC) IQR for MC >= 0 and Q3 + 1.5 exp(4 MC) IQR for MC < 0.  On (1, 2, 7, 9, 10):
    Q1 = 2, Q3 = 9 (type-7 quantiles), MC = -1/3  ->  9 + 1.5 exp(-4/3) 7 = 11.76779..."""
    assert abs(A.tukey_mc_up([1, 2, 7, 9, 10], coef=1.5) - (9 + 10.5 * np.exp(-4.0 / 3.0))) < 1e-12
    x = [1, 2, 2, 2, 3, 5, 9]                                    # MC = 0.875 >= 0
    q1, q3 = np.quantile(x, [0.25, 0.75])
    assert abs(A.tukey_mc_up(x, coef=1.5) - (q3 + 1.5 * np.exp(3 * 0.875) * (q3 - q1))) < 1e-12


def test_tau_scale_properties_of_maronna_zamar():
    """Maronna & Zamar (2002), Technometrics 44(4), section 2: the tau-scale with c1 = 4.5, c2 = 3 is
    affine equivariant and Fisher-consistent at the normal; robustbase::scaleTau2 (consistency = TRUE)
    divides by E[rho_c2] accordingly.  On the quantile grid of N(0, 1) (no sampling noise) it returns
    1 within the discretisation error, and mu is the centre of symmetry."""
    from scipy.stats import norm
    z = norm.ppf((np.arange(1, 20002) - 0.5) / 20001)
    mu, s = A.scale_tau2(z, mu_too=True)
    assert abs(s - 1.0) < 2e-3 and abs(mu) < 1e-10
    mu2, s2 = A.scale_tau2(5.0 - 3.0 * z, mu_too=True)
    assert abs(s2 - 3.0 * s) < 1e-9 and abs(mu2 - 5.0) < 1e-9
    assert A.scale_tau2(np.array([2.0, 2.0, 2.0, 2.0, 7.0])) == 0.0     # MAD = 0 -> scale 0 (robustbase returns 0)


def test_rollmean_invariants():
    """bigutilsr::rollmean (Privé et al. 2020, Bioinformatics 36(16), 'Efficient toolkit implementing best
    practices for PCA of population genetic data', section 2.2: Gaussian-weighted rolling mean used to
    smooth the outlier statistic along the genome): weights are positive and symmetric, so constants are
    preserved, size = 0 is the identity, a linear trend is preserved away from the edges, and the output
    never leaves the range of the input."""
    x = np.linspace(-3, 5, 400)
    for size in (1, 5, 20):
        y = A.rollmean(x, size)
        np.testing.assert_allclose(y[size:-size], x[size:-size], atol=1e-12)
        assert y.min() >= x.min() - 1e-12 and y.max() <= x.max() + 1e-12
        np.testing.assert_allclose(A.rollmean(np.full(100, 2.5), size), 2.5, rtol=1e-15)
    r = np.random.default_rng(0).normal(size=100)
    assert A.rollmean(r, 0) is r or np.array_equal(A.rollmean(r, 0), r)


def test_two_independent_restatements_agree():
    """bigutilsr is not in the reference tree: the product's restatements (bigsnpr_amd/autosvd.py: bisection medcouple,
    convolution rolling mean, vectorised OGK) and the oracle's (oracle/autosvd_oracle.py: all-pairs medcouple, direct
    sums, loop-by-loop OGK), written independently from the published definitions, must give the same numbers —
    and the same intervals as the reference's getIntervals (R/autoSVD.R:4-12, which IS in the tree)."""
    from bigsnpr_amd import autosvd as prod
    from oracle import autosvd_oracle as orc_a
    rng = np.random.default_rng(12)
    for n in (3, 4, 7, 50, 301):
        for x in (rng.normal(size=n), rng.exponential(size=n), np.round(rng.normal(size=n), 1),   # ties, also at the median
                  np.r_[np.zeros(n // 2 + 1), rng.normal(size=n // 2)]):
            assert abs(prod.medcouple(x) - orc_a.medcouple(x)) < 1e-12, (n, x[:5])
    for n, size in ((40, 3), (200, 50), (200, 7.9), (30, 0), (12, 5)):
        x = rng.normal(size=n)
        np.testing.assert_allclose(prod.rollmean(x, size), orc_a.rollmean(x, size), rtol=1e-12, atol=1e-14)
    for m, k in ((300, 3), (500, 10), (120, 1)):
        U = rng.normal(size=(m, k)) * rng.uniform(0.5, 3, size=k)
        U[:7] += 6.0                                              # a few outliers
        np.testing.assert_allclose(prod.dist_ogk(U), orc_a.dist_ogk(U), rtol=1e-9)
    for alpha in (0.05, 0.999, 1e-4):
        x = rng.exponential(size=400)
        assert abs(prod.tukey_mc_up(x, alpha=alpha) - orc_a.tukey_mc_up(x, alpha=alpha)) < 1e-10
    for x, n in (([1, 2, 3, 7, 8, 12, 13, 14, 15], 2), ([1, 2, 3, 7, 8, 12, 13, 14, 15], 3), ([5], 2), ([4, 9], 1),
                 (list(range(10, 40)) + [50, 51], 20), ([2, 3], 2)):
        got = [tuple(r) for r in prod.getIntervals(np.asarray(x), n=n).tolist()]
        assert got == orc_a.get_intervals(x, n=n), (x, n)


def test_scaled_up_paths_give_the_same_numbers():
    """what only large inputs reach: the robust scales of dist_ogk on several threads (same values as one after the
    other), and the medcouple's windowed rank selection deep into a bisection (against the all-pairs definition)"""
    from bigsnpr_amd import autosvd as prod
    from oracle import autosvd_oracle as orc_a
    rng = np.random.default_rng(77)
    U = rng.normal(size=(25000, 8)) * rng.uniform(0.5, 3, size=8)
    U[:40] += 5.0
    threaded = prod.dist_ogk(U)                                   # 25 000 x 8 >= the threshold of the threaded path
    keep = prod._map_threads
    try:
        prod._map_threads = lambda f, items: [f(t) for t in items]
        serial = prod.dist_ogk(U)
    finally:
        prod._map_threads = keep
    np.testing.assert_array_equal(threaded, serial)
    for x in (rng.normal(size=1500) ** 2, np.round(rng.gamma(2.0, size=1201), 1)):
        assert abs(prod.medcouple(x) - orc_a.medcouple(x)) < 1e-12


def test_quantile_of_a_sorted_vector_is_numpys_bit_for_bit():
    """tukey_mc_up's device path sorts once and reads the quartiles off the sorted vector (round 6): numpy's default
    method (R's type 7) restated, its virtual index and two-sided interpolation included"""
    from bigsnpr_amd import autosvd as prod
    rng = np.random.default_rng(2)
    for _ in range(4000):
        n = int(rng.integers(1, 400))
        x = np.sort(rng.normal(size=n) * 10.0 ** int(rng.integers(-3, 4)))
        q = float(rng.choice([0.25, 0.75, 0.5, 0.0, 1.0, rng.random()]))
        assert prod._quantile_sorted(x, q) == float(np.quantile(x, q))
    for n in (1000000, 999999, 524288):
        x = np.sort(rng.normal(size=n))
        for q in (0.25, 0.75):
            assert prod._quantile_sorted(x, q) == float(np.quantile(x, q))


def test_rollmean_in_groups_on_the_host():
    """rollmean_groups (host path): the rolling mean inside every chromosome, whatever the order of the labels; the
    denominators taken from partial sums of the weights equal the convolution of ones they replace"""
    from scipy.stats import norm
    from bigsnpr_amd import autosvd as prod
    from bigsnpr_amd.ld import chr_groups
    rng = np.random.default_rng(3)

    def by_convolution(x, size):
        half = int(np.floor(size)); length = 2 * half + 1
        a = 3.0 / 8 if length <= 10 else 0.5
        pp = (np.arange(1, length + 1) - a) / (length + 1 - 2 * a)
        lims = norm.ppf([pp[0], pp[-1]])
        w = norm.pdf(np.linspace(lims[0], lims[1], length))
        return (np.convolve(x, w[::-1], mode="full")[half:half + x.size] /
                np.convolve(np.ones_like(x), w[::-1], mode="full")[half:half + x.size])

    for n, size in ((50, 3), (1000, 50), (102, 50), (12, 5.5), (7, 1)):
        x = rng.normal(size=n)
        np.testing.assert_allclose(prod.rollmean(x, size), by_convolution(x, size), rtol=1e-14, atol=1e-16)
    chrom = np.repeat([1, 2, 3, 4], [300, 120, 500, 200])
    for labels in (chrom, rng.permutation(chrom)):
        x = rng.lognormal(size=labels.size)
        got = prod.rollmean_groups(x, 50, chr_groups(labels))
        for c in np.unique(labels):
            np.testing.assert_array_equal(got[labels == c], prod.rollmean(x[labels == c], 50))
    with pytest.raises(ValueError, match="too large"):
        prod.rollmean_groups(np.arange(300.0), 50, chr_groups(np.repeat([1, 2, 3], 100)))
