"""Size-independent properties at BASELINE.json's full sizes (the oracle cannot run there):
C2 = 50K x 200K through the FBM-style products, C3 = 400K x 1M bed_randomSVD k = 20, C5 = windowed LD on
one chromosome of 400K x 100K.
The matrix is generated in HBM; each test takes a few seconds on MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def test_c2_products_linearity_adjointness_counts(ba):
    n, m = 50000, 200000
    gb = ba.bed.synthetic(n, m, seed=77)
    counts = ba.bed_counts(gb)
    assert np.all(counts.sum(0) == n) and counts.min() >= 0           # checksum of checksums
    st = ba.bed_colstats(gb)
    np.testing.assert_array_equal(st["sumX"], counts[1] + 2.0 * counts[2])
    np.testing.assert_array_equal(st["nb_nona_col"], n - counts[3])
    sc = ba.bed_scaleBinom(gb)
    rng = np.random.default_rng(0)
    x1, x2, y = rng.normal(size=m), rng.normal(size=m), rng.normal(size=n)
    kw = dict(center=sc["center"], scale=sc["scale"])
    a1, a2 = ba.bed_prodVec(gb, x1, **kw), ba.bed_prodVec(gb, x2, **kw)
    a3 = ba.bed_prodVec(gb, 3 * x1 - x2, **kw)
    assert np.abs(a3 - (3 * a1 - a2)).max() <= 1e-9 * np.abs(a3).max()
    z = ba.bed_cprodVec(gb, y, **kw)
    assert abs(a1 @ y - x1 @ z) <= 1e-10 * np.linalg.norm(a1) * np.linalg.norm(y)
    # mean-imputed scaling => every scaled column sums to 0:  A~' 1 = 0
    z1 = ba.bed_cprodVec(gb, np.ones(n), **kw)
    assert np.abs(z1).max() <= 1e-7 * np.sqrt(n)
    # a column subset of the product equals the product on the subset (gather paths)
    ic = np.sort(rng.choice(m, 5000, replace=False))
    np.testing.assert_allclose(ba.bed_cprodVec(gb, y, ind_col=ic, center=sc["center"][ic], scale=sc["scale"][ic]),
                               z[ic], rtol=0, atol=1e-9 * np.abs(z).max())


def test_c3_randomsvd_full_size_properties(ba):
    n, m, k = 400000, 1000000, 20
    gb = ba.bed.synthetic(n, m)
    res = ba.bed_randomSVD(gb, k=k)                                   # reference defaults (tol 1e-4)
    assert res["converged"] and res["u"].shape == (n, k) and res["v"].shape == (m, k)
    d, u, v = res["d"], res["u"], res["v"]
    assert np.all(np.diff(d) < 0) and d[-1] > 0
    np.testing.assert_allclose(u.T @ u, np.eye(k), atol=1e-6)
    np.testing.assert_allclose(v.T @ v, np.eye(k), atol=1e-4)
    assert np.abs(u.mean(0)).max() < 1e-5                            # colMeans(u) ~ 0
    kw = dict(center=res["center"], scale=res["scale"])
    # the stopping rule of the default solve (16-bit panels, residual estimate and rounding floor
    # combined in quadrature) against the TRUE eigen-residuals of A~A~' for all k pairs, through the
    # 56-bit products:  || A~ A~' u - d^2 u || <= tol d^2   (the criterion of RSpectra, tol = 1e-4)
    worst = 0.0
    for t in range(k):
        atu = ba.bed_cprodVec(gb, u[:, t], **kw)
        aatu = ba.bed_prodVec(gb, atu, **kw)
        worst = max(worst, np.linalg.norm(aatu - d[t] ** 2 * u[:, t]) / d[t] ** 2)
        assert np.linalg.norm(atu - d[t] * v[:, t]) <= 1e-10 * d[t]      # v = A~' u / d by construction
    assert worst <= 1.0e-4, worst
    av = ba.bed_prodVec(gb, v[:, k - 1], **kw)
    assert np.linalg.norm(av - d[k - 1] * u[:, k - 1]) <= 1e-3 * d[k - 1]
    # north_star: singular values within 1e-6 of the reference's — pinned at full size (where no
    # oracle can run) by a solve on 56-bit panels to tol 1e-10 with another block size
    tight = ba.bed_randomSVD(gb, k=k, tol=1e-10, slices=7, block=5)
    assert tight["converged"]
    np.testing.assert_allclose(d, tight["d"], rtol=1e-6)
    # ... and the VECTORS of the default solve against the tight solve's (round 5: precision schedule, the early block steps
    # on 24-bit panels).  What an fp64 solve stopped at the SAME tol leaves is measured, not assumed (round 6):
    # the same Krylov trajectory — same block, same start, same tol — on 56-bit panels.  The default's leading half must be
    # within 1.5 x of that solve's angle or inside north_star's 1e-6, per vector; measured at this size (two matrices):
    # default 1.56e-7 / 2.34e-7 against 1.33e-7 / 2.27e-7 for the 56-bit trajectory (u; v 8.4e-9 / 1.2e-8 against 7.2e-9 /
    # 1.2e-8) — the arithmetic is not what limits the vectors, the Lanczos process at tol 1e-4 is
    # (profiles/r06_vectors_c3.txt; 16-bit panels at every step: 2.2e-5).
    assert res["slices_max"] == 3 and 1 <= res["wide_steps"] < res["niter"]
    same = ba.bed_randomSVD(gb, k=k, slices=7, block=res["block"])          # tol 1e-4 as the default
    assert same["converged"] and same["niter"] == res["niter"]
    lam = tight["d"] ** 2
    amp = np.array([lam[i] / np.min(np.abs(lam[i] - np.delete(lam, i))) for i in range(k)])
    h = (k + 1) // 2

    def angles(x, ref):
        s = np.sign(np.sum(x * ref, axis=0))
        return np.linalg.norm(x * s - ref, axis=0)
    for name in ("u", "v"):
        a, a56 = angles(res[name], tight[name]), angles(same[name], tight[name])
        # every vector inside the Davis-Kahan bound of the solve's own residual estimate + its 16-bit floor
        assert np.all(a <= 2.0 * (res["max_rel_resid"] + 1.2 * 2.0 ** -16) * amp), (name, a)
        assert np.all(a[:h] <= np.maximum(1e-6, 1.5 * a56[:h])), (name, a[:h], a56[:h])
        assert a[:h].max() <= 1e-6, (name, a[:h])                              # north_star, at this matrix
        print("[C3 %s vs 56-bit tol-1e-10 solve] default: leading half %.2e, all %.2e; 56-bit panels on the same trajectory: "
              "%.2e, %.2e" % (name, a[:h].max(), a.max(), a56[:h].max(), a56.max()))


def test_c5_ld_window_full_size_spot_checks(ba):
    """C5: one chromosome of 400K x 100K, window of ~2 000 variants.  The oracle cannot run at
    this size; spot-check entries of the band against r computed on the host from the two
    decoded columns (pairwise-complete Pearson, src/corr.cpp:54-80), and the LD scores of a few
    columns against the sum of their r^2 (src/ld-scores.cpp:52-78)."""
    n, m, W = 400000, 100000, 2000
    gb = ba.bed.synthetic(n, m, seed=5)
    pos = np.arange(m, dtype=np.float64)
    rng = np.random.default_rng(5)
    cols = np.sort(rng.choice(np.arange(W, m - W), 3, replace=False))
    ic = np.unique(np.concatenate([np.arange(c - 40, c + 41) for c in cols]))
    # a narrow band first (cheap): all pairs within 40 variants of the sampled columns
    sub = ba.bed_cor(gb, ind_col=ic, size=40 / 1000.0, infos_pos=pos[ic], fill_diag=False)
    G = ba.read_bed(gb, np.arange(n), ic, na_val=-1).astype(np.float64)
    G[G < 0] = np.nan
    jj = np.repeat(np.arange(ic.size), np.diff(sub.p))
    pick = rng.choice(sub.i.size, 60, replace=False)
    for t in pick:
        x, y = G[:, jj[t]], G[:, sub.i[t]]
        ok = ~(np.isnan(x) | np.isnan(y))
        r = np.corrcoef(x[ok], y[ok])[0, 1]
        assert abs(sub.x[t] - r) < 1e-10
    # full-width LD scores: finite, >= 1, and not below the scores of a narrower window (which adds
    # a subset of the same non-negative terms); the narrow scores are checked exactly for two columns
    ld = ba.bed_ld_scores(gb, size=W / 1000.0, infos_pos=pos)
    assert ld.shape == (m,) and np.all(ld >= 1.0) and np.all(np.isfinite(ld))
    Wn = 150
    ldn = ba.bed_ld_scores(gb, size=Wn / 1000.0, infos_pos=pos)
    assert np.all(ld >= ldn * (1 - 1e-12)) and np.all(ldn >= 1.0)
    for c in cols[:2]:
        win = np.arange(c - Wn, c + Wn + 1)
        Gw = ba.read_bed(gb, np.arange(n), win, na_val=-1)
        x = Gw[:, Wn].astype(np.float64)
        s = 1.0
        for j in range(win.size):
            if j == Wn:
                continue
            y = Gw[:, j].astype(np.float64)
            ok = (x >= 0) & (y >= 0)
            r = np.corrcoef(x[ok], y[ok])[0, 1]
            s += r * r
        assert abs(ldn[c] - s) < 1e-8 * s
