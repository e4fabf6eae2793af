"""pca_OADP_proj2 (bigsnpr_amd/prs.py) against the brute-force definition of the OADP projection (Zhang, Dey & Lee
2020): append the new sample to the reference, redo the PCA of the augmented matrix, align the augmented reference
scores to the original ones with a Procrustes similarity transformation over ALL reference samples, and apply it
to the new sample's scores.  The restated shortcut only touches (K+1) x (K+1) matrices; for a rank-K reference both
are the same numbers.  No GPU, no library: pure host arithmetic."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oadp():
    # prs.py imports the package (which needs the built library only when called): load the one function from source
    src = open(os.path.join(ROOT, "bigsnpr_amd", "prs.py")).read()
    start = src.index("def pca_OADP_proj2(")
    end = src.index("\ndef ", start + 10)
    ns = {"np": np}
    exec(src[start:end], ns)
    return ns["pca_OADP_proj2"]


def _brute(Xk, U, d, y, K):
    aug = np.vstack([Xk, y[None, :]])
    Ua, sa, _ = np.linalg.svd(aug, full_matrices=False)
    scores = Ua[:, :K] * sa[:K]
    ref_aug, new_aug = scores[:-1], scores[-1]
    target = U * d
    Up, sp, Vtp = np.linalg.svd(ref_aug.T @ target)
    R = Up @ Vtp
    rho = sp.sum() / (ref_aug ** 2).sum()
    return rho * new_aug @ R


def test_shortcut_equals_the_definition_on_a_rank_k_reference():
    f = _oadp()
    rng = np.random.default_rng(0)
    n, p, K = 60, 300, 5
    X = rng.normal(size=(n, p)) + np.outer(rng.normal(size=n), rng.normal(size=p)) * 3
    X -= X.mean(axis=0)
    U, d, Vt = np.linalg.svd(X, full_matrices=False)
    U, d, V = U[:, :K], d[:K], Vt[:K].T
    Xk = (U * d) @ V.T
    Y = rng.normal(size=(7, p)) * 1.5 + np.outer(rng.normal(size=7), Vt[0]) * 20
    XV, X_norm = Y @ V, (Y ** 2).sum(axis=1)
    got = f(XV, X_norm, d)
    for i in range(Y.shape[0]):
        np.testing.assert_allclose(got[i], _brute(Xk, U, d, Y[i], K), rtol=1e-9, atol=1e-9 * d[0])
    # a sample that lies in the span of the PCs needs no correction of its direction, and the shrinkage correction
    # only ever lengthens a projection
    assert np.all((got ** 2).sum(axis=1) >= (XV ** 2).sum(axis=1) * (1 - 1e-12))


def test_oadp_undoes_the_shrinkage_of_out_of_sample_projections():
    """what tests/testthat/test-2-pca-project.R:49-57 asserts on real data, on a synthetic three-population
    sample: simple projections of left-out individuals are shrunk towards 0, OADP projections sit closer to the
    population centres of the reference PCs"""
    f = _oadp()
    rng = np.random.default_rng(3)
    n, p, K = 240, 4000, 3
    pop = np.repeat([0, 1, 2], n // 3)
    freq = np.clip(0.3 + 0.08 * rng.normal(size=(3, p)), 0.05, 0.95)
    G = rng.binomial(2, freq[pop]).astype(float)
    train = np.zeros(n, dtype=bool)
    train[rng.choice(n, 120, replace=False)] = True
    mu = G[train].mean(axis=0)
    sd = G[train].std(axis=0) + 1e-9
    A = (G - mu) / sd
    U, d, Vt = np.linalg.svd(A[train], full_matrices=False)
    U, d, V = U[:, :K], d[:K], Vt[:K].T
    XV, X_norm = A[~train] @ V, (A[~train] ** 2).sum(axis=1)
    oadp = f(XV, X_norm, d)
    ref = np.array([np.median((U * d)[pop[train] == c][:, :2], axis=0) for c in range(3)])
    pred1 = np.array([np.median(XV[pop[~train] == c][:, :2], axis=0) for c in range(3)])
    pred2 = np.array([np.median(oadp[pop[~train] == c][:, :2], axis=0) for c in range(3)])
    assert (ref ** 2).sum() > (pred1 ** 2).sum()
    assert ((ref - pred2) ** 2).sum() < ((ref - pred1) ** 2).sum()
