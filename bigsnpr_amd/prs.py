"""snp_PRS and bed_tcrossprodSelf — host mirrors of R/PRS.R and R/bed-tcrossprodSelf.R."""
import numpy as np

from . import _lib
from ._lib import as_f64, check, f64p, i64p, ptr
from .bed import _args, assert_lengths, bed_prodVec, bed_scaleBinom
from .ld import _ind, _no_missing, _r_order_decreasing


def prodVecRev(G, betas_col, same_col, ind_row, ind_col):
    """R/PRS.R:3-7: big_prodVec(G, (2 * same - 1) * beta, ind.row, ind.col) + 2 * sum(beta[!same]).
    big_prodVec is bigstatsr's unscaled FBM product (external); here the same streaming
    kernel as bed_prodVec with center 0 / scale 1 on the FBM's 2-bit image."""
    im, ir, ic = _ind(G, ind_row, ind_col)
    _no_missing(G, "snp_PRS", ir, ic)
    betas_col = as_f64(betas_col)
    same_col = np.asarray(same_col, dtype=bool)
    if ic.size == 0:
        return np.zeros(ir.size)
    mod = (2.0 * same_col - 1.0) * betas_col
    return bed_prodVec(im, mod, ir, ic) + 2.0 * betas_col[~same_col].sum()


def snp_PRS(G, betas_keep, ind_test=None, ind_keep=None, same_keep=None, lpS_keep=None, thr_list=0):
    """R/PRS.R:36-76.  Returns an n x T matrix; T thresholds cost one pass over the kept
    columns in total (scores are accumulated from the highest threshold down)."""
    im, ind_test, ind_keep = _ind(G, ind_test, ind_keep)
    betas_keep = as_f64(np.ravel(betas_keep))
    same_keep = np.ones(ind_keep.size, dtype=bool) if same_keep is None else np.asarray(same_keep)
    if same_keep.dtype != np.bool_:
        raise TypeError("'same.keep' is not of type 'logical'.")
    assert_lengths(same_keep, ind_keep)
    assert_lengths(betas_keep, ind_keep)
    thr_arr = np.atleast_1d(np.asarray(thr_list, dtype=np.float64))
    if lpS_keep is None or (thr_arr.size == 1 and thr_arr[0] == 0):
        import warnings
        warnings.warn("'lpS.keep' or 'thr.list' was not specified. Thresholding disabled.")
        return prodVecRev(G, betas_keep, same_keep, ind_test, ind_keep)[:, None]
    lpS_keep = as_f64(np.ravel(lpS_keep))
    assert_lengths(lpS_keep, ind_keep)
    if np.any(lpS_keep < 0):
        raise ValueError("'lpS.keep' should have only positive values.")
    scores = np.full((ind_test.size, thr_arr.size), np.nan)
    ind_rem = np.arange(ind_keep.size)
    last = np.zeros(ind_test.size)
    for i in _r_order_decreasing(thr_arr):
        pass_thr = lpS_keep[ind_rem] > thr_arr[i]
        ind = ind_rem[pass_thr]
        last = last + prodVecRev(G, betas_keep[ind], same_keep[ind], ind_test, ind_keep[ind])
        scores[:, i] = last
        ind_rem = ind_rem[~pass_thr]
    return scores


def bed_tcrossprodSelf(obj_bed, fun_scaling=bed_scaleBinom, ind_row=None, ind_col=None,
                       block_size=0):
    """R/bed-tcrossprodSelf.R:21-52: returns (K, dict(center, scale)); fun.scaling is applied
    per column block there, which is the same as applying it to all columns at once."""
    ir, ic = _args(obj_bed, ind_row, ind_col)
    ms = fun_scaling(obj_bed, ind_row=ir, ind_col=ic)
    center, scale = as_f64(ms["center"]), as_f64(ms["scale"])
    if ir.size * ir.size * 8 >= (8 << 20):   # a large K lands in a page-locked block of the result pool (direct DMA)
        K = _lib.result_pool.empty((ir.size, ir.size)).T      # (symmetric: the orientation of the block is immaterial)
    else:
        K = np.empty((ir.size, ir.size), dtype=np.float64, order="F")
    check(_lib.load().bsn_bed_tcrossprod(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p),
                                         ic.size, ptr(center, f64p), ptr(scale, f64p),
                                         int(block_size), K.ctypes.data_as(f64p)))
    _lib.result_pool.kick()
    return K, dict(center=center, scale=scale)


def prod_and_rowSumsSq(obj_bed, ind_row, ind_col, center, scale, V):
    """src/bed-fun.cpp:103-133: (X V, rowSums(X^2)) for the scaled sub-matrix X"""
    ir, ic = _args(obj_bed, ind_row, ind_col)
    center, scale = as_f64(np.ravel(center)), as_f64(np.ravel(scale))
    V = np.asfortranarray(np.asarray(V, dtype=np.float64))
    if V.ndim == 1:
        V = V[:, None]
    if V.shape[0] != ic.size:
        raise ValueError("Incompatibility between dimensions.")     # myassert_size(m, V.rows())
    assert_lengths(center, ic); assert_lengths(scale, ic)
    XV = np.empty((ir.size, V.shape[1]), dtype=np.float64, order="F")
    rs = np.empty(ir.size)
    check(_lib.load().bsn_bed_prod_and_rowsumssq(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p),
                                                 ic.size, ptr(center, f64p), ptr(scale, f64p),
                                                 V.ctypes.data_as(f64p), V.shape[1],
                                                 XV.ctypes.data_as(f64p), ptr(rs, f64p)))
    return XV, rs


def pca_OADP_proj2(XV, X_norm, sval):
    """OADP_proj of R/bed-projectPCA.R:62-67 -> bigutilsr::pca_OADP_proj2 (external, not in the reference tree):
    the Online Augmentation - Decomposition - Procrustes projection of Zhang, Dey & Lee (Bioinformatics 2020),
    restated from the paper.  For one new sample with simple projection l = V'y (a row of XV) and squared norm
    |y|^2, the reference X = U D V' augmented by y' has the factorisation
        [X; y'] = [U 0; 0 1] [D 0; l' r] [V q]',      r^2 = |y|^2 - |l|^2,
    so its PCA follows from the (K+1) x (K+1) matrix  M M' = [D^2, D l; (D l)', |y|^2]: eigenvectors A, eigenvalues
    s^2.  The top-K scores of the augmented reference samples are U A[1:K, 1:K] diag(s), those of the new sample
    A[K+1, 1:K] diag(s); the similarity transformation (rotation R, scale rho) that maps the former onto the
    original scores U D only involves K x K matrices because U has orthonormal columns, and the projection is
    rho * (A[K+1, 1:K] diag(s)) R.  Checked against the brute-force definition (SVD of the augmented matrix +
    Procrustes on all reference samples) in tests/test_oadp_cpu.py; parity with bigutilsr itself is UNPINNED
    (whether its Procrustes step also translates cannot be told from the reference tree; on PC scores, whose
    columns sum to zero, a translation is zero up to rounding)."""
    XV = np.asarray(XV, dtype=np.float64)
    X_norm = np.asarray(X_norm, dtype=np.float64).ravel()
    sval = np.asarray(sval, dtype=np.float64).ravel()
    n, K = XV.shape
    if sval.size != K or X_norm.size != n:
        raise ValueError("Incompatibility between dimensions.")
    idx = np.arange(K)
    Q = np.zeros((n, K + 1, K + 1))
    Q[:, idx, idx] = sval ** 2
    Q[:, idx, K] = Q[:, K, idx] = sval * XV
    Q[:, K, K] = X_norm
    w, A = np.linalg.eigh(Q)                                   # ascending
    w, A = w[:, ::-1][:, :K], A[:, :, ::-1][:, :, :K]          # the K largest
    S = np.sqrt(np.maximum(w, 0.0))
    ref_aug = A[:, :K, :] * S[:, None, :]                      # coordinates of the reference samples in U
    new_aug = A[:, K, :] * S
    M = np.einsum("nik,i->nki", ref_aug, sval)                 # ref_aug' D
    Up, sp, Vtp = np.linalg.svd(M)
    R = Up @ Vtp
    rho = sp.sum(axis=1) / (ref_aug ** 2).sum(axis=(1, 2))
    return rho[:, None] * np.einsum("nk,nkj->nj", new_aug, R)


def bed_projectSelfPCA(obj_svd, obj_bed, ind_row, ind_col=None, ncores=1):
    """R/bed-projectPCA.R:196-227: simple projection X V of new rows on the PCs of `obj_svd`, the squared
    row norms, and the OADP projection computed from them (pca_OADP_proj2 above)."""
    ind_col = obj_svd["subset"] if ind_col is None and "subset" in obj_svd else ind_col
    if ind_col is None:
        raise ValueError("'ind.col' can't be `NULL`.")     # check_args(), test-2-pca-project.R:16-17
    ir, ic = _args(obj_bed, ind_row, ind_col)
    if np.asarray(obj_svd["v"]).shape[0] != ic.size:
        raise ValueError("Incompatibility between dimensions.")
    XV, X_norm = prod_and_rowSumsSq(obj_bed, ir, ic, obj_svd["center"], obj_svd["scale"], obj_svd["v"])
    return dict(obj_svd_ref=obj_svd, simple_proj=XV, X_norm=X_norm,
                OADP_proj=pca_OADP_proj2(XV, X_norm, obj_svd["d"]))


def prod_and_rowSumsSq2(G, ind_row, ind_col, center, scale, V):
    """src/project-utils.cpp:12-43, the FBM.code256 twin of prod_and_rowSumsSq.  The FBM accessor
    has no missing-value handling: a missing code is NA_real and poisons its whole row of XV and
    its rowSumsSq entry (src/project-utils.cpp:33-38), which is reproduced here (NaN)."""
    im, ir, ic = _ind(G, ind_row, ind_col)
    center, scale = as_f64(np.ravel(center)), as_f64(np.ravel(scale))
    V = np.asfortranarray(np.asarray(V, dtype=np.float64))
    if V.ndim == 1:
        V = V[:, None]
    if V.shape[0] != ic.size:
        raise ValueError("Incompatibility between dimensions.")     # myassert_size(m, V.rows())
    assert_lengths(center, ic); assert_lengths(scale, ic)
    XV = np.empty((ir.size, V.shape[1]), dtype=np.float64, order="F")
    rs = np.empty(ir.size)
    check(_lib.load().bsn_snp_prod_and_rowsumssq2(im.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p), ic.size,
                                                  ptr(center, f64p), ptr(scale, f64p), V.ctypes.data_as(f64p),
                                                  V.shape[1], XV.ctypes.data_as(f64p), ptr(rs, f64p)))
    return XV, rs


def snp_projectSelfPCA(obj_svd, G, ind_row, ind_col=None, ncores=1):
    """R/bed-projectPCA.R:252-281 (a row with a missing value is NaN in all three outputs, as in the reference)"""
    ind_col = obj_svd["subset"] if ind_col is None and "subset" in obj_svd else ind_col
    if ind_col is None:
        raise ValueError("'ind.col' can't be `NULL`.")
    im, ir, ic = _ind(G, ind_row, ind_col)
    assert_lengths(np.arange(np.asarray(obj_svd["v"]).shape[0]), ic)
    XV, X_norm = prod_and_rowSumsSq2(im, ir, ic, obj_svd["center"], obj_svd["scale"], obj_svd["v"])
    ok = ~(np.isnan(X_norm) | np.isnan(XV).any(axis=1))
    oadp = np.full(XV.shape, np.nan)
    if ok.any():
        oadp[ok] = pca_OADP_proj2(XV[ok], X_norm[ok], obj_svd["d"])
    return dict(obj_svd_ref=obj_svd, simple_proj=XV, OADP_proj=oadp)
