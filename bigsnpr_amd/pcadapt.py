"""multLinReg / snp_pcadapt / bed_pcadapt — host mirror of src/multLinReg.cpp and R/pcadapt.R
(SURVEY.md §8f-2).  bsn_mult_lin_reg: the sums over samples run on the GPU (plane sums of k_cprod at
56-bit fixed point + exact genotype counts); the K t-scores per variant are evaluated inside the library
with the reference's expressions.  The robust distance is the restated bigutilsr::dist_ogk of
autosvd.py (parity unpinned for that step, as for autoSVD)."""
import numpy as np

from . import _lib
from ._lib import check, f64p, i64p, ptr
from .autosvd import dist_ogk
from .bed import assert_lengths
from .ld import _ind


def multLinReg(obj, ind_row=None, ind_col=None, U=None, ncores=1):
    """src/multLinReg.cpp:8-60; returns the m x K matrix of t-scores (NaN where the reference
    returns NA_REAL: zero denominator or fewer than 2 non-missing values)."""
    im, ir, ic = _ind(obj, ind_row, ind_col)
    U = np.asarray(U, dtype=np.float64)
    if U.ndim == 1:
        U = U[:, None]
    if U.shape[0] != ir.size:
        raise ValueError("Incompatibility between dimensions.")     # myassert_size(U.nrow(), n)
    K = U.shape[1]
    U = np.asfortranarray(U)
    t = np.empty((ic.size, K), dtype=np.float64, order="F")
    check(_lib.load().bsn_mult_lin_reg(im.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p), ic.size,
                                       U.ctypes.data_as(f64p), K, t.ctypes.data_as(f64p)))
    return t


def _pcadapt0(G, U_row, ind_row, ind_col):
    """R/pcadapt.R:3-27 + snp_gc (R/man-qq-gc.R:151-165)"""
    from scipy.stats import chi2
    im, ir, ic = _ind(G, ind_row, ind_col)
    U_row = np.asarray(U_row, dtype=np.float64)
    if U_row.ndim == 1:
        U_row = U_row[:, None]
    assert_lengths(U_row, ir)
    K = U_row.shape[1]
    t = multLinReg(G, ir, ic, U_row)
    ok = ~np.isnan(t).any(axis=1)
    score = np.full(ic.size, np.nan)
    score[ok] = (t[ok, 0] - np.median(t[ok, 0])) ** 2 if K == 1 else dist_ogk(t[ok])
    lam = np.median(score[ok]) / chi2.isf(0.5, K)                     # getLambdaGC
    def predict(log10=True):
        lp = chi2.logsf(score / lam, K) / np.log(10)
        return lp if log10 else 10.0 ** lp
    return dict(score=score, lamGC=lam, df=K, predict=predict, tscores=t)


def snp_pcadapt(G, U_row, ind_row=None, ind_col=None, ncores=1):
    """R/pcadapt.R:57-64"""
    return _pcadapt0(G, U_row, ind_row, ind_col)


def bed_pcadapt(obj_bed, U_row, ind_row=None, ind_col=None, ncores=1):
    """R/pcadapt.R:70-77"""
    return _pcadapt0(obj_bed, U_row, ind_row, ind_col)
