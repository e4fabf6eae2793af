"""Stacked clumping + thresholding grids — host mirror of R/SCT.R.

snp_grid_clumping: the reference calls clumping_chr_cached (src/clumping-cached.cpp:11-107) once
per (thr.r2, base.size) and threads a sparse r2 cache through the calls; here one call of
bsn_clumping_chr_cached per (chromosome, thr.imp, group) computes the r2 band once on the GPU at
the largest window and sweeps every grid point over it.

snp_grid_PRS: the reference runs one snp_PRS (R/PRS.R:36-76) per clumping set; here all the sets
of a chromosome and all thresholds come out of one sweep over the union of the kept columns,
as n x C int8-MFMA GEMMs per threshold bin (bsn_snp_grid_prs).

snp_grid_stacking (R/SCT.R:278-319) is bigstatsr::big_spLogReg / big_spLinReg (external
penalised regression, not under /root/reference/src) plus index arithmetic: out of scope.
Indices are 0-based."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, check, f64p, i32p, i64p, ptr
from .bed import assert_lengths
from .ld import _ind, _r_order_decreasing, assert_sorted, snp_colstats


def seq_log(from_, to, length_out):
    """R/SCT.R:150-154: evenly spaced on a log scale"""
    if not (length_out >= 0):
        raise ValueError("'length.out' must be a non-negative number")
    return np.exp(np.linspace(np.log(from_), np.log(to), int(length_out)))


class GridKeep(list):
    """list (one entry per chromosome) of lists (one index array per grid row); `.grid` holds the
    columns size / thr_r2 / grp_num / thr_imp of attr(all_keep, "grid"), `.names` the chromosomes"""
    grid = None
    names = None


def snp_grid_clumping(G, infos_chr, infos_pos, lpS, ind_row=None,
                      grid_thr_r2=(0.01, 0.05, 0.1, 0.2, 0.5, 0.8, 0.95),
                      grid_base_size=(50, 100, 200, 500), infos_imp=None, grid_thr_imp=1,
                      groups=None, exclude=None, ncores=1):
    """R/SCT.R:32-134"""
    im, ir, _ = _ind(G, ind_row, None)
    m_all = im.ncol
    infos_chr, infos_pos, lpS = np.asarray(infos_chr), as_f64(np.ravel(infos_pos)), as_f64(np.ravel(lpS))
    infos_imp = np.ones(m_all) if infos_imp is None else as_f64(np.ravel(infos_imp))
    for v in (infos_chr, infos_pos, infos_imp, lpS):
        assert_lengths(np.arange(m_all), v)
    if groups is None:
        groups = [np.arange(m_all)]
    if not isinstance(groups, list):
        raise TypeError("'groups' is not of class 'list'.")
    THR_IMP = np.unique(np.atleast_1d(np.asarray(grid_thr_imp, dtype=np.float64)))
    THR_CLMP = np.unique(np.atleast_1d(np.asarray(grid_thr_r2, dtype=np.float64)))
    BASE = np.unique(np.atleast_1d(np.asarray(grid_base_size, dtype=np.float64)))
    # expand.grid(size, thr.r2, grp.num, thr.imp): first factor varies fastest (R/SCT.R:54-60)
    shape = (THR_IMP.size, len(groups), THR_CLMP.size, BASE.size)
    ti_, g_, tc_, bs_ = np.meshgrid(THR_IMP, np.arange(len(groups)), THR_CLMP, BASE, indexing="ij")
    grid = dict(size=(bs_ / tc_).astype(np.int64).ravel(), thr_r2=tc_.ravel(), grp_num=g_.ravel(),
                thr_imp=ti_.ravel())
    # the grid points of one clumping_chr_cached call, thr.r2 outer / base.size inner
    sizes = as_f64((1000.0 * bs_[0, 0] / tc_[0, 0]).ravel())   # in bp, R/SCT.R:125
    thrs = as_f64(tc_[0, 0].ravel())
    n_grid = sizes.size
    excl = np.zeros(m_all, dtype=bool)
    if exclude is not None and len(exclude):
        excl[np.asarray(exclude, dtype=np.int64)] = True
    lib = _lib.load()
    out = GridKeep()
    out.grid, out.names = grid, []
    for chrom in np.unique(infos_chr[~excl]):
        ind_chr = np.nonzero((infos_chr == chrom) & ~excl)[0].astype(np.int64)
        info, S, pos = infos_imp[ind_chr], lpS[ind_chr], infos_pos[ind_chr]
        st = snp_colstats(G, ir, ind_chr)
        sumX, denoX = st["sumX"], st["denoX"]
        assert_sorted(pos, "pos.chr")
        ind_keep = []
        for ti in THR_IMP:
            sel = np.nonzero(info >= ti)[0]
            ind_chr, info, pos, S = ind_chr[sel], info[sel], pos[sel], S[sel]
            sumX, denoX = sumX[sel], denoX[sel]
            for group in groups:
                grp = np.zeros(0, dtype=np.int64) if group is None else np.asarray(group, dtype=np.int64)
                ind2 = np.nonzero(np.isin(ind_chr, grp))[0]
                if ind2.size == 0:
                    ind_keep += [np.zeros(0, dtype=np.int64) for _ in range(n_grid)]
                    continue
                cols = np.ascontiguousarray(ind_chr[ind2])
                ord_ = _r_order_decreasing(S[ind2]).astype(np.int32)
                rank = np.empty(ord_.size, dtype=np.int32)
                rank[ord_] = np.arange(ord_.size, dtype=np.int32)
                pos_g, sum_g, den_g = as_f64(pos[ind2]), as_f64(sumX[ind2]), as_f64(denoX[ind2])
                keep = np.empty((n_grid, cols.size), dtype=np.int32)
                check(lib.bsn_clumping_chr_cached(
                    im.handle, ptr(ir, i64p), ir.size, ptr(cols, i64p), cols.size, 0, ptr(sum_g, f64p),
                    ptr(den_g, f64p), ptr(ord_, i32p), ptr(rank, i32p), ptr(pos_g, f64p), n_grid,
                    ptr(sizes, f64p), ptr(thrs, f64p), keep.ctypes.data_as(i32p)))
                assert np.all((keep == 0) | (keep == 1))     # R/SCT.R:129
                ind_keep += [cols[keep[g] == 1] for g in range(n_grid)]
        out.append(ind_keep)
        out.names.append(chrom)
    assert all(len(k) == int(np.prod(shape)) for k in out)
    return out


class MultiPRS:
    """The C+T score matrix of snp_grid_PRS with the attributes the reference attaches to its FBM
    (R/SCT.R:251-257): lpS, grid_lpS_thr, betas, all_keep."""

    def __init__(self, scores, lpS, grid_lpS_thr, betas, all_keep):
        self.scores, self.lpS, self.grid_lpS_thr = scores, lpS, grid_lpS_thr
        self.betas, self.all_keep = betas, all_keep

    shape = property(lambda self: self.scores.shape)
    dtype = property(lambda self: self.scores.dtype)

    def __getitem__(self, idx):
        return self.scores[idx]

    def __array__(self, dtype=None, copy=None):
        return self.scores if dtype is None else self.scores.astype(dtype)


def snp_grid_PRS(G, all_keep, betas, lpS, n_thr_lpS=50, grid_lpS_thr=None, ind_row=None,
                 type="float", ncores=1):
    """R/SCT.R:201-262.  Returns a MultiPRS (n x (number of sets * number of thresholds),
    float32 for type = "float", float64 for "double")."""
    from .ld import _no_missing
    im, ir, _ = _ind(G, ind_row, None)
    _no_missing(G, "snp_grid_PRS", ir, None)
    betas, lpS = as_f64(np.ravel(betas)), as_f64(np.ravel(lpS))
    assert_lengths(np.arange(im.ncol), betas)
    assert_lengths(np.arange(im.ncol), lpS)
    if type not in ("float", "double"):
        raise ValueError("'arg' should be one of \"float\", \"double\"")      # match.arg
    if grid_lpS_thr is None:
        grid_lpS_thr = 0.9999 * seq_log(max(0.1, np.nanmin(lpS)), np.nanmax(lpS), n_thr_lpS)
    thr = np.atleast_1d(np.asarray(grid_lpS_thr, dtype=np.float64))
    T = thr.size
    order = np.argsort(thr, kind="stable")
    thr_sorted = thr[order]
    n_sets = sum(len(k) for k in all_keep)
    scores = np.empty((ir.size, n_sets * T), dtype=np.float32 if type == "float" else np.float64,
                      order="F")
    lib = _lib.load()
    ic0 = 0
    for sets in all_keep:                 # the sets of one chromosome share one sweep
        Cn = len(sets)
        if Cn == 0:
            continue
        union = np.unique(np.concatenate([np.asarray(s, dtype=np.int64) for s in sets]
                                         + [np.zeros(0, dtype=np.int64)]))
        block = np.zeros((ir.size, Cn * T), order="F")
        if union.size:
            member = np.zeros((Cn, union.size), dtype=np.uint8)       # m x C column-major
            for c, s in enumerate(sets):
                member[c, np.searchsorted(union, np.asarray(s, dtype=np.int64))] = 1
            lp = lpS[union]
            # number of thresholds below lpS[j] (NA never passes, R/PRS.R:68)
            bins = np.where(np.isnan(lp), 0, np.searchsorted(thr_sorted, lp, side="left")).astype(np.int32)
            b_u = as_f64(betas[union])
            check(lib.bsn_snp_grid_prs(im.handle, ptr(ir, i64p), ir.size, ptr(union, i64p), union.size,
                                       ptr(b_u, f64p), ptr(bins, i32p),
                                       member.ctypes.data_as(C.POINTER(C.c_uint8)), Cn, T,
                                       4 if type == "float" else 7, block.ctypes.data_as(f64p)))
        # back to the caller's threshold order
        blk = block.reshape(ir.size, T, Cn, order="F")          # column c * T + t -> [:, t, c]
        unsorted = np.empty_like(blk)
        unsorted[:, order, :] = blk
        scores[:, ic0 * T:(ic0 + Cn) * T] = unsorted.reshape(ir.size, Cn * T, order="F")
        ic0 += Cn
    return MultiPRS(scores, lpS, thr, betas, all_keep)
