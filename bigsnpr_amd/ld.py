"""Windowed LD on the GPU: snp_cor / bed_cor, snp_ld_scores / bed_ld_scores,
snp_clumping / bed_clumping — host mirror of R/corr.R, R/ld-scores.R, R/clumping.R,
R/bed-clumping.R over bsn_cormat / bsn_ld_scores / bsn_clumping_chr.

`Gna` / `G` may be a `bed` (PLINK file on the device) or an `FBM_code256` (byte matrix with
the CODE_012 coding, repacked to the same 2-bit image), the dispatch of src/corr.cpp:113-125.
Indices are 0-based.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, check, f64p, i32p, i64p, ptr, vp
from .bed import (_check_ind, assert_lengths, bed, bed_colstats, cols_along, rows_along)

CODE_012 = np.array([0, 1, 2] + [np.nan] * 253)  # R/bigSNP-class.R:7
CODE_DOSAGE = np.array([0, 1, 2, np.nan, 0, 1, 2] + list(np.round(np.arange(201) * 0.01, 2)) + [np.nan] * 48)  # :13
CODE_IMPUTE_PRED = np.array([0, 1, 2, np.nan, 0, 1, 2] + [np.nan] * 249)  # R/bigSNP-class.R:10


class FBM_code256:
    """bigstatsr's FBM.code256 (n x m bytes + a 256-entry decode table) held on the device
    (bsn_fbm_open): tables that decode to genotype calls (CODE_012, CODE_IMPUTE_PRED) become the 2-bit
    image and support every snp_* function; tables on a regular grid (CODE_DOSAGE) become a byte image
    that supports snp_colstats / snp_MAF / snp_scale*, big_prodVec / big_cprodVec / snp_PRS,
    big_randomSVD, snp_cor / snp_ld_scores (pairwise complete, like the reference) and snp_clumping;
    anything else is refused by the library."""

    def __init__(self, bytes_nm, code=CODE_012):
        a = np.asfortranarray(np.asarray(bytes_nm, dtype=np.uint8))
        if a.ndim != 2:
            raise ValueError("a genotype FBM is a matrix")
        code = np.ascontiguousarray(code, dtype=np.float64)
        if code.size != 256:
            raise ValueError("'code256' must have 256 values")
        self.code256 = code
        self.nrow, self.ncol = a.shape
        h = vp()
        check(_lib.load().bsn_fbm_open(a.ctypes.data_as(_lib.u8p), self.nrow, self.ncol, self.nrow,
                                       ptr(code, f64p), C.byref(h)))
        self._bed = bed(_handle=h, _n=self.nrow, _m=self.ncol)
        self.bits = int(_lib.load().bsn_bed_bits(h))
        self._has_na = int(_lib.load().bsn_bed_na_known(h)) != 0   # counted on the device at creation

    @property
    def handle(self):
        return self._bed.handle

    @property
    def shape(self):
        return (self.nrow, self.ncol)


def snp_fake(n, m):
    """R/fake.R:27-54: the skeleton of a bigSNP — an n x m genotype byte matrix filled with the
    missing code 3 (CODE_012), `fam` and `map` with the reference's default columns.  Fill
    ``obj["genotypes"]`` and hand it to ``FBM_code256`` to put it on the device."""
    seq_m = np.arange(1, m + 1)
    return dict(genotypes=np.full((n, m), 3, dtype=np.uint8),
                fam=dict(family_ID=["fam_%d" % i for i in range(1, n + 1)],
                         sample_ID=["ind_%d" % i for i in range(1, n + 1)],
                         paternal_ID=np.zeros(n, dtype=np.int64), maternal_ID=np.zeros(n, dtype=np.int64),
                         sex=np.zeros(n, dtype=np.int64), affection=np.full(n, -9, dtype=np.int64)),
                map=dict(chromosome=np.ones(m, dtype=np.int64), marker_ID=["snp_%d" % j for j in seq_m],
                         genetic_dist=np.zeros(m, dtype=np.int64), physical_pos=seq_m * 1000,
                         allele1=["A"] * m, allele2=["T"] * m))


def _no_missing(G, what, ind_row=None, ind_col=None):
    """bigstatsr's FBM products have no missing-value handling: a missing code decodes to NA_real and
    turns every result it touches into NA.  Rather than return such a vector (or, worse, a finite one
    that silently took the value for 0), the GPU path refuses — when the SELECTED rows and columns hold a
    missing value: the reference returns finite results for a complete sub-matrix of an FBM that has missing
    values elsewhere (callers exclude such columns through ind.col), and so does this."""
    if not (isinstance(G, FBM_code256) and G._has_na):
        return
    # A missing value poisons the sums of its column (src/colstats.cpp:14-27): one statistics pass over ALL columns for
    # this row selection says which columns are touched; the answer is kept on the FBM object (an FBM is read-only
    # here), so the loops of snp_PRS over thresholds or of snp_autoSVD over rounds pay for it once, not per call.
    import hashlib
    key = "all" if ind_row is None else hashlib.sha1(np.ascontiguousarray(ind_row, dtype=np.int64).tobytes()).hexdigest()
    cache = G.__dict__.setdefault("_na_cols_cache", {})
    if key not in cache:
        if len(cache) >= 4:
            cache.pop(next(iter(cache)))
        cache[key] = np.isnan(snp_colstats(G, ind_row, None)["sumX"])
    if ind_col is None:
        bad = cache[key]
    else:
        ic = np.asarray(ind_col, dtype=np.int64)
        if ic.size and (ic.min() < 0 or ic.max() >= cache[key].size):   # (numpy would wrap a negative index silently)
            raise IndexError("Tested %d < %d. Subscript out of bounds (ind.col)."
                             % (int(ic.max() if ic.max() >= cache[key].size else ic.min()), cache[key].size))
        bad = cache[key][ic]
    if bad.any():
        raise ValueError("%s: the selected rows and columns of the FBM hold missing values (bigstatsr would return "
                         "NA); impute first (snp_fastImputeSimple) or exclude them." % what)


def _image(obj):
    if isinstance(obj, FBM_code256):
        return obj._bed
    if isinstance(obj, bed):
        return obj
    raise TypeError("'Gna' is not of class 'FBM.code256' (or 'bed').")


def _ind(obj, ind_row, ind_col):
    im = _image(obj)
    ir = rows_along(im) if ind_row is None else _check_ind("ind.row", ind_row, im.nrow)
    ic = cols_along(im) if ind_col is None else _check_ind("ind.col", ind_col, im.ncol)
    return im, ir, ic


def assert_sorted(x, name="infos.pos"):
    if np.any(np.diff(x) < 0):
        raise ValueError("'%s' is not sorted." % name)


def _cor_thresholds(n, alpha, thr_r2, n_min=1):
    """R/corr.R:18-23,29: |r| threshold per number of non-missing pairs, thr[nona - 1].
    qt() costs 1 us per value in scipy, so only the entries a pair can actually reach
    (nona >= n_min, from the missing counts of the selected variants) are evaluated; the
    others are never read by the kernel and are left NaN."""
    THR = np.full(n, np.nan)
    lo = max(int(n_min), 1)
    df = np.arange(lo, n + 1, dtype=np.float64) - 2
    if alpha >= 1:
        # the default of snp_cor: qt(0.5, df) is exactly 0, so the threshold is 0 wherever it is defined (df > 0) —
        # no quantile evaluation at all (0.4 s for 400 000 values otherwise)
        THR[lo - 1:] = np.where(df > 0, 0.0, np.nan)
    else:
        from scipy import special
        with np.errstate(all="ignore"):
            q = -special.stdtrit(np.where(df > 0, df, np.nan), alpha / 2)   # == stats.t.isf(alpha / 2, df)
            THR[lo - 1:] = q / np.sqrt(df + q * q)
    return np.maximum(THR, np.sqrt(thr_r2))


class CorResult:
    """upper-triangular dsCMatrix slots (R/corr.R:43-47): i, p, x, Dim"""

    def __init__(self, i, p, x, m):
        self.i, self.p, self.x, self.Dim = i, p, x, (m, m)
        self.uplo = "U"

    def tocsc(self):
        from scipy import sparse
        return sparse.csc_matrix((self.x, self.i, self.p), shape=self.Dim)


def _cor0(obj, ind_row, ind_col, size, alpha, thr_r2, fill_diag, infos_pos):
    im, ir, ic = _ind(obj, ind_row, ind_col)
    pos = 1000.0 * np.arange(1, ic.size + 1) if infos_pos is None else as_f64(np.ravel(infos_pos))
    assert_lengths(pos, ic)
    assert_sorted(pos)
    # a pair of variants shares at least n - na_x - na_y samples
    from .bed import bed_counts
    if int(_lib.load().bsn_bed_bits(im.handle)) == 8:
        # a byte (dosage) image keeps no per-variant counts on the host: without missing values every pair has all
        # n samples; with some, every threshold a pair could reach is evaluated
        na = np.zeros(ic.size, dtype=np.int64)
        if getattr(obj, "_has_na", False):
            na[:] = ir.size // 2
    else:
        na = bed_counts(im, ir, ic)[3].astype(np.int64)
    top2 = np.sort(na)[-2:].sum() if na.size > 1 else int(na.sum())
    thr = as_f64(_cor_thresholds(ir.size, alpha, thr_r2, n_min=ir.size - int(top2)))
    p = np.empty(ic.size + 1, dtype=np.int32)
    nnz = C.c_int64(0)
    h = vp()
    L = _lib.load()
    check(L.bsn_cormat(im.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p), ic.size, float(size) * 1000.0,
                       ptr(thr, f64p), ptr(pos, f64p), int(bool(fill_diag)), ptr(p, i32p),
                       C.byref(nnz), C.byref(h)))
    try:
        # a large result goes to page-locked blocks of the library's result pool, like u and v of the SVD: written by
        # the DMA engines directly (no staging copy, no first-touch page faults on 12 bytes per pair)
        alloc = _lib.result_pool.empty if nnz.value * 12 >= (8 << 20) else np.empty
        i = alloc(nnz.value, dtype=np.int32)
        x = alloc(nnz.value, dtype=np.float64)
        check(L.bsn_cormat_fetch(h, ptr(i, i32p), ptr(x, f64p)))
        _lib.result_pool.kick()
        has_nan = C.c_int(0)      # noted by the kernel that wrote x (no pass over nnz doubles on the host)
        check(L.bsn_cormat_has_nan(h, C.byref(has_nan)))
    finally:
        L.bsn_cormat_free(h)
    if has_nan.value:
        import warnings
        warnings.warn("NA or NaN values in the resulting correlation matrix.")  # R/corr.R:53-54
    return CorResult(i, p, x, ic.size)


def snp_cor(Gna, ind_row=None, ind_col=None, size=500, alpha=1, thr_r2=0, fill_diag=True,
            infos_pos=None, ncores=1):
    """R/corr.R:95-110"""
    return _cor0(Gna, ind_row, ind_col, size, alpha, thr_r2, fill_diag, infos_pos)


def bed_cor(obj_bed, ind_row=None, ind_col=None, size=500, alpha=1, thr_r2=0, fill_diag=True,
            infos_pos=None, ncores=1):
    """R/corr.R:116-132"""
    return _cor0(obj_bed, ind_row, ind_col, size, alpha, thr_r2, fill_diag, infos_pos)


def _ld0(obj, ind_row, ind_col, size, infos_pos):
    im, ir, ic = _ind(obj, ind_row, ind_col)
    pos = 1000.0 * np.arange(1, ic.size + 1) if infos_pos is None else as_f64(np.ravel(infos_pos))
    assert_lengths(pos, ic)
    assert_sorted(pos)
    out = np.empty(ic.size)
    check(_lib.load().bsn_ld_scores(im.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p), ic.size,
                                    float(size) * 1000.0, ptr(pos, f64p), ptr(out, f64p)))
    return out


def snp_ld_scores(Gna, ind_row=None, ind_col=None, size=500, infos_pos=None, ncores=1):
    """R/ld-scores.R:41-53"""
    return _ld0(Gna, ind_row, ind_col, size, infos_pos)


def bed_ld_scores(obj_bed, ind_row=None, ind_col=None, size=500, infos_pos=None, ncores=1):
    """R/ld-scores.R:59-72"""
    return _ld0(obj_bed, ind_row, ind_col, size, infos_pos)


def _r_order_decreasing(S):
    """R's order(S, decreasing = TRUE): stable w.r.t. the original index"""
    return np.argsort(-np.asarray(S, dtype=np.float64), kind="stable")


def snp_colstats(G, ind_row=None, ind_col=None, ncores=1):
    """src/colstats.cpp:8-35 on the device image of an FBM.code256"""
    im, ir, ic = _ind(G, ind_row, ind_col)
    sumX, denoX = np.empty(ic.size), np.empty(ic.size)
    check(_lib.load().bsn_snp_colstats(im.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p), ic.size,
                                       ptr(sumX, f64p), ptr(denoX, f64p)))
    return dict(sumX=sumX, denoX=denoX)


def chr_groups(infos_chr, keep=None):
    """[(label, indices)] for every chromosome label among the kept entries, labels in sorted order like R's
    unique / split on a sorted vector.  The labels of a genotype file come in runs (sorted by chromosome), for which the
    groups are found from the run boundaries — np.unique and one full-length comparison per chromosome cost 0.1 s per call at
    a million variants (round 6: what the 'rest of the host loop' of snp_autoSVD was); any other order falls back to them."""
    infos_chr = np.asarray(infos_chr)
    n = infos_chr.size
    if n == 0:
        return []
    change = np.flatnonzero(infos_chr[1:] != infos_chr[:-1]) + 1
    starts = np.r_[0, change]
    labels = infos_chr[starts]
    if np.unique(labels).size == labels.size:          # runs: one run per label
        order = np.argsort(labels, kind="stable")
        ends = np.r_[change, n]
        out = []
        for t in order:
            idx = np.arange(starts[t], ends[t], dtype=np.int64)
            if keep is not None:
                idx = idx[keep[starts[t]:ends[t]]]
            if idx.size:
                out.append((labels[t], idx))
        return out
    sel = np.ones(n, dtype=bool) if keep is None else keep
    return [(c, np.nonzero((infos_chr == c) & sel)[0].astype(np.int64)) for c in np.unique(infos_chr[sel])]


def _order_and_rank(S_chr):
    ord_ = _r_order_decreasing(S_chr).astype(np.int32)
    rank = np.empty(ord_.size, dtype=np.int32)
    rank[ord_] = np.arange(ord_.size, dtype=np.int32)
    return ord_, rank


class _OrdersAhead:
    """order(S, decreasing = TRUE) of every chromosome BEFORE the loop that consumes them, in one device call
    (bsn_order_decreasing: two stable radix sorts over the concatenated statistics): the stable host sort of a
    chromosome's statistic (3 - 4 ms for 45 000 variants) otherwise sits between two GPU calls — 22 of them were a third
    of snp_clumping's wall time at a million variants.  Statistics with NaN (R puts NA last) take the host sort."""

    def __init__(self, stats):
        self.stats = [np.ascontiguousarray(s, dtype=np.float64) for s in stats]
        self.ord = self.rank = None
        total = sum(s.size for s in self.stats)
        if total >= 4096 and not any(np.isnan(s).any() for s in self.stats):
            allS = np.concatenate(self.stats)
            self.off = np.concatenate([[0], np.cumsum([s.size for s in self.stats])]).astype(np.int64)
            self.ord, self.rank = np.empty(total, dtype=np.int32), np.empty(total, dtype=np.int32)
            check(_lib.load().bsn_order_decreasing(ptr(allS, f64p), total, self.off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                   len(self.stats), ptr(self.ord, i32p), ptr(self.rank, i32p)))

    def get(self, t):
        if self.ord is None:
            return _order_and_rank(self.stats[t])
        a, b = int(self.off[t]), int(self.off[t + 1])
        return self.ord[a:b], self.rank[a:b]


def _clump_chr(im, ir, ind_chr, mode, aux1, aux2, S_chr, pos_chr, size, thr_r2, order=None):
    ord_, rank = _order_and_rank(S_chr) if order is None else order
    keep = np.empty(ind_chr.size, dtype=np.int32)
    aux1, aux2, pos_chr = as_f64(aux1), as_f64(aux2), as_f64(pos_chr)
    assert_sorted(pos_chr)
    check(_lib.load().bsn_clumping_chr(im.handle, ptr(ir, i64p), ir.size, ptr(ind_chr, i64p),
                                       ind_chr.size, mode, ptr(aux1, f64p), ptr(aux2, f64p),
                                       ptr(ord_, i32p), ptr(rank, i32p), ptr(pos_chr, f64p),
                                       float(size), float(thr_r2), ptr(keep, i32p)))
    assert np.all((keep == 0) | (keep == 1))  # stopifnot(all(keep[] %in% 0:1)), R/clumping.R:134
    return ind_chr[keep == 1]


def _clump_jobs(im, ir, jobs, mode, thr_r2, one_call=None):
    """The chromosomes of a clumping, each (ind_chr, aux1, aux2, S_chr, pos_chr, size): R/clumping.R:95-137 loops over them.
    On a resident handle with integer positions they go to the device as ONE call — the chromosomes laid end to end with
    more than a window between them, one stable order by S over all of them (inside a chromosome it is the
    chromosome's own order; variants of different chromosomes never meet in a window, so the sweep prunes the same
    variants) — instead of 22 calls with their uploads, band allocations, bit-image downloads and tails of half-empty
    launches (a third of snp_clumping's wall time at a million variants).  one_call=False / BSN_CLUMP_PER_CHR=1: the loop."""
    import os
    if not jobs:
        return []
    sz = jobs[0][5]
    spans = [float(j[4][-1] - j[4][0]) for j in jobs]
    shift = max(spans) + sz + 2.0
    whole = all(np.all(j[4] == np.floor(j[4])) for j in jobs) and sz == np.floor(sz) and shift * (len(jobs) + 1) < 2.0 ** 52
    if one_call is None:
        one_call = not os.environ.get("BSN_CLUMP_PER_CHR")
    if one_call and len(jobs) > 1 and whole and not im.streamed and all(np.all(np.diff(j[4]) >= 0) for j in jobs):
        ind_all = np.concatenate([j[0] for j in jobs])
        a1, a2 = np.concatenate([j[1] for j in jobs]), np.concatenate([j[2] for j in jobs])
        S_all = np.concatenate([j[3] for j in jobs])
        pos_all = np.concatenate([j[4] - j[4][0] + t * shift for t, j in enumerate(jobs)])
        order = _OrdersAhead([S_all]).get(0)
        kept = _clump_chr(im, ir, ind_all, mode, a1, a2, S_all, pos_all, sz, thr_r2, order=order)
        return [kept]
    ahead = _OrdersAhead([j[3] for j in jobs])
    return [_clump_chr(im, ir, ind_chr, mode, a1, a2, S_chr, pos_chr, sz_, thr_r2, order=ahead.get(t))
            for t, (ind_chr, a1, a2, S_chr, pos_chr, sz_) in enumerate(jobs)]


def snp_clumping(G, infos_chr, ind_row=None, S=None, thr_r2=0.2, size=None, infos_pos=None,
                 exclude=None, ncores=1):
    """R/clumping.R:62-137 (snp_clumping + clumpingChr)"""
    im, ir, _ = _ind(G, ind_row, None)
    infos_chr = np.asarray(infos_chr)
    assert_lengths(infos_chr, cols_along(im))
    size = 100.0 / thr_r2 if size is None else size
    if infos_pos is not None:
        assert_lengths(infos_pos, infos_chr)
    if S is not None:
        assert_lengths(S, infos_chr)
    excl = np.zeros(im.ncol, dtype=bool)
    if exclude is not None and len(exclude):
        excl[np.asarray(exclude, dtype=np.int64)] = True
    kept = []
    # the column statistics of ALL kept variants in one call (the reference computes them per chromosome, R/clumping.R:104-108:
    # the same exact sums per column; 22 calls cost 65 ms of a 270-ms snp_clumping at a million variants, one costs 20)
    sel = np.nonzero(~excl)[0].astype(np.int64)
    st_all = snp_colstats(G, ir, sel) if sel.size else None
    where = np.cumsum(~excl) - 1
    jobs = []
    for chrom, ind_chr in chr_groups(infos_chr, ~excl):
        st = {key: val[where[ind_chr]] for key, val in st_all.items()}
        if S is None:
            af = st["sumX"] / (2.0 * ir.size)
            S_chr = np.minimum(af, 1 - af)
        else:
            S_chr = np.asarray(S, dtype=np.float64)[ind_chr]
        if infos_pos is None:
            pos_chr, sz = np.arange(1, ind_chr.size + 1, dtype=np.float64), float(size)
        else:
            pos_chr, sz = np.asarray(infos_pos, dtype=np.float64)[ind_chr], float(size) * 1000.0
        jobs.append((ind_chr, st["sumX"], st["denoX"], S_chr, pos_chr, sz))
    kept = _clump_jobs(im, ir, jobs, 0, thr_r2)
    return np.sort(np.concatenate(kept)) if kept else np.zeros(0, dtype=np.int64)


def bed_clumping(obj_bed, ind_row=None, S=None, thr_r2=0.2, size=None, exclude=None, ncores=1,
                 infos_chr=None, infos_pos=None):
    """R/bed-clumping.R:7-74; chromosome / position default to the .bim columns"""
    im, ir, _ = _ind(obj_bed, ind_row, None)
    infos_chr = np.asarray(obj_bed.map["chromosome"] if infos_chr is None else infos_chr)
    infos_pos = np.asarray(obj_bed.map["physical_pos"] if infos_pos is None else infos_pos, dtype=np.float64)
    size = 100.0 / thr_r2 if size is None else size
    if S is not None:
        assert_lengths(infos_chr, S)
    excl = np.zeros(im.ncol, dtype=bool)
    if exclude is not None and len(exclude):
        excl[np.asarray(exclude, dtype=np.int64)] = True
    kept = []
    sel = np.nonzero(~excl)[0].astype(np.int64)           # (all kept variants in one call, as in snp_clumping)
    st_all = bed_colstats(obj_bed, ir, sel) if sel.size else None
    where = np.cumsum(~excl) - 1
    jobs = []
    for chrom, ind_chr in chr_groups(infos_chr, ~excl):
        st = {key: val[where[ind_chr]] for key, val in st_all.items() if isinstance(val, np.ndarray) and val.shape[:1] == (sel.size,)}
        with np.errstate(all="ignore"):
            center = st["sumX"] / st["nb_nona_col"]
            scale = np.sqrt(st["denoX"])
        if S is None:
            S_chr = np.minimum(st["sumX"], 2.0 * st["nb_nona_col"] - st["sumX"])  # MAC
        else:
            S_chr = np.asarray(S, dtype=np.float64)[ind_chr]
        jobs.append((ind_chr, center, scale, S_chr, infos_pos[ind_chr], float(size) * 1000.0))
    kept = _clump_jobs(im, ir, jobs, 1, thr_r2)
    return np.sort(np.concatenate(kept)) if kept else np.zeros(0, dtype=np.int64)


# ---- FBM.code256 scaling helpers and SVD (R/binom-scaling.R:12-106, R/autoSVD.R:129-134) ----
def snp_scaleBinom(nploidy=2):
    """R/binom-scaling.R:62-77: returns fun.scaling(X, ind.row, ind.col, ncores)"""
    def fun(X, ind_row=None, ind_col=None, ncores=1):
        im, ir, ic = _ind(X, ind_row, ind_col)
        af = snp_colstats(X, ir, ic)["sumX"] / (ir.size * nploidy)
        with np.errstate(all="ignore"):
            return dict(center=nploidy * af, scale=np.sqrt(nploidy * af * (1 - af)))
    return fun


def snp_scaleAlpha(alpha=-1):
    """R/binom-scaling.R:12-27"""
    def fun(X, ind_row=None, ind_col=None, ncores=1):
        im, ir, ic = _ind(X, ind_row, ind_col)
        af = snp_colstats(X, ir, ic)["sumX"] / (2 * ir.size)
        with np.errstate(all="ignore"):
            return dict(center=2 * af, scale=(2 * af * (1 - af)) ** (-alpha / 2))
    return fun


def snp_MAF(G, ind_row=None, ind_col=None, nploidy=2, ncores=1):
    """R/binom-scaling.R:94-106"""
    im, ir, ic = _ind(G, ind_row, ind_col)
    af = snp_colstats(G, ir, ic)["sumX"] / (ir.size * nploidy)
    return np.minimum(af, 1 - af)


def big_prodVec(X, y_col, ind_row=None, ind_col=None, center=None, scale=None, ncores=1):
    """bigstatsr::big_prodVec for an FBM.code256 (external; callers R/PRS.R:5, R/autoSVD.R:129-134):
    ((X[ind.row, ind.col] - center) / scale) %*% y.col, centre / scale defaulting to 0 / 1."""
    from .bed import bed_prodVec
    im, ir, ic = _ind(X, ind_row, ind_col)
    _no_missing(X, "big_prodVec", ir, ic)
    return bed_prodVec(im, y_col, ir, ic, center, scale)


def big_cprodVec(X, y_row, ind_row=None, ind_col=None, center=None, scale=None, ncores=1):
    """bigstatsr::big_cprodVec: crossprod((X[ind.row, ind.col] - center) / scale, y.row)"""
    from .bed import bed_cprodVec
    im, ir, ic = _ind(X, ind_row, ind_col)
    _no_missing(X, "big_cprodVec", ir, ic)
    return bed_cprodVec(im, y_row, ir, ic, center, scale)


def big_randomSVD(X, fun_scaling=None, ind_row=None, ind_col=None, k=10, tol=1e-4, verbose=False,
                  ncores=1, **kw):
    """The FBM entry of the partial SVD as snp_autoSVD calls it (R/autoSVD.R:129-134;
    bigstatsr::big_randomSVD is external): same device solver as bed_randomSVD, on the
    FBM's 2-bit image."""
    from .svd import bed_randomSVD
    from .bed import bed_scaleBinom
    im, ir, ic = _ind(X, ind_row, ind_col)
    _no_missing(X, "big_randomSVD", ir, ic)
    if fun_scaling is None and int(_lib.load().bsn_bed_bits(im.handle)) == 2:
        # the default snp_scaleBinom() on data without missing values is bed_scaleBinom's formula with
        # nb_nona = n (R/binom-scaling.R:62-77 vs :133-142): evaluated inside the solve, its counts ride
        # along the first crossproduct pass
        return bed_randomSVD(im, fun_scaling=bed_scaleBinom, ind_row=ir, ind_col=ic, k=k, tol=tol,
                             verbose=verbose, ncores=ncores, **kw)
    fs = snp_scaleBinom() if fun_scaling is None else fun_scaling
    return bed_randomSVD(im, fun_scaling=lambda obj, ind_row, ind_col, ncores=1: fs(X, ind_row, ind_col, ncores),
                         ind_row=ir, ind_col=ic, k=k, tol=tol, verbose=verbose, ncores=ncores, **kw)


def last_stats():
    """figures of the last LD call of this process (bsn_ld_last_stats): bench.py --workload ld"""
    out = np.zeros(5)
    check(_lib.load().bsn_ld_last_stats(ptr(out, f64p)))
    names = ("k_pair_stats<6 products, fused fp64 epilogue>", "k_pair_stats<6 products, K split> + k_band_fill",
             "k_pair_xy64 (cross product only: no missing values) + k_band_fill",
             "k_pair_stats8 (dosage bytes with missing values: 8 products) + k_band_fill8na",
             "k_pair_stats_b<6 products, column operand decoded once per workgroup through LDS, fused fp64 epilogue>",
             "(unused)",
             "k_pair_stats_f4<6 products on the FP4 matrix pipe (v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales, exact), "
             "column operand decoded once per workgroup through LDS, fused fp64 epilogue>",
             "k_pair_xy_f4 (cross product only on the FP4 matrix pipe: no missing values, codes as E2M1 nibbles) + k_band_fill",
             "k_quad_xy_f4 (cross product only on the FP4 matrix pipe, 2 x 2 tile pairs per workgroup through LDS) + k_band_fill",
             "k_pair_stats_f4<4 of the 6 products (the bed clumping formula reads no sum of squares) on the FP4 matrix pipe, "
             "column operand decoded once per workgroup through LDS, fused fp64 epilogue>",
             "k_pair_stats_f4<RAW: 6 products of look-up-free planes (code, high bit, missing) on the FP4 matrix pipe, recombined with "
             "the per-variant totals in the fused fp64 epilogue; column operand decoded once per workgroup through LDS>",
             "k_pair_stats_f4<RAW: 4 products (the bed clumping formula) of look-up-free planes (code, missing) on the FP4 matrix pipe, "
             "recombined with the per-variant totals in the fused fp64 epilogue>")
    return dict(pairs=out[0], tile_pairs=out[1], stats_ms=out[2], launches=int(out[3]), kernel=names[int(out[4])],
                products={2: 1, 3: 8, 7: 1, 8: 1, 9: 4, 11: 4}.get(int(out[4]), 6))
