"""Host-side mirror of bigsnpr's R API for the .bed-backed hot path.

Same function names and argument meaning as the reference (``ind.row`` -> ``ind_row``
etc.); the only deliberate difference is that indices are 0-based, as is idiomatic in
Python (the R shim in bindings/R keeps R's 1-based convention).  Everything here is a thin
argument-checking layer over the C ABI (include/bigsnpr_hip.h); no arithmetic on genotypes
happens in Python.

Reference: R/bed-class.R, R/bed-mult-vec.R, R/binom-scaling.R, R/utils-assert.R.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import as_f64, as_i64, check, f64p, i32p, i64p, ptr, u8p, vp

ERROR_DIM = "Incompatibility between dimensions."  # bigstatsr:::GET_ERROR_DIM()


def assert_lengths(*arrs):
    """bigassertr::assert_lengths"""
    ls = {len(a) for a in arrs}
    if len(ls) > 1:
        raise ValueError(ERROR_DIM + "\nArguments should have the same length.")


def _count_lines(path):
    with open(path, "rb") as f:
        return sum(1 for _ in f)


def sub_bed(path, replacement="", stop_if_not_ext=True):
    """R/bed-class.R:20-32: replace the extension '.bed'"""
    if not str(path).endswith(".bed"):
        raise ValueError("Path '%s' must have 'bed' extension." % path)
    if stop_if_not_ext and len(replacement) > 0 and replacement[0] != ".":
        raise ValueError("Replacement must be an extension starting with '.' if provided.")
    return str(path)[:-4] + replacement


class bed:
    """RC class ``bed`` (R/bed-class.R:65-134): a PLINK .bed file attached to the GPU.

    ``bed(bedfile)`` checks that the .bed/.bim/.fam triple exists, takes nrow/ncol from
    the .fam/.bim line counts and maps + uploads the genotypes (the reference's lazy
    ``address`` binding, R/bed-class.R:105-110, calls bedXPtr the same way)."""

    def __init__(self, bedfile=None, _handle=None, _n=None, _m=None):
        self._h = None
        self._map = None
        self._fam = None
        if _handle is not None:
            self.bedfile = None
            self._h, self.nrow, self.ncol = _handle, int(_n), int(_m)
            return
        bedfile = os.path.expanduser(str(bedfile))
        if not bedfile.endswith(".bed"):
            raise ValueError("Path '%s' must have 'bed' extension." % bedfile)
        self.bedfile = bedfile
        for f in (self.bedfile, self.bimfile, self.famfile):
            if not os.path.exists(f):
                raise FileNotFoundError("File '%s' doesn't exist." % f)
        self.nrow = _count_lines(self.famfile)
        self.ncol = _count_lines(self.bimfile)
        h = vp()
        check(_lib.load().bsn_bed_open(self.bedfile.encode(), self.nrow, self.ncol, C.byref(h)))
        self._h = h

    # -- constructors for data that does not come from a file --------------------------
    @classmethod
    def from_payload(cls, payload, n, m):
        payload = np.ascontiguousarray(payload, dtype=np.uint8).ravel()
        n_byte = (n + 3) // 4
        if payload.size != n_byte * m:
            raise ValueError("n or p does not match the dimensions of the file.")
        h = vp()
        check(_lib.load().bsn_bed_from_host(ptr(payload, u8p), n, m, n_byte, C.byref(h)))
        return cls(_handle=h, _n=n, _m=m)

    @classmethod
    def from_fbm(cls, bytes_nm):
        """FBM.code256 with CODE_012 (one byte per genotype) repacked to 2 bits in HBM."""
        a = np.asfortranarray(np.asarray(bytes_nm, dtype=np.uint8))
        n, m = a.shape
        h = vp()
        check(_lib.load().bsn_bed_from_fbm(a.ctypes.data_as(u8p), n, m, n, C.byref(h)))
        return cls(_handle=h, _n=n, _m=m)

    @classmethod
    def synthetic(cls, n, m, seed=20250905, npop=24, na16=655, j_begin=0):
        h = vp()
        check(_lib.load().bsn_bed_synthetic(n, m, seed, npop, na16, j_begin, C.byref(h)))
        return cls(_handle=h, _n=n, _m=m)

    # -- R fields ---------------------------------------------------------------------
    @property
    def prefix(self):
        return self.bedfile[:-4]

    @property
    def bimfile(self):
        return self.prefix + ".bim"

    @property
    def famfile(self):
        return self.prefix + ".fam"

    @property
    def map(self):
        """columns of the .bim file (NAMES.MAP, R/utils.R:49-50)"""
        if self._map is None:
            chrom, snp, gd, pos, a1, a2 = [], [], [], [], [], []
            with open(self.bimfile) as f:
                for line in f:
                    t = line.split()
                    chrom.append(t[0]); snp.append(t[1]); gd.append(float(t[2]))
                    pos.append(int(float(t[3]))); a1.append(t[4]); a2.append(t[5])
            try:
                chrom = np.array([int(c) for c in chrom])
            except ValueError:
                chrom = np.array(chrom)
            self._map = dict(chromosome=chrom, marker_ID=np.array(snp),
                             genetic_dist=np.array(gd), physical_pos=np.array(pos),
                             allele1=np.array(a1), allele2=np.array(a2))
        return self._map

    @property
    def fam(self):
        """columns of the .fam file (NAMES.FAM, R/utils.R:47-48)"""
        if self._fam is None:
            cols = [[] for _ in range(6)]
            with open(self.famfile) as f:
                for line in f:
                    t = line.split()
                    for c in range(6):
                        cols[c].append(t[c])
            names = ("family_ID", "sample_ID", "paternal_ID", "maternal_ID", "sex", "affection")
            self._fam = {k: np.array(v) for k, v in zip(names, cols)}
            for k in ("sex", "affection"):
                try:
                    self._fam[k] = self._fam[k].astype(np.int64)
                except ValueError:
                    pass
        return self._fam

    @property
    def light(self):
        return self  # bed_light only exists to make forked R workers cheap (R/bed-class.R:176)

    @property
    def handle(self):
        if self._h is None:
            raise _lib.BsnError("bed object is closed")
        return self._h

    @property
    def shape(self):
        return (self.nrow, self.ncol)

    def __len__(self):
        return self.nrow * self.ncol

    def __repr__(self):
        return "A 'bed' object with %d samples and %d variants." % (self.nrow, self.ncol)

    def hbm_bytes(self):
        return int(_lib.load().bsn_bed_bytes(self.handle))

    def tile(self):
        """builds the streaming-layout copy of the image ahead of time (bsn_bed_tile); True if it exists"""
        built = C.c_int(0)
        check(_lib.load().bsn_bed_tile(self.handle, C.byref(built)))
        return bool(built.value)

    @property
    def streamed(self):
        """True for an out-of-core handle: the image did not fit the device (or BSN_IMAGE_BUDGET) when the file was
        opened; products, counts and the accessor then walk the mapped file in slabs (bsn_bed_is_streamed)"""
        return bool(_lib.load().bsn_bed_is_streamed(self.handle))

    def release_workspace(self):
        """bsn_bed_release_workspace: the solve's workspace, the second copies of the image and the cached work buffers"""
        check(_lib.load().bsn_bed_release_workspace(self.handle))

    def sample_major(self):
        """builds the sample-major copy of the image ahead of time (bsn_bed_sample_major); True if it exists"""
        built = C.c_int(0)
        check(_lib.load().bsn_bed_sample_major(self.handle, C.byref(built)))
        return bool(built.value)

    def download(self):
        out = np.empty(((self.nrow + 3) // 4) * self.ncol, dtype=np.uint8)
        check(_lib.load().bsn_bed_download(self.handle, ptr(out, u8p)))
        return out

    def close(self):
        if self._h is not None:
            _lib.load().bsn_bed_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # `[` accessor, R/bed-mat-acc.R:21-38 -> read_bed (NA as -1 here)
    def __getitem__(self, key):
        i, j = key if isinstance(key, tuple) else (key, slice(None))
        ir = np.arange(self.nrow)[i]
        ic = np.arange(self.ncol)[j]
        return read_bed(self, np.atleast_1d(ir), np.atleast_1d(ic))


def assert_bed(obj):
    if not isinstance(obj, bed):
        raise TypeError("'obj.bed' is not of class 'bed' or 'bed_light'.")


def rows_along(obj):
    return np.arange(obj.nrow, dtype=np.int64)


def cols_along(obj):
    return np.arange(obj.ncol, dtype=np.int64)


def _check_ind(name, ind, limit):
    """check_args(): assert_not_null / assert_int / assert_pos (R/utils-assert.R:28-33)"""
    if ind is None:
        raise ValueError("'%s' can't be `NULL`." % name)
    a = np.asarray(ind)
    if a.dtype.kind not in "iu":
        if a.dtype.kind == "f" and np.all(a == np.floor(a)):
            a = a.astype(np.int64)
        else:
            raise ValueError("'%s' should contain only integers." % name)
    a = np.ascontiguousarray(a, dtype=np.int64).ravel()
    if a.size and (a.min() < 0 or a.max() >= limit):
        raise IndexError("'%s' should have only positive values (and be in range)." % name)
    return a


def _args(obj_bed, ind_row, ind_col):
    assert_bed(obj_bed)
    ir = rows_along(obj_bed) if ind_row is None else _check_ind("ind.row", ind_row, obj_bed.nrow)
    ic = cols_along(obj_bed) if ind_col is None else _check_ind("ind.col", ind_col, obj_bed.ncol)
    return ir, ic


def _lazy_ind(obj_bed, name, ind, limit):
    """(pointer, length) of an index list for the C ABI; `None` (the R default rows_along / cols_along) goes down
    as a NULL pointer, which every entry point reads as "the leading `length` ones": no 8 m-byte list is built,
    uploaded or scanned for a call over the whole matrix"""
    if ind is None:
        return None, int(limit)
    a = _check_ind(name, ind, limit)
    return a, a.size


def _center_scale_lazy(center, scale, m):
    """centre / scale for the C ABI: `None` (R default rep(0, m) / rep(1, m)) goes down as NULL"""
    c = None if center is None else as_f64(np.ravel(center))
    s = None if scale is None else as_f64(np.ravel(scale))
    for v in (c, s):
        if v is not None and v.size != m:
            raise ValueError(ERROR_DIM)
    return c, s


def _center_scale(center, scale, ic):
    center = np.zeros(ic.size) if center is None else as_f64(np.ravel(center))
    scale = np.ones(ic.size) if scale is None else as_f64(np.ravel(scale))
    assert_lengths(center, ic)
    assert_lengths(scale, ic)
    return center, scale


def bed_prodVec(obj_bed, y_col, ind_row=None, ind_col=None, center=None, scale=None, ncores=1, comm=None):
    """R/bed-mult-vec.R:58-75 -> bed_pMatVec4.  ``ncores`` is accepted and ignored.  ``comm`` (a
    bigsnpr_amd.Comm, not in the reference): this rank holds a shard of the columns; every rank gets the product
    with the whole matrix (bsn_bed_prodvec_sharded)."""
    assert_bed(obj_bed)
    ir, n = _lazy_ind(obj_bed, "ind.row", ind_row, obj_bed.nrow)
    ic, m = _lazy_ind(obj_bed, "ind.col", ind_col, obj_bed.ncol)
    y_col = as_f64(np.ravel(y_col))
    if y_col.size != m:
        raise ValueError(ERROR_DIM)
    center, scale = _center_scale_lazy(center, scale, m)
    out = np.empty(n)
    if comm is not None:
        check(_lib.load().bsn_bed_prodvec_sharded(obj_bed.handle, ptr(ir, i64p), n, ptr(ic, i64p), m,
                                                  ptr(center, f64p), ptr(scale, f64p), ptr(y_col, f64p),
                                                  comm.handle, ptr(out, f64p)))
        return out
    check(_lib.load().bsn_bed_prodvec(obj_bed.handle, ptr(ir, i64p), n, ptr(ic, i64p), m, ptr(center, f64p),
                                      ptr(scale, f64p), ptr(y_col, f64p), ptr(out, f64p)))
    return out


def bed_cprodVec(obj_bed, y_row, ind_row=None, ind_col=None, center=None, scale=None, ncores=1):
    """R/bed-mult-vec.R:20-37 -> bed_cpMatVec4."""
    assert_bed(obj_bed)
    ir, n = _lazy_ind(obj_bed, "ind.row", ind_row, obj_bed.nrow)
    ic, m = _lazy_ind(obj_bed, "ind.col", ind_col, obj_bed.ncol)
    y_row = as_f64(np.ravel(y_row))
    if y_row.size != n:
        raise ValueError(ERROR_DIM)
    center, scale = _center_scale_lazy(center, scale, m)
    out = np.empty(m)
    check(_lib.load().bsn_bed_cprodvec(obj_bed.handle, ptr(ir, i64p), n, ptr(ic, i64p), m, ptr(center, f64p),
                                       ptr(scale, f64p), ptr(y_row, f64p), ptr(out, f64p)))
    return out


def bed_colstats(obj_bed, ind_row=None, ind_col=None, ncores=1):
    """src/bed-fun.cpp:9-46; warns like the reference when variants have > 50 % NA."""
    ir, ic = _args(obj_bed, ind_row, ind_col)
    sumX, denoX = np.empty(ic.size), np.empty(ic.size)
    nona = np.empty(ic.size, dtype=np.int32)
    n_bad = C.c_int32(0)
    check(_lib.load().bsn_bed_colstats(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p),
                                       ic.size, ptr(sumX, f64p), ptr(denoX, f64p),
                                       ptr(nona, i32p), C.byref(n_bad)))
    if n_bad.value > 0:
        import warnings
        warnings.warn("%d variants have >50%% missing values." % n_bad.value)
    return dict(sumX=sumX, denoX=denoX, nb_nona_col=nona)


def bed_counts(obj_bed, ind_row=None, ind_col=None, byrow=False, ncores=1):
    """R/binom-scaling.R:166-178; 4 x m (4 x n if byrow), rows = counts of 0, 1, 2, NA."""
    ir, ic = _args(obj_bed, ind_row, ind_col)
    if byrow:  # src/bed-fun.cpp:72-99
        res = np.empty((ir.size, 4), dtype=np.int32)
        check(_lib.load().bsn_bed_row_counts(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p),
                                             ic.size, ptr(res, i32p)))
        return res.T
    res = np.empty((ic.size, 4), dtype=np.int32)
    check(_lib.load().bsn_bed_col_counts(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p),
                                         ic.size, ptr(res, i32p)))
    return res.T


def bed_MAF(obj_bed, ind_row=None, ind_col=None, ncores=1):
    """R/binom-scaling.R:203-222"""
    ir, ic = _args(obj_bed, ind_row, ind_col)
    counts = bed_counts(obj_bed, ir, ic).astype(np.int64)
    ac = counts[1] + 2 * counts[2]
    nb_nona = ir.size - counts[3]
    with np.errstate(all="ignore"):
        af = ac / (2.0 * nb_nona)
    return dict(ac=ac, mac=np.minimum(ac, 2 * nb_nona - ac), af=af,
                maf=np.minimum(af, 1 - af), N=nb_nona)


def bed_scaleBinom(obj_bed, ind_row=None, ind_col=None, ncores=1):
    """R/binom-scaling.R:133-142"""
    st = bed_colstats(obj_bed, ind_row, ind_col, ncores)
    with np.errstate(all="ignore"):
        af = st["sumX"] / (2.0 * st["nb_nona_col"])
        return dict(center=2 * af, scale=np.sqrt(2 * af * (1 - af)))


def read_bed(obj_bed, ind_row, ind_col, na_val=-1):
    """src/bed-mat-acc.cpp:8-25"""
    ir, ic = _args(obj_bed, ind_row, ind_col)
    out = np.empty((ic.size, ir.size), dtype=np.int32)
    check(_lib.load().bsn_bed_read(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p), ic.size,
                                   na_val, ptr(out, i32p)))
    return out.T


def read_bed_scaled(obj_bed, ind_row, ind_col, center, scale):
    """src/bed-mat-acc.cpp:30-49"""
    ir, ic = _args(obj_bed, ind_row, ind_col)
    center, scale = _center_scale(center, scale, ic)
    out = np.empty((ic.size, ir.size), dtype=np.float64)
    check(_lib.load().bsn_bed_read_scaled(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p),
                                          ic.size, ptr(center, f64p), ptr(scale, f64p),
                                          ptr(out, f64p)))
    return out.T


class ScaledOp:
    """Device-resident A~ = scaled G[ind_row, ind_col] (the fun.prod / fun.cprod pair that
    bed_randomSVD hands to big_randomSVD, R/autoSVD.R:216-218), operating on panels that
    stay in HBM."""

    def __init__(self, obj_bed, ind_row=None, ind_col=None, center=None, scale=None, slices=4):
        ir, ic = _args(obj_bed, ind_row, ind_col)
        center, scale = _center_scale(center, scale, ic)
        self.bed, self.n, self.m = obj_bed, ir.size, ic.size
        h = vp()
        check(_lib.load().bsn_op_create(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p),
                                        ic.size, ptr(center, f64p), ptr(scale, f64p), C.byref(h)))
        self._h = h
        check(_lib.load().bsn_op_set_slices(h, slices))

    def prod(self, X, Y=None):
        """Y[n x k] = A~ X[m x k] on DeviceArrays"""
        Y = _lib.DeviceArray(self.n, X.cols) if Y is None else Y
        check(_lib.load().bsn_op_prod(self._h, X.ptr, X.rows, X.cols, Y.ptr, Y.rows))
        return Y

    def cprod(self, X, Z=None):
        Z = _lib.DeviceArray(self.m, X.cols) if Z is None else Z
        check(_lib.load().bsn_op_cprod(self._h, X.ptr, X.rows, X.cols, Z.ptr, Z.rows))
        return Z

    def sync(self):
        check(_lib.load().bsn_op_sync(self._h))

    def close(self):
        if self._h is not None:
            _lib.load().bsn_op_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
