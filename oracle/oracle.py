"""ctypes/numpy front-end of the CPU oracle (oracle/bsn_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of bsn_oracle.c.  Imported by tests/,
by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of bench.py;
never by anything under ``bigsnpr_amd/``.

Indices are 0-based everywhere (the reference's R API is 1-based; its native
side subtracts 1, src/bed-acc.h:64-65).
"""
import ctypes as C
import gzip
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSN_SANITIZE=1 (tests/test_sanitizers_cpu.py): the same source under -fsanitize=address,undefined
_SAN = bool(os.environ.get("BSN_SANITIZE"))
_SO = os.path.join(_HERE, "libbsn_oracle_san.so" if _SAN else "libbsn_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "bsn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", os.path.basename(_SO)])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_corMat.restype = C.c_int64
        _lib.orc_clumping_chr_cached.restype = C.c_int64
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


u8p, i64p, i32p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int64),
                         C.POINTER(C.c_int32), C.POINTER(C.c_double))


class BedFile:
    """A .bed file held in host memory (src/bed-acc.h:18-48 `class bed`)."""

    def __init__(self, path=None, n=None, m=None, raw=None):
        if raw is None:
            with open(path, "rb") as f:
                raw = np.frombuffer(f.read(), dtype=np.uint8)
        self.raw = np.ascontiguousarray(raw, dtype=np.uint8)
        if n is None or m is None:
            n, m = dims_from_sidecars(path)
        self.n, self.m = int(n), int(m)
        self.n_byte = (self.n + 3) // 4
        rc = lib().orc_bed_check(_p(self.raw, C.c_uint8), C.c_int64(self.raw.size),
                                 C.c_int64(self.n), C.c_int64(self.m))
        if rc == 1:
            raise ValueError("File is not a binary PED file.")
        if rc == 2:
            raise ValueError("Variant-major is the only mode supported.")
        if rc == 3:
            raise ValueError("n or p does not match the dimensions of the file.")
        self.payload = self.raw[3:]

    @classmethod
    def from_payload(cls, payload, n, m):
        raw = np.concatenate([np.array([0x6C, 0x1B, 0x01], dtype=np.uint8),
                              np.ascontiguousarray(payload, dtype=np.uint8).ravel()])
        return cls(raw=raw, n=n, m=m)

    def rows(self):
        return np.arange(self.n, dtype=np.int64)

    def cols(self):
        return np.arange(self.m, dtype=np.int64)


def dims_from_sidecars(bedpath):
    base = bedpath[:-4]
    with open(base + ".fam") as f:
        n = sum(1 for _ in f)
    with open(base + ".bim") as f:
        m = sum(1 for _ in f)
    return n, m


def read_bim(bedpath):
    chrom, pos = [], []
    with open(bedpath[:-4] + ".bim") as f:
        for line in f:
            t = line.split()
            chrom.append(int(t[0]))
            pos.append(int(float(t[3])))
    return np.array(chrom, dtype=np.int64), np.array(pos, dtype=np.float64)


def _sub(bed, ind_row, ind_col):
    ir = bed.rows() if ind_row is None else _i64(ind_row)
    ic = bed.cols() if ind_col is None else _i64(ind_col)
    return ir, ic


def read_bed(bed, ind_row=None, ind_col=None, na_val=-1):
    ir, ic = _sub(bed, ind_row, ind_col)
    out = np.empty((ic.size, ir.size), dtype=np.int32)
    lib().orc_read_bed(_p(bed.payload, C.c_uint8), C.c_int64(bed.n_byte), _p(ir, C.c_int64),
                       C.c_int64(ir.size), _p(ic, C.c_int64), C.c_int64(ic.size),
                       C.c_int32(na_val), _p(out, C.c_int32))
    return out.T  # n x m view (column-major storage like R)


def get_code(na_val=3):
    """getCode(), R/utils.R:21-31: the 4 x 256 raw table of snp_readBed — column `byte` holds the four genotypes of
    that .bed byte (bit pair 00 -> 2, 10 -> 1, 11 -> 0, 01 -> NA.VAL), lowest bit pair first."""
    bits = ((np.arange(256)[:, None] >> np.arange(8)[None, :]) & 1) == 0      # !as.logical(rawToBits(0:255))
    geno1, geno2 = bits[:, 0::2], bits[:, 1::2]                               # s = c(TRUE, FALSE)
    geno = geno1.astype(np.int64) + geno2
    geno[~geno1 & geno2] = na_val
    return np.ascontiguousarray(geno.T.astype(np.uint8))                      # dim 4 x 256


def readbina(bedpath, n, m, tab):
    """readbina, src/read-plink.cpp:13-56: the FBM bytes (n x m, one per genotype) of a whole .bed file decoded
    through the 4 x 256 table `tab`, and the end-of-file flag (`!myFile.get(c)` after the m-th variant)."""
    tab = np.asarray(tab, dtype=np.uint8).reshape(4, 256)
    raw = np.fromfile(bedpath, dtype=np.uint8)
    assert raw[0] == 108 and raw[1] == 27, "Wrong magic number. Aborting.."   # (:31-32; the third test is an assignment)
    length, extra = n // 4, n % 4
    length_extra = length + (extra > 0)
    out = np.empty((n, m), dtype=np.uint8, order="F")
    for j in range(m):
        buf = raw[3 + j * length_extra: 3 + (j + 1) * length_extra]
        col = tab[:, buf[:length]].T.reshape(-1)                              # std::copy(code_ptr, code_ptr + 4, ptr)
        if extra:
            col = np.concatenate([col, tab[:extra, buf[length]]])
        out[:, j] = col
    return out, raw.size <= 3 + m * length_extra


def read_bed_scaled(bed, ind_row, ind_col, center, scale):
    ir, ic = _sub(bed, ind_row, ind_col)
    center, scale = _f64(center), _f64(scale)
    out = np.empty((ic.size, ir.size), dtype=np.float64)
    lib().orc_read_bed_scaled(_p(bed.payload, C.c_uint8), C.c_int64(bed.n_byte),
                              _p(ir, C.c_int64), C.c_int64(ir.size), _p(ic, C.c_int64),
                              C.c_int64(ic.size), _p(center, C.c_double),
                              _p(scale, C.c_double), _p(out, C.c_double))
    return out.T


def _matvec(fn, bed, x, ind_row, ind_col, center, scale, ncores, out_len_is_rows):
    ir, ic = _sub(bed, ind_row, ind_col)
    center = np.zeros(ic.size) if center is None else _f64(center)
    scale = np.ones(ic.size) if scale is None else _f64(scale)
    x = _f64(x)
    out = np.empty(ir.size if out_len_is_rows else ic.size, dtype=np.float64)
    fn(_p(bed.payload, C.c_uint8), C.c_int64(bed.n_byte), _p(ir, C.c_int64),
       C.c_int64(ir.size), _p(ic, C.c_int64), C.c_int64(ic.size), _p(center, C.c_double),
       _p(scale, C.c_double), _p(x, C.c_double), _p(out, C.c_double), C.c_int(ncores))
    return out


def bed_prodVec(bed, y_col, ind_row=None, ind_col=None, center=None, scale=None, ncores=1):
    return _matvec(lib().orc_pMatVec4, bed, y_col, ind_row, ind_col, center, scale, ncores, True)


def bed_cprodVec(bed, y_row, ind_row=None, ind_col=None, center=None, scale=None, ncores=1):
    return _matvec(lib().orc_cpMatVec4, bed, y_row, ind_row, ind_col, center, scale, ncores, False)


def bed_colstats(bed, ind_row=None, ind_col=None, ncores=1):
    ir, ic = _sub(bed, ind_row, ind_col)
    sumX = np.empty(ic.size)
    denoX = np.empty(ic.size)
    nona = np.empty(ic.size, dtype=np.int32)
    with np.errstate(all="ignore"):
        lib().orc_bed_colstats(_p(bed.payload, C.c_uint8), C.c_int64(bed.n_byte),
                               _p(ir, C.c_int64), C.c_int64(ir.size), _p(ic, C.c_int64),
                               C.c_int64(ic.size), _p(sumX, C.c_double),
                               _p(denoX, C.c_double), _p(nona, C.c_int32), C.c_int(ncores))
    return dict(sumX=sumX, denoX=denoX, nb_nona_col=nona)


def bed_col_counts(bed, ind_row=None, ind_col=None, ncores=1):
    ir, ic = _sub(bed, ind_row, ind_col)
    res = np.empty((ic.size, 4), dtype=np.int32)
    lib().orc_bed_col_counts(_p(bed.payload, C.c_uint8), C.c_int64(bed.n_byte),
                             _p(ir, C.c_int64), C.c_int64(ir.size), _p(ic, C.c_int64),
                             C.c_int64(ic.size), _p(res, C.c_int32), C.c_int(ncores))
    return res.T  # 4 x m


def bed_scaleBinom(bed, ind_row=None, ind_col=None):
    """R/binom-scaling.R:133-142"""
    st = bed_colstats(bed, ind_row, ind_col)
    af = st["sumX"] / (2.0 * st["nb_nona_col"])
    return dict(center=2 * af, scale=np.sqrt(2 * af * (1 - af)))


def bed_MAF(bed, ind_row=None, ind_col=None):
    """R/binom-scaling.R:203-222"""
    ir, _ = _sub(bed, ind_row, ind_col)
    counts = bed_col_counts(bed, ind_row, ind_col).astype(np.int64)
    ac = counts[1] + 2 * counts[2]
    nb_nona = ir.size - counts[3]
    af = ac / (2.0 * nb_nona)
    return dict(ac=ac, mac=np.minimum(ac, 2 * nb_nona - ac), af=af,
                maf=np.minimum(af, 1 - af), N=nb_nona)


# ---- FBM.code256 -----------------------------------------------------------
CODE_012 = np.array([0, 1, 2] + [np.nan] * 253, dtype=np.float64)  # R/bigSNP-class.R:7


class FBM256:
    """bigstatsr FBM.code256 stand-in: n x m bytes, column-major, + code256."""

    def __init__(self, data_nm, code256=CODE_012):
        a = np.asarray(data_nm, dtype=np.uint8)
        self.n, self.m = a.shape
        self.bytes = np.asfortranarray(a)  # column-major storage
        self.code256 = _f64(code256)

    def flat(self):
        return self.bytes.ravel(order="F")


def fbm_from_bed(bed):
    """What snp_readBed does to the genotypes (src/read-plink.cpp:13-56): decoded
    0/1/2 and 3 for missing, one byte each."""
    g = read_bed(bed, na_val=3)
    return FBM256(g.astype(np.uint8))


def snp_colstats(G, ind_row=None, ind_col=None, ncores=1):
    ir = np.arange(G.n, dtype=np.int64) if ind_row is None else _i64(ind_row)
    ic = np.arange(G.m, dtype=np.int64) if ind_col is None else _i64(ind_col)
    flat = G.flat()
    sumX, denoX = np.empty(ic.size), np.empty(ic.size)
    lib().orc_snp_colstats(_p(flat, C.c_uint8), C.c_int64(G.n), _p(G.code256, C.c_double),
                           _p(ir, C.c_int64), C.c_int64(ir.size), _p(ic, C.c_int64),
                           C.c_int64(ic.size), _p(sumX, C.c_double), _p(denoX, C.c_double),
                           C.c_int(ncores))
    return dict(sumX=sumX, denoX=denoX)


def _acc_args(obj):
    """(kind, data, ld, code256) for corMat / ld_scores dispatch, src/corr.cpp:113-125."""
    if isinstance(obj, BedFile):
        return 0, obj.payload, obj.n_byte, np.zeros(256), obj.n, obj.m
    code = obj.code256.copy()
    code[np.isnan(code)] = 3  # src/corr.cpp:115-116
    return 1, obj.flat(), obj.n, code, obj.n, obj.m


def corMat(obj, ind_row, ind_col, size, thr, pos, fill_diag=True, ncores=1):
    """Returns CSC (i, p, x) exactly as R/corr.R:43-47 assembles them."""
    kind, data, ld, code, n_tot, m_tot = _acc_args(obj)
    ir = np.arange(n_tot, dtype=np.int64) if ind_row is None else _i64(ind_row)
    ic = np.arange(m_tot, dtype=np.int64) if ind_col is None else _i64(ind_col)
    thr, pos = _f64(thr), _f64(pos)
    assert thr.size == ir.size and pos.size == ic.size
    p = np.zeros(ic.size + 1, dtype=np.int32)
    pi, px = i32p(), f64p()
    with np.errstate(all="ignore"):
        nnz = lib().orc_corMat(C.c_int(kind), _p(data, C.c_uint8), C.c_int64(ld),
                               _p(code, C.c_double), _p(ir, C.c_int64), C.c_int64(ir.size),
                               _p(ic, C.c_int64), C.c_int64(ic.size), C.c_double(size),
                               _p(thr, C.c_double), _p(pos, C.c_double), C.c_int(fill_diag),
                               C.c_int(ncores), _p(p, C.c_int32), C.byref(pi), C.byref(px))
    i = np.ctypeslib.as_array(pi, shape=(max(nnz, 1),))[:nnz].copy()
    x = np.ctypeslib.as_array(px, shape=(max(nnz, 1),))[:nnz].copy()
    lib().orc_free(pi)
    lib().orc_free(px)
    return i, p, x


def cor_thresholds(n, alpha=1.0, thr_r2=0.0):
    """R/corr.R:18-23 + :29.  THR from the t quantile; alpha = 1 gives 0."""
    from scipy import stats
    df = np.arange(1, n + 1, dtype=np.float64) - 2
    with np.errstate(all="ignore"):
        q = stats.t.isf(alpha / 2, df=np.where(df > 0, df, np.nan))
        THR = q / np.sqrt(df + q * q)
    # R's pmax() propagates NaN (df <= 0 entries), like np.maximum
    return np.maximum(THR, np.sqrt(thr_r2))


def snp_cor(obj, ind_row=None, ind_col=None, size=500, alpha=1.0, thr_r2=0.0,
            fill_diag=True, infos_pos=None, ncores=1):
    """R/corr.R:3-57 (cor0)."""
    kind, data, ld, code, n_tot, m_tot = _acc_args(obj)
    ir = np.arange(n_tot, dtype=np.int64) if ind_row is None else _i64(ind_row)
    ic = np.arange(m_tot, dtype=np.int64) if ind_col is None else _i64(ind_col)
    pos = 1000.0 * np.arange(1, ic.size + 1) if infos_pos is None else _f64(infos_pos)
    thr = cor_thresholds(ir.size, alpha, thr_r2)
    return corMat(obj, ir, ic, size * 1000.0, thr, pos, fill_diag, ncores)


def ld_scores(obj, ind_row=None, ind_col=None, size=500, infos_pos=None):
    """R/ld-scores.R:3-22 (ld0) + src/ld-scores.cpp."""
    kind, data, ld, code, n_tot, m_tot = _acc_args(obj)
    ir = np.arange(n_tot, dtype=np.int64) if ind_row is None else _i64(ind_row)
    ic = np.arange(m_tot, dtype=np.int64) if ind_col is None else _i64(ind_col)
    pos = 1000.0 * np.arange(1, ic.size + 1) if infos_pos is None else _f64(infos_pos)
    res = np.empty(ic.size)
    with np.errstate(all="ignore"):
        lib().orc_ld_scores(C.c_int(kind), _p(data, C.c_uint8), C.c_int64(ld),
                            _p(code, C.c_double), _p(ir, C.c_int64), C.c_int64(ir.size),
                            _p(ic, C.c_int64), C.c_int64(ic.size), C.c_double(size * 1000.0),
                            _p(pos, C.c_double), _p(res, C.c_double))
    return res


def r_order_decreasing(S):
    """R's order(S, decreasing = TRUE): stable, ties keep ascending index."""
    S = np.asarray(S)
    return np.argsort(-S, kind="stable")


def _rank_from_order(ord_):
    rank = np.empty(ord_.size, dtype=np.int32)
    rank[ord_] = np.arange(ord_.size, dtype=np.int32)
    return rank


def snp_clumping(G, infos_chr, ind_row=None, S=None, thr_r2=0.2, size=None,
                 infos_pos=None, exclude=None):
    """R/clumping.R:62-137 (snp_clumping + clumpingChr), 0-based indices."""
    size = 100.0 / thr_r2 if size is None else size
    infos_chr = np.asarray(infos_chr)
    ir = np.arange(G.n, dtype=np.int64) if ind_row is None else _i64(ind_row)
    excl = np.zeros(G.m, dtype=bool)
    if exclude is not None and len(exclude):
        excl[np.asarray(exclude, dtype=np.int64)] = True
    flat = G.flat()
    kept = []
    for chrom in np.unique(infos_chr[~excl]):
        ind_chr = np.nonzero((infos_chr == chrom) & ~excl)[0].astype(np.int64)
        st = snp_colstats(G, ir, ind_chr)
        n = ir.size
        if S is None:
            af = st["sumX"] / (2.0 * n)
            S_chr = np.minimum(af, 1 - af)
        else:
            S_chr = np.asarray(S, dtype=np.float64)[ind_chr]
        ord_ = r_order_decreasing(S_chr).astype(np.int32)
        rank = _rank_from_order(ord_)
        if infos_pos is None:
            pos_chr = np.arange(1, ind_chr.size + 1, dtype=np.float64)
            sz = float(size)
        else:
            sz = float(size) * 1000.0
            pos_chr = _f64(np.asarray(infos_pos)[ind_chr])
        keep = np.full(ind_chr.size, -1, dtype=np.int32)
        with np.errstate(all="ignore"):
            lib().orc_clumping_chr(_p(flat, C.c_uint8), C.c_int64(G.n),
                                   _p(G.code256, C.c_double), _p(ir, C.c_int64),
                                   C.c_int64(ir.size), _p(ind_chr, C.c_int64),
                                   C.c_int64(ind_chr.size), _p(ord_, C.c_int32),
                                   _p(rank, C.c_int32), _p(pos_chr, C.c_double),
                                   _p(st["sumX"], C.c_double), _p(st["denoX"], C.c_double),
                                   C.c_double(sz), C.c_double(thr_r2), _p(keep, C.c_int32))
        assert np.all((keep == 0) | (keep == 1))
        kept.append(ind_chr[keep == 1])
    return np.sort(np.concatenate(kept)) if kept else np.zeros(0, dtype=np.int64)


def bed_clumping(bed, infos_chr, infos_pos, ind_row=None, S=None, thr_r2=0.2, size=None,
                 exclude=None):
    """R/bed-clumping.R:7-74 (bed_clumping + bedClumpingChr), 0-based indices."""
    size = 100.0 / thr_r2 if size is None else size
    infos_chr = np.asarray(infos_chr)
    ir = bed.rows() if ind_row is None else _i64(ind_row)
    excl = np.zeros(bed.m, dtype=bool)
    if exclude is not None and len(exclude):
        excl[np.asarray(exclude, dtype=np.int64)] = True
    kept = []
    for chrom in np.unique(infos_chr[~excl]):
        ind_chr = np.nonzero((infos_chr == chrom) & ~excl)[0].astype(np.int64)
        st = bed_colstats(bed, ir, ind_chr)
        with np.errstate(all="ignore"):
            center = st["sumX"] / st["nb_nona_col"]
            scale = np.sqrt(st["denoX"])
        if S is None:
            S_chr = np.minimum(st["sumX"], 2.0 * st["nb_nona_col"] - st["sumX"])
        else:
            S_chr = np.asarray(S, dtype=np.float64)[ind_chr]
        ord_ = r_order_decreasing(S_chr).astype(np.int32)
        rank = _rank_from_order(ord_)
        pos_chr = _f64(np.asarray(infos_pos)[ind_chr])
        keep = np.full(ind_chr.size, -1, dtype=np.int32)
        with np.errstate(all="ignore"):
            lib().orc_bed_clumping_chr(_p(bed.payload, C.c_uint8), C.c_int64(bed.n_byte),
                                       _p(ir, C.c_int64), C.c_int64(ir.size),
                                       _p(ind_chr, C.c_int64), C.c_int64(ind_chr.size),
                                       _p(center, C.c_double), _p(scale, C.c_double),
                                       _p(ord_, C.c_int32), _p(rank, C.c_int32),
                                       _p(pos_chr, C.c_double), C.c_double(size * 1000.0),
                                       C.c_double(thr_r2), _p(keep, C.c_int32))
        kept.append(ind_chr[keep == 1])
    return np.sort(np.concatenate(kept)) if kept else np.zeros(0, dtype=np.int64)


def fbm_prodVec(G, x, ind_row=None, ind_col=None):
    """bigstatsr::big_prodVec without centre / scale (external; plain definition y = G[ir, ic] x on the
    decoded values, oracle/bsn_oracle.c:orc_fbm_prodVec)"""
    ir = np.arange(G.n, dtype=np.int64) if ind_row is None else _i64(ind_row)
    ic = np.arange(G.m, dtype=np.int64) if ind_col is None else _i64(ind_col)
    x, flat, y = _f64(x), G.flat(), np.empty(ir.size)
    lib().orc_fbm_prodVec(_p(flat, C.c_uint8), C.c_int64(G.n), _p(G.code256, C.c_double), _p(ir, C.c_int64),
                          C.c_int64(ir.size), _p(ic, C.c_int64), C.c_int64(ic.size), _p(x, C.c_double),
                          _p(y, C.c_double))
    return y


def fbm_cprodVec(G, x, ind_row=None, ind_col=None):
    """bigstatsr::big_cprodVec without centre / scale: z = G[ir, ic]' x (orc_fbm_cprodVec)"""
    ir = np.arange(G.n, dtype=np.int64) if ind_row is None else _i64(ind_row)
    ic = np.arange(G.m, dtype=np.int64) if ind_col is None else _i64(ind_col)
    x, flat, z = _f64(x), G.flat(), np.empty(ic.size)
    lib().orc_fbm_cprodVec(_p(flat, C.c_uint8), C.c_int64(G.n), _p(G.code256, C.c_double), _p(ir, C.c_int64),
                           C.c_int64(ir.size), _p(ic, C.c_int64), C.c_int64(ic.size), _p(x, C.c_double),
                           _p(z, C.c_double))
    return z


def prodVecRev(G, betas_col, same_col, ind_row, ind_col):
    """R/PRS.R:3-7"""
    betas_col = _f64(betas_col)
    same_col = np.asarray(same_col, dtype=bool)
    mod = (2.0 * same_col - 1.0) * betas_col
    ir, ic = _i64(ind_row), _i64(ind_col)
    flat = G.flat()
    y = np.empty(ir.size)
    lib().orc_fbm_prodVec(_p(flat, C.c_uint8), C.c_int64(G.n), _p(G.code256, C.c_double),
                          _p(ir, C.c_int64), C.c_int64(ir.size), _p(ic, C.c_int64),
                          C.c_int64(ic.size), _p(mod, C.c_double), _p(y, C.c_double))
    return y + 2.0 * betas_col[~same_col].sum()


def snp_PRS(G, betas_keep, ind_test=None, ind_keep=None, same_keep=None, lpS_keep=None,
            thr_list=(0,)):
    """R/PRS.R:36-76; returns n x T."""
    ind_test = np.arange(G.n) if ind_test is None else np.asarray(ind_test)
    ind_keep = np.arange(G.m) if ind_keep is None else np.asarray(ind_keep)
    betas_keep = _f64(betas_keep)
    same_keep = np.ones(ind_keep.size, bool) if same_keep is None else np.asarray(same_keep, bool)
    thr_list = np.atleast_1d(np.asarray(thr_list, dtype=np.float64))
    if lpS_keep is None or (thr_list.size == 1 and thr_list[0] == 0):
        return prodVecRev(G, betas_keep, same_keep, ind_test, ind_keep)[:, None]
    lpS_keep = _f64(lpS_keep)
    scores = np.full((ind_test.size, thr_list.size), np.nan)
    ind_rem = np.arange(ind_keep.size)
    last = 0.0
    for i in r_order_decreasing(thr_list):
        pass_thr = lpS_keep[ind_rem] > thr_list[i]
        ind = ind_rem[pass_thr]
        last = last + prodVecRev(G, betas_keep[ind], same_keep[ind], ind_test, ind_keep[ind])
        scores[:, i] = last
        ind_rem = ind_rem[~pass_thr]
    return scores


def bed_tcrossprodSelf(bed, ind_row=None, ind_col=None):
    """R/bed-tcrossprodSelf.R:21-52 with fun.scaling = bed_scaleBinom."""
    ir, ic = _sub(bed, ind_row, ind_col)
    ms = bed_scaleBinom(bed, ir, ic)
    A = read_bed_scaled(bed, ir, ic, ms["center"], ms["scale"])
    return A @ A.T, ms


def dense_svd(bed, ind_row=None, ind_col=None, k=10):
    """Partial SVD oracle.  bed_randomSVD (R/autoSVD.R:205-219) delegates to
    bigstatsr::big_randomSVD -> RSpectra::svds (bigstatsr >= 1.6.2, RSpectra;
    both external, not in the reference tree).  Their published contract is the
    rank-k truncated SVD of the scaled matrix, which the reference's own test
    pins through a dense eigen-decomposition (test-2-bed-clumping-SVD.R:76-79);
    this is that dense definition."""
    ir, ic = _sub(bed, ind_row, ind_col)
    ms = bed_scaleBinom(bed, ir, ic)
    A = read_bed_scaled(bed, ir, ic, ms["center"], ms["scale"])
    U, d, Vt = np.linalg.svd(A, full_matrices=False)
    return dict(d=d[:k], u=U[:, :k], v=Vt[:k].T, center=ms["center"], scale=ms["scale"])


def fake_bed(n, m, seed=20250905, npop=24, na16=655, j_begin=0):
    """Synthetic .bed payload (DESIGN.md §Synthetic inputs); identical bytes to the
    device generator."""
    n_byte = (n + 3) // 4
    payload = np.empty(n_byte * m, dtype=np.uint8)
    lib().orc_fake_bed(_p(payload, C.c_uint8), C.c_int64(n), C.c_int64(m), C.c_int64(n_byte),
                       C.c_uint32(seed), C.c_uint32(npop), C.c_uint32(na16),
                       C.c_int64(j_begin))
    return BedFile.from_payload(payload, n, m)


# ---- minimal reader for R's serialize() XDR v2/v3 (gzip'd .rds) -------------
def read_rds(path):
    """Parses the subset of R serialization used by the reference's golden .rds
    files (INTSXP / REALSXP vectors, optionally with attributes)."""
    with gzip.open(path, "rb") as f:
        b = f.read()
    assert b[:2] == b"X\n"
    off = [2]

    def i32():
        v = struct.unpack_from(">i", b, off[0])[0]
        off[0] += 4
        return v

    version = i32()
    i32(); i32()
    if version == 3:
        nlen = i32()
        off[0] += nlen

    def item():
        flags = i32()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        if t == 13:  # INTSXP
            n = i32()
            v = np.frombuffer(b, dtype=">i4", count=n, offset=off[0]).astype(np.int32)
            off[0] += 4 * n
        elif t == 14:  # REALSXP
            n = i32()
            v = np.frombuffer(b, dtype=">f8", count=n, offset=off[0]).astype(np.float64)
            off[0] += 8 * n
        elif t == 16:  # STRSXP
            n = i32()
            v = [item() for _ in range(n)]
        elif t == 9:  # CHARSXP
            n = i32()
            v = None if n == -1 else b[off[0]:off[0] + n].decode()
            off[0] += max(n, 0)
        elif t == 2:  # LISTSXP (pairlist) — attributes
            v = {}
            fl = flags
            while True:
                has_tag = bool(fl & 0x400)
                tag = item() if has_tag else None
                val = item()
                v[tag] = val
                fl = i32()
                if (fl & 0xFF) == 254:
                    break
                assert (fl & 0xFF) == 2
            return v
        elif t == 1:  # SYMSXP
            v = item()
        elif t == 255:  # REFSXP
            v = "ref%d" % (flags >> 8)
        elif t == 254:
            return None
        else:
            raise NotImplementedError("SEXP type %d" % t)
        if has_attr:
            attrs = item()
            return dict(value=v, attr=attrs)
        return v

    return item()


def prod_and_rowSumsSq(bed, ind_row, ind_col, center, scale, V):
    """src/bed-fun.cpp:103-133"""
    ir, ic = _sub(bed, ind_row, ind_col)
    center, scale = _f64(center), _f64(scale)
    V = np.asfortranarray(np.asarray(V, dtype=np.float64))
    assert V.shape[0] == ic.size
    K = V.shape[1]
    XV = np.empty((ir.size, K), dtype=np.float64, order="F")
    rs = np.empty(ir.size)
    lib().orc_prod_and_rowSumsSq(_p(bed.payload, C.c_uint8), C.c_int64(bed.n_byte), _p(ir, C.c_int64),
                                 C.c_int64(ir.size), _p(ic, C.c_int64), C.c_int64(ic.size),
                                 _p(center, C.c_double), _p(scale, C.c_double),
                                 V.ctypes.data_as(f64p), C.c_int64(K), XV.ctypes.data_as(f64p),
                                 _p(rs, C.c_double))
    return XV, rs


def multLinReg(obj, ind_row, ind_col, U, ncores=1):
    """src/multLinReg.cpp:8-86: m x K t-scores"""
    kind, data, ld, code, n_tot, m_tot = _acc_args(obj)
    ir = np.arange(n_tot, dtype=np.int64) if ind_row is None else _i64(ind_row)
    ic = np.arange(m_tot, dtype=np.int64) if ind_col is None else _i64(ind_col)
    U = np.asfortranarray(np.asarray(U, dtype=np.float64))
    if U.ndim == 1:
        U = U[:, None]
    assert U.shape[0] == ir.size
    res = np.empty((ic.size, U.shape[1]), dtype=np.float64, order="F")
    with np.errstate(all="ignore"):
        lib().orc_multLinReg(C.c_int(kind), _p(data, C.c_uint8), C.c_int64(ld), _p(code, C.c_double),
                             _p(ir, C.c_int64), C.c_int64(ir.size), _p(ic, C.c_int64),
                             C.c_int64(ic.size), U.ctypes.data_as(f64p), C.c_int64(U.shape[1]),
                             res.ctypes.data_as(f64p), C.c_int(ncores))
    return res


# ---- f4: Stacked C+T grids, R/SCT.R --------------------------------------------------------
def seq_log(from_, to, length_out):
    """R/SCT.R:150-154"""
    if length_out < 0:
        raise ValueError("'length.out' must be a non-negative number")
    return np.exp(np.linspace(np.log(from_), np.log(to), int(length_out)))


def snp_grid_clumping(G, infos_chr, infos_pos, lpS, ind_row=None,
                      grid_thr_r2=(0.01, 0.05, 0.1, 0.2, 0.5, 0.8, 0.95),
                      grid_base_size=(50, 100, 200, 500), infos_imp=None, grid_thr_imp=(1,),
                      groups=None, exclude=None, stats=None):
    """R/SCT.R:32-134 + src/clumping-cached.cpp:11-107, 0-based indices.  Returns
    (all_keep: list per chromosome of list of index arrays, grid dict, n_computed)."""
    infos_chr, infos_pos, lpS = np.asarray(infos_chr), _f64(infos_pos), _f64(lpS)
    m_all = G.m
    infos_imp = np.ones(m_all) if infos_imp is None else _f64(infos_imp)
    groups = [np.arange(m_all)] if groups is None else groups
    ir = np.arange(G.n, dtype=np.int64) if ind_row is None else _i64(ind_row)
    THR_IMP = np.unique(np.asarray(grid_thr_imp, dtype=np.float64))
    THR_CLMP = np.unique(np.asarray(grid_thr_r2, dtype=np.float64))
    BASE = np.unique(np.asarray(grid_base_size, dtype=np.float64))
    grid = dict(size=[], thr_r2=[], grp_num=[], thr_imp=[])
    for ti in THR_IMP:
        for g in range(len(groups)):
            for tc in THR_CLMP:
                for bs in BASE:
                    grid["size"].append(int(bs / tc)); grid["thr_r2"].append(tc)
                    grid["grp_num"].append(g); grid["thr_imp"].append(ti)
    grid = {k: np.asarray(v) for k, v in grid.items()}
    excl = np.zeros(m_all, dtype=bool)
    if exclude is not None and len(exclude):
        excl[np.asarray(exclude, dtype=np.int64)] = True
    flat = G.flat()
    all_keep, computed = [], 0
    for chrom in np.unique(infos_chr[~excl]):
        ind_chr = np.nonzero((infos_chr == chrom) & ~excl)[0].astype(np.int64)
        info, S, pos = infos_imp[ind_chr], lpS[ind_chr], infos_pos[ind_chr]
        st = snp_colstats(G, ir, ind_chr)
        sumX, denoX = st["sumX"], st["denoX"]
        sp_id = np.arange(ind_chr.size, dtype=np.int32)      # row / column in spcor.chr
        if np.any(np.diff(pos) < 0):
            raise ValueError("'pos.chr' is not sorted.")
        cap = 1 << 22
        keys, vals = np.full(cap, -1, dtype=np.int64), np.zeros(cap)
        ind_keep = []
        for ti in THR_IMP:
            sel = np.nonzero(info >= ti)[0]
            ind_chr, info, pos, S = ind_chr[sel], info[sel], pos[sel], S[sel]
            sumX, denoX, sp_id = sumX[sel], denoX[sel], sp_id[sel]
            for group in groups:
                ind2 = np.nonzero(np.isin(ind_chr, np.asarray(group, dtype=np.int64)))[0]
                if ind2.size == 0:
                    ind_keep += [np.zeros(0, dtype=np.int64)] * (THR_CLMP.size * BASE.size)
                    continue
                cols = _i64(ind_chr[ind2])
                ord_ = r_order_decreasing(S[ind2]).astype(np.int32)
                rank = _rank_from_order(ord_)
                pos_g, sum_g, den_g = _f64(pos[ind2]), _f64(sumX[ind2]), _f64(denoX[ind2])
                sp_g = _i32(sp_id[ind2])
                for tc in THR_CLMP:
                    for bs in BASE:
                        keep = np.full(cols.size, -1, dtype=np.int32)
                        with np.errstate(all="ignore"):
                            computed += lib().orc_clumping_chr_cached(
                                _p(flat, C.c_uint8), C.c_int64(G.n), _p(G.code256, C.c_double),
                                _p(keys, C.c_int64), _p(vals, C.c_double), C.c_int64(cap),
                                _p(sp_g, C.c_int32), _p(ir, C.c_int64), C.c_int64(ir.size),
                                _p(cols, C.c_int64), C.c_int64(cols.size), _p(ord_, C.c_int32),
                                _p(rank, C.c_int32), _p(pos_g, C.c_double), _p(sum_g, C.c_double),
                                _p(den_g, C.c_double), C.c_double(1000.0 * bs / tc), C.c_double(tc),
                                _p(keep, C.c_int32))
                        assert np.all((keep == 0) | (keep == 1))
                        ind_keep.append(cols[keep == 1])
        all_keep.append(ind_keep)
    return all_keep, grid, computed


def snp_grid_PRS(G, all_keep, betas, lpS, grid_lpS_thr, ind_row=None):
    """R/SCT.R:201-262 (scores only): one snp_PRS per clumping set, n x (sets * thresholds)"""
    betas, lpS = _f64(betas), _f64(lpS)
    ir = np.arange(G.n, dtype=np.int64) if ind_row is None else _i64(ind_row)
    sets = [k for chrom in all_keep for k in chrom]
    thr = np.atleast_1d(np.asarray(grid_lpS_thr, dtype=np.float64))
    out = np.empty((ir.size, len(sets) * thr.size))
    for ic, ind_keep in enumerate(sets):
        out[:, ic * thr.size:(ic + 1) * thr.size] = snp_PRS(
            G, betas[ind_keep], ind_test=ir, ind_keep=ind_keep, lpS_keep=lpS[ind_keep], thr_list=thr)
    return out


def prod_and_rowSumsSq2(G, ind_row, ind_col, center, scale, V):
    """src/project-utils.cpp:12-43 (FBM.code256: decoded value through code256, NA propagates)"""
    ir = np.arange(G.n) if ind_row is None else np.asarray(ind_row)
    ic = np.arange(G.m) if ind_col is None else np.asarray(ind_col)
    X = G.code256[G.bytes[np.ix_(ir, ic)]]
    X = (X - _f64(center)[None, :]) / _f64(scale)[None, :]
    return X @ np.asarray(V, dtype=np.float64), (X * X).sum(1)


# ---- GWAS step of the reference's PRS test (external to the path, needed to use its golden files) ----------
def univ_logreg(G, y01, covar, ind_col=None, tol=1e-10, maxiter=50):
    """bigstatsr::big_univLogReg (EXTERNAL to the reference tree; tests/testthat/test-6-PRS.R:19-22 calls it):
    for every variant j the maximum-likelihood logistic regression y ~ 1 + covar + G[, j]; returns the
    coefficient of the variant (`estim`), its standard error from the inverse Fisher information
    (`std.err`), the Wald statistic `score` = estim / std.err and the two-sided normal p-value, which is what
    predict(gwas, log10 = FALSE) reports.  The MLE is unique (Newton / IRLS from the covariates-only fit),
    so this restatement of the published definition is pinned by the reference's own pval.rds."""
    from scipy.stats import norm
    ic = np.arange(G.m) if ind_col is None else np.asarray(ind_col)
    dec = G.code256[G.bytes[:, ic]]                                   # n x m decoded genotypes
    y = np.asarray(y01, dtype=np.float64)
    n = y.size
    X0 = np.column_stack([np.ones(n), np.asarray(covar, dtype=np.float64)])
    p0 = X0.shape[1]
    b0 = np.zeros(p0)                                                 # the null model: covariates only
    for _ in range(100):
        mu = 1.0 / (1.0 + np.exp(-(X0 @ b0)))
        step = np.linalg.solve(X0.T @ (X0 * (mu * (1 - mu))[:, None]), X0.T @ (y - mu))
        b0 += step
        if np.abs(step).max() < 1e-12:
            break
    estim, se = np.empty(ic.size), np.empty(ic.size)
    for j0 in range(0, ic.size, 256):
        g = dec[:, j0:j0 + 256]                                       # n x B
        B = g.shape[1]
        X = np.concatenate([np.broadcast_to(X0[:, None, :], (n, B, p0)), g[:, :, None]], axis=2)   # n x B x p
        beta = np.concatenate([np.broadcast_to(b0, (B, p0)), np.zeros((B, 1))], axis=1)            # B x p
        for _ in range(maxiter):
            eta = np.einsum("nbp,bp->nb", X, beta)
            mu = 1.0 / (1.0 + np.exp(-eta))
            w = mu * (1 - mu)
            H = np.einsum("nbp,nb,nbq->bpq", X, w, X)
            grad = np.einsum("nbp,nb->bp", X, y[:, None] - mu)
            step = np.linalg.solve(H, grad[:, :, None])[:, :, 0]
            beta = beta + step
            if np.abs(step).max() < tol:
                break
        eta = np.einsum("nbp,bp->nb", X, beta)
        mu = 1.0 / (1.0 + np.exp(-eta))
        H = np.einsum("nbp,nb,nbq->bpq", X, mu * (1 - mu), X)
        cov = np.linalg.inv(H)
        estim[j0:j0 + B] = beta[:, -1]
        se[j0:j0 + B] = np.sqrt(cov[:, -1, -1])
    score = estim / se
    return dict(estim=estim, std_err=se, score=score, pval=2.0 * norm.sf(np.abs(score)))
