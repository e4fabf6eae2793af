"""ctypes binding of libbigsnpr_hip.so (the C ABI in include/bigsnpr_hip.h).

There is no fallback: if the shared library is missing, or a compute entry point is
called without a HIP device, this raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbigsnpr_hip.so")

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p
i64 = C.c_int64

ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_void_p)


class SvdOptions(C.Structure):
    _fields_ = [("k", C.c_int32), ("tol", C.c_double), ("block", C.c_int32),
                ("slices", C.c_int32), ("max_basis", C.c_int32), ("seed", C.c_uint32),
                ("verbose", C.c_int32), ("m_total", C.c_int64), ("allreduce", ALLREDUCE_FN),
                ("allreduce_ctx", C.c_void_p), ("hook_rank", C.c_int32), ("hook_world", C.c_int32),
                ("comm", C.c_void_p), ("binom_scaling", C.c_int32),
                ("center_out", C.POINTER(C.c_double)), ("scale_out", C.POINTER(C.c_double)),
                ("warm_start", C.c_int32), ("warm_denominator", C.c_int32), ("max_restarts", C.c_int32),
                ("vec_floor", C.c_double), ("exchange_timing", C.c_int32), ("exchange_timeout_ms", C.c_int32)]


class SvdInfo(C.Structure):
    _fields_ = [("niter", C.c_int32), ("nops", C.c_int32), ("basis", C.c_int32),
                ("converged", C.c_int32), ("max_rel_resid", C.c_double), ("gpu_ms", C.c_double),
                ("cprod_ms", C.c_double), ("prod_ms", C.c_double), ("n_cprod", C.c_int32),
                ("n_prod", C.c_int32), ("block", C.c_int32), ("slices", C.c_int32),
                ("n_bad", C.c_int32), ("fused_stats", C.c_int32), ("cprod_stats_ms", C.c_double),
                ("n_cprod_stats", C.c_int32), ("warm_launches", C.c_int32), ("warm_fraction", C.c_double),
                ("warm_ms", C.c_double), ("tiled", C.c_int32), ("segmented_passes", C.c_int32), ("compact_gathers", C.c_int32),
                ("slices_max", C.c_int32), ("wide_steps", C.c_int32), ("wide_cprod_ms", C.c_double),
                ("wide_prod_ms", C.c_double), ("n_wide_cprod", C.c_int32), ("n_wide_prod", C.c_int32),
                ("lead_rel_resid", C.c_double), ("exchange_mode", C.c_int32), ("n_exchange", C.c_int32 * 4),
                ("exchange_ms", C.c_double * 4), ("compacted", C.c_int32), ("compact_ms", C.c_double),
                ("out_of_core", C.c_int32), ("na_free_steps", C.c_double * 2), ("na_skip", C.c_int32)]


# name -> (restype, argtypes); kept in one table so tests can check that every symbol
# declared in include/bigsnpr_hip.h is exported and bound.
SIGNATURES = {
    "bsn_last_error": (C.c_char_p, []),
    "bsn_version": (C.c_int, []),
    "bsn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bsn_set_device": (C.c_int, [C.c_int]),
    "bsn_selftest": (C.c_int, []),
    "bsn_comm_unique_id": (C.c_int, [u8p]),
    "bsn_comm_init": (C.c_int, [u8p, C.c_int, C.c_int, C.POINTER(vp)]),
    "bsn_comm_rank": (C.c_int, [vp]),
    "bsn_comm_world": (C.c_int, [vp]),
    "bsn_comm_allreduce": (C.c_int, [vp, vp, i64]),
    "bsn_comm_abort": (C.c_int, [vp]),
    "bsn_comm_destroy": (C.c_int, [vp]),
    "bsn_ld_last_stats": (C.c_int, [f64p]),
    "bsn_robust_scale_tau2": (C.c_int, [vp, i64, i64, C.c_int32, C.c_double, C.c_double, f64p, f64p]),
    "bsn_robust_pair_scales": (C.c_int, [vp, i64, i64, C.c_int32, C.c_double, C.c_double, f64p, f64p]),
    "bsn_robust_mc_count": (C.c_int, [vp, i64, vp, i64, C.c_double, C.POINTER(C.c_int64)]),
    "bsn_robust_scale_cols": (C.c_int, [vp, i64, i64, C.c_int32, f64p]),
    "bsn_robust_rotate": (C.c_int, [vp, i64, i64, C.c_int32, f64p]),
    "bsn_robust_wdist": (C.c_int, [vp, i64, i64, C.c_int32, f64p, f64p, f64p]),
    "bsn_robust_dist_ogk": (C.c_int, [vp, i64, i64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, f64p, C.POINTER(C.c_int64)]),
    "bsn_robust_rollmean": (C.c_int, [f64p, i64, f64p, C.c_int32, C.POINTER(C.c_int64), C.c_int32, f64p]),
    "bsn_robust_sort": (C.c_int, [f64p, i64]),
    "bsn_order_decreasing": (C.c_int, [f64p, i64, C.POINTER(C.c_int64), C.c_int32, i32p, i32p]),
    "bsn_robust_mc_window": (C.c_int, [vp, i64, vp, i64, C.c_double, C.c_double, i64, f64p, C.POINTER(C.c_int64)]),
    "bsn_bed_open": (C.c_int, [C.c_char_p, i64, i64, C.POINTER(vp)]),
    "bsn_bed_from_host": (C.c_int, [u8p, i64, i64, i64, C.POINTER(vp)]),
    "bsn_bed_from_fbm": (C.c_int, [u8p, i64, i64, i64, C.POINTER(vp)]),
    "bsn_fbm_open": (C.c_int, [u8p, i64, i64, i64, f64p, C.POINTER(vp)]),
    "bsn_bed_bits": (C.c_int, [vp]),
    "bsn_bed_tile": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "bsn_bed_sample_major": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "bsn_bed_na_known": (i64, [vp]),
    "bsn_bed_synthetic": (C.c_int, [i64, i64, C.c_uint32, C.c_uint32, C.c_uint32, i64, C.POINTER(vp)]),
    "bsn_bed_close": (C.c_int, [vp]),
    "bsn_bed_release_workspace": (C.c_int, [vp]),
    "bsn_bed_nrow": (i64, [vp]),
    "bsn_bed_ncol": (i64, [vp]),
    "bsn_bed_bytes": (i64, [vp]),
    "bsn_bed_download": (C.c_int, [vp, u8p]),
    "bsn_bed_prodvec": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, f64p, f64p]),
    "bsn_bed_cprodvec": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, f64p, f64p]),
    "bsn_bed_prodvec_sharded": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, f64p, vp, f64p]),
    "bsn_bed_col_counts": (C.c_int, [vp, i64p, i64, i64p, i64, i32p]),
    "bsn_bed_row_counts": (C.c_int, [vp, i64p, i64, i64p, i64, i32p]),
    "bsn_bed_colstats": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, i32p, i32p]),
    "bsn_snp_colstats": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p]),
    "bsn_bed_read": (C.c_int, [vp, i64p, i64, i64p, i64, C.c_int32, i32p]),
    "bsn_bed_read_scaled": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, f64p]),
    "bsn_bed_cprod_planes": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, i64, f64p, f64p]),
    "bsn_bed_prod_and_rowsumssq": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, f64p, i64, f64p, f64p]),
    "bsn_snp_prod_and_rowsumssq2": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, f64p, i64, f64p, f64p]),
    "bsn_mult_lin_reg": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, i64, f64p]),
    "bsn_bed_to_fbm": (C.c_int, [vp, i64p, i64, i64p, i64, u8p]),
    "bsn_bed_readbina": (C.c_int, [vp, u8p, u8p]),
    "bsn_bed_is_streamed": (C.c_int, [vp]),
    "bsn_bed_streaming_kernels": (C.c_int, [vp, C.c_char_p, i64]),
    "bsn_bed_subset_payload": (C.c_int, [vp, i64p, i64, i64p, i64, u8p]),
    "bsn_op_create": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, C.POINTER(vp)]),
    "bsn_op_destroy": (C.c_int, [vp]),
    "bsn_op_set_slices": (C.c_int, [vp, C.c_int]),
    "bsn_op_prod": (C.c_int, [vp, vp, i64, C.c_int, vp, i64]),
    "bsn_op_cprod": (C.c_int, [vp, vp, i64, C.c_int, vp, i64]),
    "bsn_op_sync": (C.c_int, [vp]),
    "bsn_bed_randomsvd": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, C.POINTER(SvdOptions),
                                    f64p, f64p, f64p, C.POINTER(SvdInfo)]),
    "bsn_bed_tcrossprod": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, f64p, i64, f64p]),
    "bsn_cormat": (C.c_int, [vp, i64p, i64, i64p, i64, C.c_double, f64p, f64p, C.c_int, i32p,
                             C.POINTER(C.c_int64), C.POINTER(vp)]),
    "bsn_cormat_fetch": (C.c_int, [vp, i32p, f64p]),
    "bsn_cormat_has_nan": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "bsn_cormat_free": (C.c_int, [vp]),
    "bsn_ld_scores": (C.c_int, [vp, i64p, i64, i64p, i64, C.c_double, f64p, f64p]),
    "bsn_clumping_chr": (C.c_int, [vp, i64p, i64, i64p, i64, C.c_int, f64p, f64p, i32p, i32p, f64p,
                                   C.c_double, C.c_double, i32p]),
    "bsn_clumping_chr_cached": (C.c_int, [vp, i64p, i64, i64p, i64, C.c_int, f64p, f64p, i32p, i32p, f64p,
                                          i64, f64p, f64p, i32p]),
    "bsn_snp_grid_prs": (C.c_int, [vp, i64p, i64, i64p, i64, f64p, i32p, C.POINTER(C.c_uint8), i64, i64,
                                   C.c_int, f64p]),
    "bsn_malloc": (C.c_int, [C.POINTER(vp), i64]),
    "bsn_free": (C.c_int, [vp]),
    "bsn_host_alloc": (C.c_int, [C.POINTER(vp), i64]),
    "bsn_host_free": (C.c_int, [vp]),
    "bsn_memcpy_h2d": (C.c_int, [vp, vp, i64]),
    "bsn_memcpy_d2h": (C.c_int, [vp, vp, i64]),
    "bsn_device_sync": (C.c_int, []),
    "bsn_timer_start": (C.c_int, [vp]),
    "bsn_timer_stop": (C.c_int, [vp, f64p]),
}

_lib = None


class BsnError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        # BSN_LIB_PATH: another build of the same ABI (the -DBSN_ABLATION profiling build, tools/)
        LIB_PATH = os.environ.get("BSN_LIB_PATH") or globals()["LIB_PATH"]
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -m bigsnpr_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise BsnError(load().bsn_last_error().decode("utf-8", "replace"))


def ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def as_i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def as_f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class DeviceArray:
    """A column-major fp64 matrix in HBM owned by the library (rows x cols)."""

    def __init__(self, rows, cols=1):
        self.rows, self.cols = int(rows), int(cols)
        p = vp()
        check(load().bsn_malloc(C.byref(p), self.rows * self.cols * 8))
        self.ptr = p

    @classmethod
    def from_numpy(cls, a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            a = a[:, None]
        d = cls(a.shape[0], a.shape[1])
        h = np.asfortranarray(a)
        check(load().bsn_memcpy_h2d(d.ptr, h.ctypes.data_as(vp), h.nbytes))
        return d

    def to_numpy(self):
        h = np.empty((self.rows, self.cols), dtype=np.float64, order="F")
        check(load().bsn_memcpy_d2h(h.ctypes.data_as(vp), self.ptr, h.nbytes))
        return h

    def col_ptr(self, j):
        return vp(self.ptr.value + 8 * self.rows * j)

    def free(self):
        if self.ptr is not None and self.ptr.value:
            load().bsn_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PinnedBlock:
    """a block of page-locked host memory (bsn_host_alloc); goes back to the pool when the last array
    over it is collected"""

    def __init__(self, pool, addr, size):
        self.pool, self.addr, self.size = pool, addr, size
        self.__array_interface__ = {"data": (addr, False), "shape": (size,), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self.pool._give_back(self.addr, self.size)
        except Exception:
            pass


class PinnedPool:
    """Result buffers for the large outputs (u, v of bed_randomSVD): page-locked blocks that the DMA
    engines write directly.  A block is reused once every numpy array over it has been collected, so
    results stay valid for as long as the caller holds them; at most `keep` bytes of free blocks are kept."""

    def __init__(self, keep=None):
        # free blocks kept for reuse: 1 GiB unless BSN_RESULT_POOL_KEEP (bytes) says otherwise — u + v of the
        # 400K x 1M, k = 20 solve are 224 MB
        self.free = []
        self.keep = int(os.environ.get("BSN_RESULT_POOL_KEEP", 1 << 30)) if keep is None else keep
        # ... but never less than what the caller's own pattern needs: the largest total of blocks that were alive at
        # once (bed_cor at config C5 hands back @i + @x = 2.4 GB per call: with a fixed 1-GiB cap every call page-locked
        # its result afresh, 339 -> 463 ms), up to BSN_RESULT_POOL_MAX (8 GiB)
        self.keep_max = int(os.environ.get("BSN_RESULT_POOL_MAX", 8 << 30))
        self.live = 0
        self.peak_live = 0
        # lazy: a request that no free block serves is answered with ordinary memory while a helper thread page-locks a block
        # of that size for the next one (BSN_RESULT_POOL_EAGER=1: page-lock in the call itself, as before round 6)
        self.lazy = not os.environ.get("BSN_RESULT_POOL_EAGER")
        self.pending = set()
        self.wanted = []
        self.prefetched = 0
        # re-entrant: __del__ of a block may run on any thread, also on THIS one inside empty() (a cyclic-GC pass
        # triggered by an allocation under the lock finalises a block, whose _give_back takes the lock again)
        self.lock = threading.RLock()

    def empty(self, shape, dtype=np.float64):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        size = max(4096, (n + 4095) // 4096 * 4096)
        addr = None
        with self.lock:
            pick = None
            for i, (_, sz) in enumerate(self.free):
                if sz >= size and (pick is None or sz < self.free[pick][1]):
                    pick = i
            if pick is not None and self.free[pick][1] <= 2 * size:
                addr, sz = self.free.pop(pick)
        if addr is None and self.lazy:
            # Nothing fits: page-locking a large block costs ~0.2 ms per MB (45 ms for u + v of the 400K x 1M, k = 20 solve)
            # and the caller is about to wait for it — the FIRST call of a process, the one the reference's users time.
            # That call gets ordinary memory (the library stages the download: + 6 ms) and a helper thread page-locks a
            # block of this size meanwhile, so that the next request finds it (round 6).
            with self.lock:
                if size not in self.pending:
                    self.pending.add(size)
                    self.wanted.append(size)
            return np.empty(shape, dtype=dtype)
        if addr is None:
            p = vp()
            if load().bsn_host_alloc(C.byref(p), size) != 0 or not p.value:
                # page-locked memory is an optimisation (DMA straight into the result): when the host refuses it
                # (ulimit -l, fragmentation) the result goes to ordinary memory through the staging buffers
                return np.empty(shape, dtype=dtype)
            addr, sz = p.value, size
        with self.lock:
            self.live += sz
            self.peak_live = max(self.peak_live, self.live)
        blk = _PinnedBlock(self, addr, sz)
        return np.asarray(blk)[:n].view(dtype).reshape(shape)

    def kick(self):
        """page-lock, on a helper thread, the blocks that requests since the last call had to do without.  Called AFTER the
        library call that fills the result (bed_randomSVD, bed_cor): beside it the helper's hipHostMalloc and the solve's own
        allocations queue up on the runtime's lock (31 ms of the first solve, profiles/r06_cold.txt)"""
        with self.lock:
            sizes, self.wanted = self.wanted, []
        for size in sizes:
            threading.Thread(target=self._prefetch, args=(size,), daemon=True).start()

    def _prefetch(self, size):
        p = vp()
        try:
            ok = load().bsn_host_alloc(C.byref(p), size) == 0 and p.value
        except Exception:
            ok = False
        drop = []
        with self.lock:
            self.pending.discard(size)
            if ok:
                # (free blocks of sizes nobody asks for any more make room: the list stays below BSN_RESULT_POOL_MAX)
                while self.free and sum(sz for _, sz in self.free) + size > self.keep_max:
                    drop.append(self.free.pop(0))
                self.free.append((p.value, size))
                self.prefetched += 1
        for addr, _ in drop:
            load().bsn_host_free(vp(addr))

    def _give_back(self, addr, size):
        with self.lock:
            self.live -= size
            limit = max(self.keep, min(self.peak_live, self.keep_max))
            keep = sum(sz for _, sz in self.free) + size <= limit
            if keep:
                self.free.append((addr, size))
        if not keep:
            load().bsn_host_free(vp(addr))

    def drain(self):
        with self.lock:
            blocks, self.free = self.free, []
            self.peak_live = self.live
        for addr, _ in blocks:
            load().bsn_host_free(vp(addr))


result_pool = PinnedPool()
