"""bed_randomSVD — host mirror of R/autoSVD.R:205-219.

The reference forwards to bigstatsr::big_randomSVD(obj.bed$light, fun.scaling, ind.row,
ind.col, k, tol, verbose, ncores, bed_prodVec, bed_cprodVec); here the whole solve runs in
libbigsnpr_hip (bsn_bed_randomsvd): scaling statistics, block-Lanczos passes and panel
algebra all stay on the GPU.  The result mirrors class "big_SVD": d, u, v, niter, nops,
center, scale.
"""
import ctypes as C
import warnings

import numpy as np

from . import _lib
from ._lib import check, f64p, i64p, ptr
from .bed import _args, assert_bed, bed_scaleBinom


def bed_randomSVD(obj_bed, fun_scaling=bed_scaleBinom, ind_row=None, ind_col=None, k=10,
                  tol=1e-4, verbose=False, ncores=1, block=0, slices=0, max_basis=0, seed=1,
                  comm=None, allreduce=None, rank=0, world=1, m_total=0, return_uv=True, warm_start=0, warm_denominator=0,
                  max_restarts=0, vec_floor=0.0, exchange_timing=False, exchange_timeout_ms=0):
    """Partial SVD of the scaled matrix.  Extra (non-reference) arguments: ``block``
    (vectors per streaming pass), ``slices`` (int8 slices per fp64 value), ``vec_floor`` (relative
    residual floor wanted for the singular vectors that converge beyond ``tol``: 0 = 1e-7 unless
    ``slices`` is given, < 0 = none, i.e. every step on ``slices`` digits; include/bigsnpr_hip.h); column-sharded
    multi-GPU: ``comm`` (a bigsnpr_amd.Comm: the exchange runs inside the library over RCCL),
    ``m_total`` (columns over all ranks); tests: ``allreduce`` (callable(ptr, count) summing a
    device buffer of doubles over ranks) with ``rank`` / ``world``."""
    assert_bed(obj_bed)
    ir, ic = _args(obj_bed, ind_row, ind_col)
    opts = _lib.SvdOptions()
    if fun_scaling is bed_scaleBinom:
        # the default scaling is evaluated inside the solve: its code counts ride along the first
        # crossproduct pass (same values as bed_scaleBinom(obj_bed, ind_row, ind_col), bit for bit)
        center, scale = np.empty(ic.size), np.empty(ic.size)
        opts.binom_scaling = 1
        opts.center_out, opts.scale_out = ptr(center, f64p), ptr(scale, f64p)
    else:
        ms = fun_scaling(obj_bed, ind_row=ir, ind_col=ic, ncores=ncores)
        center = np.ascontiguousarray(ms["center"], dtype=np.float64)
        scale = np.ascontiguousarray(ms["scale"], dtype=np.float64)
    opts.k, opts.tol, opts.block, opts.slices = int(k), float(tol), int(block), int(slices)
    opts.warm_start, opts.warm_denominator = int(warm_start), int(warm_denominator)
    opts.max_restarts = int(max_restarts)
    opts.vec_floor = float(vec_floor)
    opts.exchange_timing, opts.exchange_timeout_ms = int(bool(exchange_timing)), int(exchange_timeout_ms)
    opts.max_basis, opts.seed, opts.verbose, opts.m_total = int(max_basis), int(seed), int(verbose), int(m_total)
    cb = None
    if comm is not None:
        opts.comm = comm.handle
    elif allreduce is not None:
        cb = _lib.ALLREDUCE_FN(lambda p, count, ctx: allreduce(p, count))
        opts.allreduce = cb
        opts.hook_rank, opts.hook_world = int(rank), int(world)
    info = _lib.SvdInfo()
    d = np.empty(k)
    # u and v go to page-locked memory of the library's result pool (bsn_host_alloc): written by the DMA
    # engines directly, and — once an earlier result has been collected — without a first-touch page fault
    # per 4 KB; a small result is not worth a pinned block
    big = k * (ir.size + ic.size) * 8 >= (8 << 20)
    alloc = _lib.result_pool.empty if big else np.empty
    u = alloc((k, ir.size)) if return_uv else None
    v = alloc((k, ic.size)) if return_uv else None
    L = _lib.load()
    rc = L.bsn_bed_randomsvd(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p), ic.size,
                             ptr(center, f64p), ptr(scale, f64p), C.byref(opts), ptr(d, f64p),
                             ptr(u, f64p), ptr(v, f64p), C.byref(info))
    _lib.result_pool.kick()   # (page-locks, behind this call, result blocks for the next one)
    if rc == 2:  # RSpectra::svds warns likewise when fewer than k triplets converged
        warnings.warn("bed_randomSVD did not converge: " + L.bsn_last_error().decode(), RuntimeWarning)
    else:
        check(rc)
    if info.n_bad > 0:  # src/bed-fun.cpp:40-41 (bed_colstats, called by bed_scaleBinom)
        warnings.warn("%d variants have >50%% missing values." % info.n_bad)
    return dict(d=d, u=None if u is None else u.T, v=None if v is None else v.T,
                niter=info.niter, nops=info.nops, center=center, scale=scale,
                basis=info.basis, converged=bool(info.converged),
                max_rel_resid=info.max_rel_resid, gpu_ms=info.gpu_ms,
                cprod_ms=info.cprod_ms, prod_ms=info.prod_ms, n_cprod=info.n_cprod,
                n_prod=info.n_prod, block=info.block, slices=info.slices,
                fused_stats=bool(info.fused_stats), cprod_stats_ms=info.cprod_stats_ms,
                n_cprod_stats=info.n_cprod_stats, warm_launches=info.warm_launches,
                warm_fraction=info.warm_fraction, warm_ms=info.warm_ms, tiled=int(info.tiled),
                segmented_passes=int(info.segmented_passes), compact_gathers=int(info.compact_gathers),
                slices_max=int(info.slices_max), wide_steps=int(info.wide_steps),
                wide_cprod_ms=info.wide_cprod_ms, wide_prod_ms=info.wide_prod_ms,
                n_wide_cprod=int(info.n_wide_cprod), n_wide_prod=int(info.n_wide_prod),
                lead_rel_resid=info.lead_rel_resid, compacted=bool(info.compacted), compact_ms=info.compact_ms,
                out_of_core=bool(info.out_of_core),
                na_free_steps=[float(x) for x in info.na_free_steps], na_skip=int(info.na_skip),
                exchange_mode=("none", "whole pass", "segments, one stream", "segments, reduce-scatters on a second stream")[
                    max(0, min(3, int(info.exchange_mode)))],
                exchange_ms=dict(zip(("reduce_scatter", "all_gather", "small", "exposed_wait"), [float(x) for x in info.exchange_ms])),
                n_exchange=dict(zip(("reduce_scatter", "all_gather", "small", "exposed_wait"), [int(x) for x in info.n_exchange])))
