#!/usr/bin/env python3
"""bench.py — bed_randomSVD (k = 20) on a synthetic 2-bit genotype matrix resident in HBM.

    python bench.py --gpus N --steps K --warmup W            (the headline: BASELINE.json configs[2] / [3])
    python bench.py --workload ld                            (config C5: windowed LD on one chromosome)

One "step" = one complete bed_randomSVD solve (binomial scaling statistics riding on the first
crossproduct pass + block-Lanczos passes + panel algebra) on the matrix that is already in HBM.
Metric (BASELINE.json / BASELINE.md §2): SNP-columns/s = m_total * passes / wall, passes = streaming
passes over the image actually executed (A~ / A~' panel applications; + 1 when the scaling statistics
needed a pass of their own).  N > 1: the m_total columns are sharded over the ranks (strong scaling, the
matrix of BASELINE configs[3]); the exchange (reduce-scatter of the n x 8 panel by sample blocks, small
Gram all-reduces, all-gather of the finished basis block) runs INSIDE libbigsnpr_hip over RCCL/xGMI
(bsn_comm_*); this script only carries the RCCL unique id between the ranks (gloo) and times.

Extra JSON objects: "roofline" (dominant streaming kernel: algorithmic bytes per launch / HIP-event
duration measured inside the timed solves), "cpu_baseline" (the OpenMP CPU restatement of the reference
kernels, oracle/bsn_oracle.c, on a bounded sample of the same matrix on this host; rank 0, N = 1 only)
and "ingest" (bsn_bed_open of a real .bed file of a bounded size: host -> HBM rate, BASELINE.md §2
"reported separately"; never part of `value`); "cold" (the first solves of a fresh process), "accuracy" (two matrices),
"auto_svd" (the caller of the hot path, snp_autoSVD, twice on the timed image) — all outside the timed region.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# multi-process GPU work on this pool: dmabuf IPC only (the driver exports this too; keep it in any
# environment built here), and single-node RCCL bootstraps over loopback (the container hostname may not
# resolve to a usable interface)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    # a rank still blocked 20 s after its communicator was aborted ends with status 86 (the library leaves that to its caller)
    os.environ.setdefault("BSN_WATCHDOG_EXIT", "1")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
I8_PEAK_TOPS = 5000.0      # MI355X_MICROARCH.md / SURVEY.md §8d: dense int8 MFMA peak (2x the 2.5 PFLOP/s bf16)
FP4_PEAK_TOPS = 10000.0    # MI355X_MICROARCH.md: dense FP6 / FP4 MFMA peak (block-scaled f8f6f4 forms)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["svd", "ld", "matvec"], default="svd")
    ap.add_argument("--n", "--samples", type=int, default=400000)
    ap.add_argument("--m", "--variants", type=int, default=0, help="total SNP columns over all ranks (svd: 1e6, ld: 1e5)")
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--block", type=int, default=0, help="vectors per pass (0 = library default: 8 at tol 1e-4)")
    ap.add_argument("--tol", type=float, default=1e-4)
    ap.add_argument("--slices", type=int, default=0)
    ap.add_argument("--window", type=int, default=2000, help="ld: variants per 3 cM window")
    ap.add_argument("--no-uv", action="store_true",
                    help="leave u and v on the device (diagnostic: the timed solve then ends before the 224-MB download "
                         "of the result; the default times the whole function, result on the host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-wide", action="store_true",
                    help="skip the two extra solves on 32- and 56-bit panels that fill `fp64_equivalent`")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--ind-col-fraction", type=float, default=0.0,
                    help="svd: solve over a sorted random subset of this fraction of the variants (ind.col = ind.keep of "
                         "bed_autoSVD, R/autoSVD.R:296-301) instead of all of them; one GPU")
    ap.add_argument("--exchange", choices=["overlap", "one_stream", "whole"], default=None,
                    help="sharded svd: take this exchange of the product pass (else the fastest one that passes the first-contact "
                         "probe: bigsnpr_amd.comm.negotiate)")
    ap.add_argument("--exchange-timeout-ms", type=int, default=30000,
                    help="sharded svd: watchdog of the first-contact probe and (x 4) of every later solve")
    ap.add_argument("--na16", type=int, default=655,
                    help="svd: missing genotypes per 65 536 of the synthetic image (655 = 1 %%, the headline; 0 = complete data)")
    ap.add_argument("--no-accuracy", action="store_true",
                    help="svd: skip the accuracy record (u / v of the last timed solve against a 56-bit tol-1e-10 solve, outside "
                         "the timed region)")
    ap.add_argument("--ingest-gb", type=float, default=8.0, help="size of the .bed file written and re-opened")
    ap.add_argument("--no-fallback", action="store_true",
                    help="N > 1 without a working RCCL communicator: exit with status 3 instead of taking the host all-reduce "
                         "hook over gloo (the default since round 5: a run that cannot use RCCL at all still ends with ONE line, "
                         "labelled FALLBACK in config.parallelism and carrying the reason under \"fallback\" — a number that "
                         "says what it is instead of no number)")
    ap.add_argument("--allow-fallback", action="store_true", help="(the default now; kept for older command lines)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="one GPU, N given: solve on the FIRST of N column shards (m / N variants) with the solver told the "
                         "total (--m): the per-rank work of an N-GPU run — warm start, block size and step count as there; "
                         "with --force-dist through the 1-rank RCCL communicator.  The n-side panel algebra is NOT divided "
                         "by N here (one rank owns all rows), so this is an upper bound of a rank's time without the exchange")
    ap.add_argument("--force-dist", action="store_true",
                    help="go through the RCCL communicator even with one rank (self-test)")
    ap.add_argument("--cpu-sample-cols", type=int, default=0)
    ap.add_argument("--no-autosvd", action="store_true",
                    help="skip the snp_autoSVD record (two calls on the timed image, outside the timed region)")
    ap.add_argument("--no-cold", action="store_true",
                    help="svd: skip the `cold` record (a fresh process timing the FIRST bed_randomSVD of a new handle — the call the "
                         "reference makes, R/autoSVD.R:205-219 — at full size and on a real .bed file; outside the timed region)")
    ap.add_argument("--cold-child", default="", help=argparse.SUPPRESS)   # internal: "synthetic" | path of a .bed (n x m as given)
    ap.add_argument("--verbose", type=int, default=0, help="1: residual trajectory of every solve on stderr")
    ap.add_argument("--warm-start", type=int, default=0, help="warm-start iterations (0 = library default 1, -1 = none)")
    ap.add_argument("--warm-den", type=int, default=0, help="warm start on the leading 1/N of the variants (0 = 16)")
    return ap.parse_args()


def host_threads():
    """threads this process may really use: affinity mask and cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    eff = n if quota is None else max(1, min(n, int(quota)))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return dict(affinity=n, cpu_count=os.cpu_count(), cgroup_quota=quota, effective=eff, model=model)


def log(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.time() - T_START, msg))
    sys.stderr.flush()


T_START = time.time()


def cold_child(a):
    """(internal) what a caller's first call costs, in a process of its own: library load, the handle (image generated on the
    device, or bsn_bed_open of a real file), then the wall time of the first five bed_randomSVD calls on it (the sample-major copy
    is allocated and made by a helper thread behind the first: on a box whose driver clears 100 GB slowly that allocation is
    still running under the second call and holds up its launches — five calls show the steady state either way).  One JSON line."""
    t_proc = time.perf_counter()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import bigsnpr_amd as ba
    from bigsnpr_amd import _lib
    L = _lib.load()
    ba.selftest()                       # first contact with the device: runtime and code objects
    t_lib = time.perf_counter() - t_proc
    t0 = time.perf_counter()
    if a.cold_child == "synthetic":
        gb = ba.bed.synthetic(a.n, a.m or 1000000, seed=20250905, na16=a.na16)
    else:
        import ctypes as C
        h = C.c_void_p()
        _lib.check(L.bsn_bed_open(a.cold_child.encode(), a.n, a.m, C.byref(h)))
        gb = ba.bed(_handle=h, _n=a.n, _m=a.m)
    L.bsn_device_sync()
    t_handle = time.perf_counter() - t0
    ms, infos = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        r = ba.bed_randomSVD(gb, k=a.k, tol=a.tol)
        ms.append(1e3 * (time.perf_counter() - t0))
        infos.append({"niter": r["niter"], "gpu_ms": r["gpu_ms"], "product_launches_on_k_prod": r["n_prod"] + r["n_wide_prod"] if r["tiled"] != 2 else None,
                      "image_layout_at_exit": {0: "variant-major only", 1: "tiled copy", 2: "sample-major copy"}[r["tiled"]],
                      "n_prod": r["n_prod"], "n_wide_prod": r["n_wide_prod"], "prod_ms": r["prod_ms"], "wide_prod_ms": r["wide_prod_ms"],
                      "sigma1": float(r["d"][0])})
        del r
    real_stdout.write(json.dumps({"library_and_runtime_s": t_lib, "handle_s": t_handle, "solve_ms": ms, "solves": infos}) + "\n")
    real_stdout.flush()


def cold_record(a, bed_path=None, bed_n=0, bed_m=0):
    """the reference's unit of work is ONE bed_randomSVD on a freshly opened object.  A child process (fresh
    runtime, fresh handle, nothing allocated) times that call; reported beside the warm number, never part of `value`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cold-child", bed_path or "synthetic", "--n", str(bed_n or a.n),
           "--m", str(bed_m or a.m or 1000000), "--k", str(a.k), "--tol", str(a.tol), "--na16", str(a.na16)]
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": str(e)[:300]}
    first, warm = d["solve_ms"][0], min(d["solve_ms"][1:])
    d.update({"first_solve_ms": first, "second_solve_ms": d["solve_ms"][1], "warm_solve_ms": warm, "first_minus_warm_ms": first - warm})
    return d


def main():
    a = parse()
    if a.cold_child:
        return cold_child(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    # RCCL prints a version banner through C stdio on stdout; keep the real stdout for the one
    # JSON line and send everything else (fd 1) to stderr.
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    dist = torch = None
    if world > 1:
        # torch first: the library then binds to the same HIP runtime as torch.  torch.distributed is
        # only the rendezvous here (gloo: unique id broadcast, barriers, max over ranks of the wall time)
        import torch
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29511"
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif not os.environ.get("BSN_BENCH_NO_TORCH"):
        try:
            import torch
        except Exception:
            torch = None
    log("torch %s" % ("imported" if torch is not None else "not imported"))
    import bigsnpr_amd as ba
    from bigsnpr_amd import _lib
    L = _lib.load()
    # one process per GPU; if the launcher already narrowed the visible devices to one per process
    # (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES), the only device is index 0
    ndev = ba.device_count()
    if ndev < 1:
        raise SystemExit("no HIP device is visible")
    device = local_rank % ndev
    _lib.check(L.bsn_set_device(device))
    if torch is not None and torch.cuda.is_available():
        torch.cuda.set_device(device)
    ba.selftest()
    log("selftest ok")
    if a.workload in ("ld", "matvec"):
        out = (ld_bench if a.workload == "ld" else matvec_bench)(a, ba, L, rank, world, dist, torch)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            real_stdout.write(json.dumps(out) + "\n")
            real_stdout.flush()
        return

    comm = None
    hook = None
    exchange_report = None
    fallback_reason = None

    def host_bcast(obj):                # rank 0's object on every rank (gloo)
        box = [obj]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return box[0]

    def host_min(i):                    # all ranks take the same path: the minimum of their flags (gloo)
        if world == 1:
            return int(i)
        import torch as _t0
        t = _t0.tensor([int(i)])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    if world > 1 or a.force_dist:
        err = None
        # First contact with the transport: a MINIATURE sharded solve in each exchange mode — segments of
        # the product pass with their reduce-scatters on a second stream, the same on one stream, the whole pass + one
        # reduce-scatter — under the library's watchdog, compared bit for bit, the ranks agreeing over gloo; the timed
        # solves then run in the fastest mode that survived (exchange.mode in the JSON), with a fresh communicator if
        # one had to be aborted.  A communicator that cannot even carry the plainest pattern is found here too.
        try:
            comm, exchange_report = ba.comm.negotiate(rank, world, host_bcast, host_min, timeout_ms=a.exchange_timeout_ms,
                                                      modes=[a.exchange] if a.exchange else None,
                                                      log=lambda msg: log(msg) if rank == 0 else None)
            log("exchange of the sharded solve: %s" % exchange_report["mode"])
        except Exception as e:
            err, comm = e, None
        if world > 1:                   # all ranks take the same path
            import torch as _t
            ok = _t.tensor([1 if comm is not None else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:
                    comm.close()
                comm = None
                if a.no_fallback:
                    # a scaling number measured over gloo and the host is not an RCCL / xGMI number
                    log("in-library RCCL communicator unavailable (%s); not falling back (--no-fallback)" % err)
                    dist.barrier()
                    dist.destroy_process_group()
                    sys.exit(3)
                log("in-library RCCL communicator unavailable (%s): falling back to the host all-reduce hook (gloo) — the "
                    "line of this run is NOT an RCCL / xGMI number and says so" % err)
                fallback_reason = str(err) if err is not None else "another rank has no communicator"
                import ctypes as _C
                import numpy as _np

                def hook(ptr, count):
                    host = _np.empty(count)
                    _lib.check(L.bsn_memcpy_d2h(host.ctypes.data_as(_C.c_void_p), _C.c_void_p(ptr), count * 8))
                    t = _t.from_numpy(host)
                    dist.all_reduce(t)
                    _lib.check(L.bsn_memcpy_h2d(_C.c_void_p(ptr), host.ctypes.data_as(_C.c_void_p), count * 8))
        elif comm is None:
            raise err

    cold = None
    if (world == 1 and rank == 0 and a.workload == "svd" and not a.no_cold and a.shard_of <= 1 and a.ind_col_fraction <= 0
            and not a.force_dist):
        cold = {"what": "a FRESH process: library + runtime, a new handle, then the first five bed_randomSVD(k = %d) calls on it; "
                        "the first solve runs on the variant-major image alone; the sample-major copy is made behind it, for the later ones" % a.k,
                "synthetic_full_size": cold_record(a)}
        log("cold record (full size) done: first solve %s ms, warm %s ms" % (
            cold["synthetic_full_size"].get("first_solve_ms"), cold["synthetic_full_size"].get("warm_solve_ms")))
    n, m_total = a.n, a.m or 1000000
    j0 = (m_total * rank) // world
    j1 = (m_total * (rank + 1)) // world
    if a.shard_of > 1 and world == 1:
        j0, j1 = 0, m_total // a.shard_of
    m_local = j1 - j0
    t0 = time.time()
    gb = ba.bed.synthetic(n, m_local, seed=20250905, j_begin=j0, na16=a.na16)
    L.bsn_device_sync()
    gen_s = time.time() - t0
    log("image generated (%.2f s)" % gen_s)
    # The sample-major second copy of the image (what the product passes of a 16-vector solve read) is built once per
    # handle — by the first such solve, or ahead of time like here, so that a run without warm-up solves does not time
    # the one-off 100-GB allocation + transposition inside its first step.  Reported, never part of `value`.
    smaj_s = None
    # (the library's rule for the default block: 16 vectors when 4 k >= 56 and the warm start applies)
    two_blocks = a.block == 16 or (a.block == 0 and 4 * a.k >= 56 and m_total >= 262144 and a.warm_start >= 0)
    if two_blocks and a.slices in (0, 2) and not os.environ.get("BSN_NO_SMAJ") and a.ind_col_fraction <= 0:
        t0 = time.time()
        built = gb.sample_major()
        L.bsn_device_sync()
        smaj_s = (time.time() - t0) if built else None
        log("sample-major copy %s (%.2f s)" % ("built" if built else "not built (no room)", time.time() - t0))

    def sync():
        L.bsn_device_sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            L.bsn_device_sync()

    ind_col = None
    m_image = m_local
    if a.ind_col_fraction > 0 and world == 1:
        import numpy as _npf
        ind_col = _npf.sort(_npf.random.default_rng(20250905).choice(m_local, int(round(a.ind_col_fraction * m_local)), replace=False))
        m_local = m_total = int(ind_col.size)        # the columns a pass streams
        log("ind.col: %d of %d variants (sorted random subset)" % (m_local, m_image))

    def step():
        return ba.bed_randomSVD(gb, ind_col=ind_col, k=a.k, tol=a.tol, block=a.block, slices=a.slices, comm=comm,
                                allreduce=hook, rank=rank, world=world,
                                m_total=m_total, return_uv=not a.no_uv, verbose=a.verbose, warm_start=a.warm_start,
                                warm_denominator=a.warm_den,
                                # sharded: every collective bracketed by HIP events (exchange.ms in the JSON), and a watchdog
                                # that turns a collective which never completes into an error
                                exchange_timing=comm is not None,
                                exchange_timeout_ms=4 * a.exchange_timeout_ms if comm is not None else 0)

    first_compact_ms = 0.0
    if ind_col is not None:        # one untimed solve so that the one-off copy is reported apart even with --warmup 0
        first_compact_ms = step()["compact_ms"]
    # The warm-up solves are the first FULL-SIZE contact of the chosen exchange (the probe above moved 1 / 100 of the
    # bytes): if one fails on any rank — watchdog, RCCL error — every rank drops to the next more conservative mode with
    # a fresh communicator and warms up again; only the whole-pass exchange failing ends the run (status 4).
    while True:
        ok, werr = 1, None
        try:
            for _ in range(a.warmup):
                step()
        except Exception as e:
            ok, werr = 0, e
        if comm is None:
            if not ok:
                raise werr
            break
        if host_min(ok):
            break
        mode_now = exchange_report["mode"]
        nxt = {"overlap": "one_stream", "one_stream": "whole"}.get(mode_now)
        log("a warm-up solve failed in exchange mode %s (%s)%s" % (mode_now, werr, "" if nxt is None else ": falling back to " + nxt))
        exchange_report.setdefault("full_size_failures", []).append(dict(mode=mode_now, error=None if werr is None else str(werr)[:300]))
        comm.close()
        if nxt is None:
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            sys.exit(4)
        uid = host_bcast(ba.Comm.unique_id() if rank == 0 else None)
        comm = ba.Comm(uid, rank, world)
        ba.set_exchange_mode(nxt)
        exchange_report["mode"] = nxt
    sync()
    log("warmup done")
    t0 = time.perf_counter()
    infos, last_uv = [], None
    for _ in range(a.steps):
        r = step()
        # like `svd <- bed_randomSVD(...)` in a loop: the previous result is dropped when the new one is bound
        # (its page-locked blocks go back to the library's result pool)
        last_uv = (r.pop("u"), r.pop("v"))
        infos.append(r)
    sync()
    wall = time.perf_counter() - t0
    log("timed solves done (%.1f ms per solve)" % (1e3 * wall / a.steps))
    per_rank = None
    if world > 1:
        # every rank's own wall time and streaming-kernel averages travel to rank 0 (gloo): the JSON names the slowest
        # rank and what ITS kernels and collectives cost, not only rank 0's
        mine = dict(rank=rank, ms_per_step=1e3 * wall / a.steps,
                    kernels={key: sum(r[key + "_ms"] for r in infos) / max(1, sum(r["n_" + key] for r in infos))
                             for key in ("prod", "cprod", "cprod_stats", "wide_prod", "wide_cprod")
                             if sum(r["n_" + key] for r in infos)},
                    exchange_ms={c: sum(r["exchange_ms"][c] for r in infos) / a.steps for c in infos[-1]["exchange_ms"]})
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # streaming passes over the image: the A~ / A~' panel applications, + 1 when the scaling statistics
    # were a pass of their own (they ride along the first crossproduct pass otherwise)
    # warm-start launches stream only a fraction of the variants: counted as that fraction of a pass
    passes = sum(r["nops"] - r["warm_launches"] * (1.0 - r["warm_fraction"]) + (0 if r["fused_stats"] else 1)
                 for r in infos)
    m_job = m_local if (a.shard_of > 1 and world == 1) else m_total   # columns this job really streamed per pass
    value = m_job * passes / wall                             # whole job, all ranks
    bytes_per_launch = ((n + 3) // 4) * m_local               # algorithmic: 2-bit payload of the shard
    kern = {}
    for key, name in (("prod", "k_prod / k_prodT (A~ panel, contraction over variants)"),
                      ("cprod", "k_cprod (A~' panel, contraction over samples)"),
                      ("cprod_stats", "k_cprod<STATS> (first A~' pass of a solve: also counts the codes of every variant)"),
                      ("wide_prod", "k_prodT<3> (A~ panel on 24-bit digits: 16 vectors x 3 slices = three column blocks)"),
                      ("wide_cprod", "k_cprod<3> (A~' panel on 24-bit digits: three column blocks)")):
        ms = sum(r[key + "_ms"] for r in infos)
        cnt = sum(r["n_" + key] for r in infos)
        if cnt:
            kern[key] = dict(name=name, total_ms=ms, launches=cnt, avg_ms=ms / cnt)
    dom_key = max(kern, key=lambda k_: kern[k_]["total_ms"])
    dom = kern[dom_key]
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside
    # the process, so this is the committed rocprofv3 measurement of the same workload
    traffic, traffic_note, clock_ghz = None, None, None
    blk, sl = infos[-1]["block"], infos[-1]["slices"]
    running = streaming_kernel_names(L, gb)  # what this build launched in the last solve, by kind
    run_key = {"wide_prod": "prod_wide", "wide_cprod": "cprod_wide"}.get(dom_key, dom_key)

    def nb_from_name(kind):   # MFMA column blocks of the kernel launched under a kind: its first template argument
        import re
        mm = re.search(r"<\s*(\d+)", running.get({"wide_prod": "prod_wide", "wide_cprod": "cprod_wide"}.get(kind, kind), ""))
        return int(mm.group(1)) if mm else (3 if kind.startswith("wide_") else (1 if blk * sl <= 16 else 2))
    nb_of = {kind: nb_from_name(kind) for kind in kern}
    nb = nb_of[dom_key]   # (three: the 24-bit passes of the precision schedule)
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        # the counters are per kernel variant: one column block (block x slices <= 16), two or three.  The record is only
        # quoted when it was taken on the SAME kernel instantiation built from the SAME source as the running library
        rec = pm["kernels" if nb == 1 else "kernels_nb%d" % nb][dom_key]
        if pm["workload"] != {"n": n, "m_per_gpu": m_local}:
            traffic_note = "profiles/pmc_traffic.json was taken on another workload"
        elif pm.get("matvec_sha256") != source_sha256("matvec.hip"):
            traffic_note = "profiles/pmc_traffic.json was taken on another build of matvec.hip"
        elif norm_kernel(rec["name"]) != norm_kernel(running.get(run_key, "")):
            traffic_note = ("profiles/pmc_traffic.json holds %s, this build launched %s"
                            % (rec["name"], running.get(run_key)))
        else:
            traffic = rec["hbm_read_bytes"]
            clock_ghz = rec.get("effective_GHz")   # GRBM_GUI_ACTIVE / 8 XCDs / kernel time, same pass
    except Exception as e:
        traffic, traffic_note = None, "no usable profiles/pmc_traffic.json (%s)" % e
    achieved = bytes_per_launch / (dom["avg_ms"] * 1e-3) / 1e9
    # the same launch priced against the matrix pipe: per 16 variants x 64 samples one v_mfma_i32_16x16x64_i8
    # (32 768 int8 ops) per plane (genotype, missing-value) and column block
    def mfma_price(nblk, ms):
        return (n / 64.0) * (m_local / 16.0) * 2 * nblk * 32768.0 / (ms * 1e-3) / 1e12
    mfma_tops = mfma_price(nb, dom["avg_ms"])
    hbm_obj = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    mfma_obj = {"achieved": mfma_tops, "peak": I8_PEAK_TOPS, "unit": "TOP/s", "frac": mfma_tops / I8_PEAK_TOPS, "column_blocks": nb}
    if clock_ghz:
        # the matrix pipe at the clock the power cap leaves (MI355X_MICROARCH.md: 1 024 SIMDs x 2 048 int8 ops per cycle)
        mfma_obj["clock_GHz"] = clock_ghz
        mfma_obj["frac_at_clock"] = mfma_tops / (1024 * 2048 * clock_ghz * 1e9 / 1e12)
    # which roofline binds: the resource with the higher utilisation; "power" when the clock sits > 15 % below 2.4 GHz
    # (the launch is then paced by the package power cap: cycles / clock)
    primary = ("mfma", mfma_obj) if mfma_obj["frac"] > hbm_obj["frac"] else ("hbm", hbm_obj)
    out = {
        "metric": "SNP-cols/sec for bed_randomSVD k=%d (m*passes/wall)" % a.k,
        "value": value, "unit": "SNP-cols/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        # what a caller pays for one bed_randomSVD: d, u, v, center, scale on the host (u / v are part of the
        # timed region unless --no-uv)
        "time_to_solution_ms": wall / a.steps * 1e3,
        "result_on_host": ["d", "center", "scale"] + ([] if a.no_uv else ["u", "v"]),
        "vs_baseline": None,
        "dtype": "i8 (2-bit codes x %s fixed-point image of the fp64 basis in int8 slices, exact int32 "
                 "MFMA accumulation; fp64 panel algebra and Rayleigh-Ritz)"
                 % ("%d-bit" % (8 * sl) if infos[-1]["slices_max"] <= sl else
                    "%d-bit (early block steps) / %d-bit (late block steps)" % (8 * infos[-1]["slices_max"], 8 * sl)),
        "data": "synthetic",
        "config": {"workload": "bed_randomSVD k=%d on synthetic %dx%d 2-bit .bed image resident in HBM"
                               % (a.k, n, m_total),
                   "n": n, "m_total": m_total, "m_per_gpu": m_local, "block": blk, "slices": sl, "tol": a.tol,
                   "parallelism": ("columns sharded x%d; in-library RCCL: reduce-scatter of the n x %d panel by "
                                   "sample blocks, b x p Gram all-reduces, all-gather of the basis block"
                                   % (world, blk)) if comm else
                                  ("columns sharded x%d; FALLBACK: host all-reduce hook over gloo (RCCL communicator "
                                   "unavailable)" % world if hook else
                                   ("single GPU holding shard 1 of %d (per-rank work of an %d-GPU run, no exchange)"
                                    % (a.shard_of, a.shard_of) if a.shard_of > 1 else "single GPU"))},
        "passes_per_solve": passes / a.steps,
        # N > 1 without any working RCCL exchange: the panels went through the host and gloo — not an xGMI measurement
        "fallback": None if hook is None else {"transport": "host all-reduce over gloo (device -> host -> gloo -> device per panel)",
                                               "reason": fallback_reason},
        "ind_col": None if ind_col is None else {
            "selected_variants": m_local, "of": m_image,
            "path": ("compacted copy of the selection (gathered once, %.1f ms in the first solve; the timed solves found it on the "
                     "handle)" % first_compact_ms) if infos[-1].get("compacted") else "gather lists on the full image"},
        # sharded solve: product passes cut into segments whose reduce-scatters overlap the next segment, basis blocks
        # all-gathered as int16 (0 / 0 on one GPU without a communicator)
        "exchange": exchange_record(infos, a.steps, exchange_report, per_rank),
        "niter": infos[-1]["niter"], "converged": infos[-1]["converged"],
        "scaling_statistics": "ride along the first crossproduct pass" if infos[-1]["fused_stats"] else "own pass",
        # share of the streaming kernels' K-steps (1 024 genotypes) without a missing code, sampled on the image, and
        # whether the passes took the kernels that skip the missing-value plane for such steps (bit 0 crossproduct,
        # bit 1 product): never at 1 % scattered missing values
        "missing_values": {"per_65536": a.na16, "steps_without": infos[-1]["na_free_steps"],
                           "skipping_kernels": infos[-1]["na_skip"]},
        "warm_start": {"launches": infos[-1]["warm_launches"], "fraction_of_variants": infos[-1]["warm_fraction"],
                       "ms": infos[-1]["warm_ms"]},
        "image_layout": ("streaming kernels read the tiled second copy (64 variants x 1024 samples per 16-KB tile; built once "
                         "per handle before the first solve, + %.0f GB of HBM)" % (bytes_per_launch / 1e9))
                        if infos[-1]["tiled"] == 1 else
                        ("crossproduct passes read the variant-major image, product passes its sample-major second copy "
                         "(k_prodT; built once per handle before the first solve: one read + one write pass, + %.0f GB of HBM)"
                         % (bytes_per_launch / 1e9)) if infos[-1]["tiled"] == 2 else "variant-major image only",
        "end_to_end_cols_per_s": m_job * a.steps / wall,
        "hbm_GBps_whole_solve": passes * ((n + 3) // 4) * m_job / wall / 1e9,
        "hbm_frac_whole_solve": passes * ((n + 3) // 4) * m_job / wall / 1e9 / HBM_PEAK_GBS / world,
        "sigma": [float(x) for x in infos[-1]["d"][:5]],
        "generate_s": gen_s,
        "sample_major_copy_build_s": smaj_s,
        "roofline": {"bound": primary[0], "kernel": dom["name"], "achieved": primary[1]["achieved"], "peak": primary[1]["peak"],
                     "unit": primary[1]["unit"], "frac": primary[1]["frac"], "traffic": traffic,
                     "paced_by": ("power cap (clock %.2f GHz under this kernel against 2.4)" % clock_ghz)
                                 if clock_ghz and clock_ghz < 0.85 * 2.4 else None,
                     "hbm": hbm_obj,
                     "traffic_source": ("profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE x2, bytes per launch; same kernel "
                                        "instantiation and matvec.hip hash as this build)") if traffic else traffic_note,
                     "kernels_launched": running,
                     "bytes_per_launch": bytes_per_launch, "avg_launch_ms": dom["avg_ms"],
                     "launches": dom["launches"],
                     # the two-column-block kernels (16 vectors per pass) are bound by the matrix pipe / the VALU issue
                     # port next to it, not by HBM: both prices of the same launch
                     "mfma": mfma_obj,
                     "other": {k: {"avg_ms": v["avg_ms"], "launches": v["launches"],
                                   "GBps": bytes_per_launch / (v["avg_ms"] * 1e-3) / 1e9,
                                   "column_blocks": nb_of[k], "mfma_TOPs": mfma_price(nb_of[k], v["avg_ms"])}
                               for k, v in kern.items()}},
    }

    if rank == 0 and world == 1:
        ref = None
        if not a.no_accuracy and not a.no_uv and a.shard_of <= 1:
            out["accuracy"], ref = accuracy(ba, gb, a, infos[-1], last_uv)
            log("accuracy against the 56-bit tol-1e-10 solve done")
        if not a.no_wide:
            out["fp64_equivalent"] = wide_solves(ba, gb, a, infos[-1], sync, ref)
            log("alternative-precision solves done")
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ba, gb, n, a.cpu_sample_cols)
            log("cpu baseline done")
        if not a.no_autosvd and a.shard_of <= 1 and a.ind_col_fraction <= 0:
            out["auto_svd"] = autosvd_record(ba, gb, m_local)
            log("snp_autoSVD record done")
        if not a.no_ingest:
            out["ingest"] = ingest(ba, L, n, a.ingest_gb, cold_hook=(lambda path, nn, mm: cold_record(a, path, nn, mm)) if cold else None)
            if cold is not None and isinstance(out["ingest"], dict) and "cold" in out["ingest"]:
                cold["real_bed_file"] = out["ingest"].pop("cold")
            log("ingest done")
        if cold is not None:
            out["cold"] = cold
        if "accuracy" in out and a.ind_col_fraction <= 0:
            # the accuracy record on a SECOND matrix (another seed of the generator), the worse of the two
            # reported.  The timed image goes first (two images, a copy and a workspace do not fit one GPU).
            try:
                gb.close()
                gb2 = ba.bed.synthetic(n, m_local, seed=7, na16=a.na16)
                r2 = ba.bed_randomSVD(gb2, k=a.k, tol=a.tol, block=a.block, slices=a.slices, warm_start=a.warm_start,
                                      warm_denominator=a.warm_den)
                rec2, _ = accuracy(ba, gb2, a, r2, (r2["u"], r2["v"]))
                gb2.close()
                first = {key: out["accuracy"][key] for key in ("u_leading_half", "u_all", "v_leading_half", "v_all")}
                out["accuracy"]["second_matrix"] = {"generator_seed": 7, **{key: rec2[key] for key in first},
                                                    "residual_estimate": rec2["residual_estimate"]}
                out["accuracy"]["worse_of_two_matrices"] = {key: max(first[key], rec2[key]) for key in first}
                out["accuracy"]["leading_half_within_tolerance"] = bool(
                    max(first["u_leading_half"], rec2["u_leading_half"]) <= 1e-6 and
                    max(first["v_leading_half"], rec2["v_leading_half"]) <= 1e-6)
                log("accuracy on a second matrix done")
            except Exception as e:
                out["accuracy"]["second_matrix"] = {"error": str(e)[:300]}
    if comm is not None:
        sync()
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stderr.flush()
    if rank == 0:
        real_stdout.write(json.dumps(out) + "\n")   # the ONE stdout line
        real_stdout.flush()


def exchange_record(infos, steps, report, per_rank):
    """what the sharded solve exchanged and what it cost: the mode that ran (and how it was chosen), HIP-event
    time per solve and number of collectives by class — reduce-scatters of the panel / of its segments, all-gathers of a basis
    block and of u, the small all-reduces / all-gathers, and the time the solve's stream WAITED for the exchange stream (what of
    the overlapped reduce-scatters stayed exposed) —, and per rank: wall time per solve, the slowest rank's kernel averages"""
    last = infos[-1]
    rec = {"mode": last.get("exchange_mode", "none"),
           "segmented_product_passes": last.get("segmented_passes", 0),
           "compact_all_gathers": last.get("compact_gathers", 0)}
    if report is not None:
        rec["first_contact"] = report
    if any(last.get("n_exchange", {}).values()):
        rec["ms_per_solve"] = {c: sum(r["exchange_ms"][c] for r in infos) / steps for c in last["exchange_ms"]}
        rec["collectives_per_solve"] = {c: sum(r["n_exchange"][c] for r in infos) / steps for c in last["n_exchange"]}
        # on the solve's stream and therefore exposed: everything but the reduce-scatters of an overlapped pass, + the wait
        over = last.get("exchange_mode", "").endswith("second stream")
        m = rec["ms_per_solve"]
        rec["exposed_ms_per_solve"] = m["all_gather"] + m["small"] + (m["exposed_wait"] if over else m["reduce_scatter"])
        rec["hidden_ms_per_solve"] = max(0.0, m["reduce_scatter"] - m["exposed_wait"]) if over else 0.0
    if per_rank:
        ms = [p["ms_per_step"] for p in per_rank]
        slow = per_rank[max(range(len(ms)), key=lambda i: ms[i])]
        rec["per_rank"] = {"ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "ms_per_step": ms,
                           "slowest_rank": slow["rank"], "slowest_rank_kernels_avg_ms": slow["kernels"],
                           "slowest_rank_exchange_ms_per_solve": slow["exchange_ms"]}
    return rec


def norm_kernel(name):
    """kernel name as rocprofv3 / the demangler print it -> comparable form"""
    name = name.split("(")[0]
    for junk in ("void ", "bsn::", " "):
        name = name.replace(junk, "")
    return name


def source_sha256(fname):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "bigsnpr_amd", "csrc", fname), "rb").read()).hexdigest()
    except OSError:
        return None


def streaming_kernel_names(L, gb):
    import ctypes as C
    buf = C.create_string_buffer(4096)
    if L.bsn_bed_streaming_kernels(gb.handle, buf, 4096) != 0:
        return {}
    return dict(line.split("=", 1) for line in buf.value.decode().splitlines() if "=" in line)


def _angles(x, ref):
    """per column || x sign - ref || = 2 sin(theta / 2): the angle between a computed singular vector and its reference"""
    import numpy as np
    s = np.sign(np.sum(x * ref, axis=0))
    return np.linalg.norm(x * s - ref, axis=0)


def _angle_record(r_u, r_v, ref, k):
    h = (k + 1) // 2
    au, av = _angles(r_u, ref["u"]), _angles(r_v, ref["v"])
    return {"u_leading_half": float(au[:h].max()), "u_all": float(au.max()),
            "v_leading_half": float(av[:h].max()), "v_all": float(av.max())}


def accuracy(ba, gb, a, info, uv):
    """what the timed configuration delivers, measured OUTSIDE the timed region — u and v of the last
    timed solve against a solve of the same matrix on 56-bit panels to tol 1e-10 with another block size (the
    full-size stand-in for the reference's fp64 Lanczos: no oracle runs at this size), per vector the angle
    || x sign - x_ref ||, worst over the leading half of the k vectors and over all of them; north_star asks for 1e-6."""
    import numpy as np
    t0 = time.perf_counter()
    ref = ba.bed_randomSVD(gb, k=a.k, tol=1e-10, slices=7, block=4 if a.block != 4 else 5)
    rec = _angle_record(uv[0], uv[1], ref, a.k)
    rec.update({"reference": "bed_randomSVD of the same matrix, 56-bit panels, tol 1e-10, block %d (%d block steps, %.1f s)"
                             % (ref["block"], ref["niter"], time.perf_counter() - t0),
                "d_max_rel_diff": float(np.max(np.abs(np.asarray(info["d"]) / ref["d"] - 1.0))),
                "residual_estimate": {"leading_half": info["lead_rel_resid"], "all": info["max_rel_resid"]},
                "precision_schedule": {"panel_bits": 8 * info["slices"], "widest_panel_bits": 8 * info["slices_max"],
                                       "wide_block_steps": info["wide_steps"], "of_block_steps": info["niter"],
                                       "three_block_launches": info["n_wide_cprod"] + info["n_wide_prod"]},
                "north_star_tolerance": 1e-6,
                "leading_half_within_tolerance": bool(rec["u_leading_half"] <= 1e-6 and rec["v_leading_half"] <= 1e-6)})
    return rec, ref


def wide_solves(ba, gb, a, default_info, sync, ref=None):
    """The accuracy / cost frontier around the timed configuration: the SAME solve (a) with every
    block step on the narrow panels of the default (round 4's default: no precision schedule), (b) / (c) with the panels
    carried at 32 and 56 bits throughout (56 bits is the width at which the products are as exact as the reference's
    own fp64 summation) — one timed solve each after one untimed, the relative difference of d to the default solve and,
    when the tight reference solve is at hand, the angles of ITS u / v to that reference."""
    import numpy as np
    res = {"default": {"slices": default_info["slices"], "slices_max": default_info["slices_max"],
                       "wide_block_steps": default_info["wide_steps"], "block": default_info["block"]}}
    d0 = np.asarray(default_info["d"])
    for tag, kw in (("narrow_panels_every_step", dict(vec_floor=-1.0, slices=a.slices)), ("slices_4", dict(slices=4)),
                    ("slices_7", dict(slices=7))):
        def one():
            return ba.bed_randomSVD(gb, k=a.k, tol=a.tol, block=a.block, return_uv=not a.no_uv,
                                    warm_start=a.warm_start, warm_denominator=a.warm_den, **kw)
        one()
        sync()
        t0 = time.perf_counter()
        r = one()
        sync()
        ms = (time.perf_counter() - t0) * 1e3
        passes = r["nops"] - r["warm_launches"] * (1.0 - r["warm_fraction"]) + (0 if r["fused_stats"] else 1)
        res[tag] = {"panel_bits": 8 * r["slices"], "ms": ms, "block": r["block"], "niter": r["niter"],
                    "passes": passes, "converged": r["converged"],
                    "max_rel_diff_d_vs_default": float(np.max(np.abs(np.asarray(r["d"]) / d0 - 1.0)))}
        if ref is not None and r.get("u") is not None:
            res[tag]["angles_to_reference"] = _angle_record(r["u"], r["v"], ref, a.k)
        del r
    return res


def autosvd_record(ba, gb, m):
    """The caller of the hot path (SURVEY section 8 'next': R/autoSVD.R:67-186): snp_autoSVD(k = 10) on the timed image with 22
    chromosomes of equal length and 500-kb windows of 250 variants a side — MAF filter, clumping, partial SVD, outlier step —
    twice; the second call is the steady state (the first pays imports and first launches).  Outside the timed region."""
    import numpy as np
    from bigsnpr_amd import autosvd as av
    try:
        chrom = np.repeat(np.arange(1, 23), (m + 21) // 22)[:m]
        pos = np.arange(m) * 2000.0
        T = {}

        def timed(name, fn):
            def w(*x, **k):
                t0 = time.perf_counter()
                r = fn(*x, **k)
                T[name] = T.get(name, 0.0) + time.perf_counter() - t0
                return r
            return w
        names = ("snp_MAF", "snp_clumping", "big_randomSVD", "dist_ogk", "rollmean_groups", "tukey_mc_up")
        saved = {nm: getattr(av, nm) for nm in names}
        calls = []
        try:
            for nm in names:
                setattr(av, nm, timed(nm, saved[nm]))
            for _ in range(2):
                T.clear()
                t0 = time.perf_counter()
                res = ba.snp_autoSVD(gb, chrom, pos, k=10, verbose=False)
                calls.append((time.perf_counter() - t0, dict(T), int(res["subset"].size)))
        finally:
            for nm in names:
                setattr(av, nm, saved[nm])
        return {"what": "snp_autoSVD(k = 10, 22 chromosomes, 500-kb windows of 250 variants) on the timed image, two calls; outside the timed region",
                "first_call_s": calls[0][0], "second_call_s": calls[1][0],
                "stages_second_call_s": {k_: round(v, 4) for k_, v in calls[1][1].items()},
                "kept_variants": calls[1][2], "variants": int(m)}
    except Exception as e:
        return {"error": str(e)[:300]}


def cpu_baseline(ba, gb, n, sample_cols):
    """The reference's OpenMP kernels restated in C (oracle/bsn_oracle.c: orc_pMatVec4,
    orc_cpMatVec4, orc_bed_colstats; src/bed-prod-vec.cpp:29-51,73-94) on the first `m_s` columns of
    the same matrix, on the threads this process may really use.  Reported as SNP-cols/s of one pass
    averaged over the two pass kinds a solve is made of (A x, A' x)."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    ht = host_threads()
    thr = ht["effective"]
    # bounded sample: ~10-30 s of CPU work at ~1-3 ns per genotype per thread-pass (1.1 / 0.9)
    m_s = sample_cols or int(max(256, min(gb.ncol, 6e9 * thr / n)))
    m_s -= m_s % 4
    n_byte = (n + 3) // 4
    src = download_cols(ba, gb, m_s)
    payload = np.empty(n_byte * m_s, dtype=np.uint8)       # pages first touched by the threads that read them
    orc.lib().orc_parallel_copy(payload.ctypes.data_as(C.POINTER(C.c_uint8)), src.ctypes.data_as(C.POINTER(C.c_uint8)),
                                C.c_int64(n_byte), C.c_int64(m_s), thr)
    del src
    ob = _bedfile_nocopy(orc, payload, n, m_s)
    st = orc.bed_colstats(ob, ncores=thr)
    af = st["sumX"] / (2.0 * st["nb_nona_col"])
    sc = dict(center=2 * af, scale=np.sqrt(2 * af * (1 - af)))
    scale = np.where(sc["scale"] > 0, sc["scale"], 1.0)
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=m_s), rng.normal(size=n)
    t = {}
    t0 = time.perf_counter(); orc.bed_colstats(ob, ncores=thr); t["colstats"] = time.perf_counter() - t0
    t0 = time.perf_counter(); orc.bed_prodVec(ob, x, None, None, sc["center"], scale, thr); t["prodVec"] = time.perf_counter() - t0
    t0 = time.perf_counter(); orc.bed_cprodVec(ob, y, None, None, sc["center"], scale, thr); t["cprodVec"] = time.perf_counter() - t0
    per_pass = (t["prodVec"] + t["cprodVec"]) / 2
    ns = {k: 1e9 * v * thr / (float(n) * m_s) for k, v in t.items()}
    res = {"value": m_s / per_pass, "unit": "SNP-cols/s", "cores": thr, "kind": "port",
           "threads_effective": thr, "host": ht,
           "ns_per_genotype_thread": ns,
           "sample": "first %d of the columns (n=%d, %.1f GB of the 2-bit payload): one bed_prodVec + one "
                     "bed_cprodVec pass (+ colstats %.2fs), OpenMP %d threads; a full 1M-column pass is %.0fx this"
                     % (m_s, n, n_byte * m_s / 1e9, t["colstats"], thr, 1e6 / m_s),
           "seconds": t}
    # BASELINE.md §3 measured the reference's kernels at 1.1 (A x) / 0.9 (A' x) ns per genotype on one
    # thread; far off that, the threads did not run in parallel (or the host is memory-starved): say so
    worst = max(ns["prodVec"] / 1.1, ns["cprodVec"] / 0.9)
    if worst > 3.0:
        res["warning"] = ("per-thread rate is %.1fx the single-thread reference measurement of BASELINE.md §3 "
                          "(memory-bound at this thread count or oversubscribed host)" % worst)
    return res


def _bedfile_nocopy(orc, payload, n, m):
    """oracle BedFile over an existing payload array (BedFile.from_payload would copy it into one
    thread's NUMA node again)"""
    ob = orc.BedFile.__new__(orc.BedFile)
    ob.raw = None
    ob.n, ob.m, ob.n_byte = int(n), int(m), (int(n) + 3) // 4
    ob.payload = payload
    return ob


def download_cols(ba, gb, m_s):
    """first m_s columns of the device image as a .bed payload: the same columns are generated again in
    a small image (same seed, same bytes) and downloaded"""
    full_nbyte = (gb.nrow + 3) // 4
    small = ba.bed.synthetic(gb.nrow, m_s, seed=20250905, j_begin=0)
    out = small.download()
    small.close()
    assert out.size == full_nbyte * m_s
    return out


def ingest(ba, L, n, gb_size, cold_hook=None):
    """bsn_bed_open on a real file: header check, parallel pread into two pinned buffers, 2-D DMA into
    the padded image, device-side recode.  The file holds the first columns of the benchmark matrix."""
    import numpy as np
    n_byte = (n + 3) // 4
    m_f = int(gb_size * 1e9 // n_byte)
    d = os.environ.get("TMPDIR") or tempfile.gettempdir()
    try:
        st = os.statvfs(d)
        if st.f_bavail * st.f_frsize < 1.5 * m_f * n_byte:
            return {"skipped": "not enough space in %s" % d}
    except Exception:
        pass
    path = os.path.join(d, "bsn_bench_ingest_%d.bed" % os.getpid())
    try:
        small = ba.bed.synthetic(n, m_f, seed=20250905, j_begin=0)
        payload = small.download()
        small.close()
        with open(path, "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]))
            f.write(memoryview(payload))
        del payload
        import ctypes as C
        from bigsnpr_amd import _lib
        times = []
        for _ in range(2):
            h = C.c_void_p()
            t0 = time.perf_counter()
            _lib.check(L.bsn_bed_open(path.encode(), n, m_f, C.byref(h)))   # returns when the image is in HBM
            times.append(time.perf_counter() - t0)
            L.bsn_bed_close(h)
        size = 3 + n_byte * m_f
        cold = None
        if cold_hook is not None:   # a fresh process opens THIS file and solves on it (this process holds no handle on it now)
            cold = cold_hook(path, n, m_f)
            if isinstance(cold, dict) and "handle_s" in cold:
                cold["open_ms"] = 1e3 * cold.pop("handle_s")
                cold["file"] = "%d x %d .bed, %.1f GB, page cache warm" % (n, m_f, size / 1e9)
        return {"cold": cold, "file_GB": size / 1e9, "seconds": times, "GBps": size / 1e9 / min(times),
                "projected_s_for_100GB": 100.0 / (size / 1e9 / min(times)),
                "note": "bsn_bed_open of a %.1f GB .bed written by this process (page cache warm); paid once per "
                        "handle, never part of `value`" % (size / 1e9)}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def ld_bench(a, ba, L, rank=0, world=1, dist=None, torch=None):
    """config C5: bed_ld_scores + bed_cor on one chromosome (n x m 2-bit image in HBM), windows of
    `window` variants; with N GPUs every rank holds a chromosome of its own (R/clumping.R:83-88 splits
    by chromosome; no collective, weak scaling).  roofline: int8 MFMA work of the pair-statistics
    kernel / its launch time."""
    import numpy as np
    n, m, W = a.n, a.m or 100000, a.window
    gb = ba.bed.synthetic(n, m, seed=5 + rank)
    pos = np.arange(m, dtype=np.float64)
    L.bsn_device_sync()
    if world > 1:
        dist.barrier()
    res, stats = {}, {}
    for name, fn in (("bed_ld_scores", lambda: ba.bed_ld_scores(gb, size=W / 1000.0, infos_pos=pos)),
                     ("bed_cor", lambda: ba.bed_cor(gb, size=W / 1000.0, infos_pos=pos))):
        for _ in range(a.warmup):
            fn()
        L.bsn_device_sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        L.bsn_device_sync()
        res[name] = (time.perf_counter() - t0) / a.steps
        if world > 1:   # slowest rank
            dist.barrier()
            t = torch.tensor([res[name]], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res[name] = float(t.item())
        stats[name] = ba.ld.last_stats()
    st = stats["bed_ld_scores"]
    pairs = st["pairs"]
    # exact integer products of 2 * n ops per variant pair of every 64 x 64 tile pair the band touches
    # (src/corr.cpp:54-75 restated as six GEMMs over the samples; one when no value is missing)
    ops = st["products"] * 2.0 * n * st["tile_pairs"] * 64 * 64
    achieved = ops / (st["stats_ms"] * 1e-3) / 1e12
    on_fp4 = "FP4" in st["kernel"]      # the six products on the FP4 matrix pipe (exact for these plane values): priced against ITS peak
    peak = FP4_PEAK_TOPS if on_fp4 else I8_PEAK_TOPS
    return {"metric": "variant pairs/sec for bed_ld_scores, window %d variants" % W,
            "value": world * pairs / res["bed_ld_scores"], "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * res["bed_ld_scores"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("fp4 (E2M1) planes holding the exact integers 0 / 1 / 2 / 4, fp32 MFMA accumulation (exact: sums < 2^24), fp64 epilogue"
                      if on_fp4 else "i8 planes, exact int32 MFMA accumulation, fp64 epilogue"), "data": "synthetic",
            "config": {"workload": "bed_ld_scores / bed_cor on synthetic %dx%d 2-bit image, window %d variants (config C5)"
                                   % (n, m, W), "n": n, "m": m, "window": W,
                       "parallelism": "one chromosome per GPU, no collective" if world > 1 else "single GPU"},
            "pairs": pairs, "bed_cor_ms": 1e3 * res["bed_cor"],
            "roofline": {"bound": "mfma", "kernel": st["kernel"], "achieved": achieved, "peak": peak,
                         "unit": "TOP/s", "frac": achieved / peak, "frac_of_int8_peak": achieved / I8_PEAK_TOPS, "traffic": None,
                         "ops_all_launches": ops, "ms_all_launches": st["stats_ms"], "launches": st["launches"],
                         "tile_pairs": st["tile_pairs"], "products": st["products"],
                         "note": "achieved = useful int8 ops of the %d launches of one call / their summed HIP-event time"
                                 % st["launches"]}}


def matvec_bench(a, ba, L, rank=0, world=1, dist=None, torch=None):
    """config C2 (BASELINE.json configs[1]): bed_prodVec / bed_cprodVec of an FBM.code256 of calls, 50 000 x
    200 000 by default, as ONE-SHOT calls — host vectors in, host vector out, everything the R function pays
    except R itself.  A step = one prodVec + one cprodVec.  With N GPUs every rank holds its own matrix (weak
    scaling, no collective).  roofline: 2-bit payload of the image / HIP-event time of the streaming kernel inside
    the calls is not separable here, so `achieved` prices the whole call (transfers included) against HBM."""
    import numpy as np
    n, m = (a.n if a.n != 400000 else 50000), a.m or 200000
    gb = ba.bed.synthetic(n, m, seed=9 + rank)
    sc = ba.bed_scaleBinom(gb)
    rng = np.random.default_rng(rank)
    x, y = rng.normal(size=m), rng.normal(size=n)
    step = lambda: (ba.bed_prodVec(gb, x, center=sc["center"], scale=sc["scale"]),
                    ba.bed_cprodVec(gb, y, center=sc["center"], scale=sc["scale"]))
    for _ in range(max(a.warmup, 3)):      # the first calls of an entry allocate its work buffers
        step()
    L.bsn_device_sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    L.bsn_device_sync()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    per_call = wall / a.steps / 2
    nbytes = ((n + 3) // 4) * m
    achieved = nbytes / per_call / 1e9
    out = {"metric": "SNP-cols/sec for one-shot bed_prodVec / bed_cprodVec calls (m / time per call)",
           "value": world * m / per_call, "unit": "SNP-cols/s", "n_gpus": world, "steps": a.steps, "warmup": max(a.warmup, 3),
           "ms_per_step": 1e3 * wall / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "i8 (2-bit codes x 56-bit fixed-point image of the vector, 7 int8 slices, exact int32 MFMA accumulation)",
           "data": "synthetic",
           "config": {"workload": "bed_prodVec + bed_cprodVec, one-shot calls with host vectors, synthetic %dx%d 2-bit image "
                                  "(config C2)" % (n, m), "n": n, "m": m,
                      "parallelism": "one matrix per GPU, no collective" if world > 1 else "single GPU"},
           "ms_per_call": 1e3 * per_call,
           "roofline": {"bound": "hbm", "kernel": "whole call (vector upload, quantise, k_prod / k_cprod, finalize, download)",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "bytes_per_launch": nbytes}}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(ba, gb, n, min(m, a.cpu_sample_cols or m))
        out["cpu_baseline"] = cb
        out["gpu_over_cpu"] = out["value"] / cb["value"]
    return out


if __name__ == "__main__":
    main()
