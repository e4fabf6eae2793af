#!/usr/bin/env python3
"""bench.py — bed_randomSVD (k = 20) on a synthetic 2-bit genotype matrix resident in HBM.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete bed_randomSVD solve (scaling-statistics pass + block-Lanczos
passes + panel algebra) on the matrix that is already in HBM.  Metric (BASELINE.json /
BASELINE.md §2): SNP-columns/s = m_total * passes / wall, passes = streaming passes over the
image (A~ / A~' panel applications) + 1 (the colstats pass).  N > 1: the m_total columns
are sharded over ranks (strong scaling, the matrix of BASELINE configs[3]); the n x 8 panel
is all-reduced over RCCL once per step.

Extra JSON objects: "roofline" (dominant streaming kernel, algorithmic bytes per launch /
HIP-event duration measured inside the timed solves) and "cpu_baseline" (the OpenMP CPU
restatement of the reference kernels, oracle/bsn_oracle.c, timed on a bounded sample of
the same matrix on this host; rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=400000)
    ap.add_argument("--m", type=int, default=1000000, help="total SNP columns over all ranks")
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--block", type=int, default=0, help="vectors per pass (0 = library default: 8 at tol 1e-4)")
    ap.add_argument("--tol", type=float, default=1e-4)
    ap.add_argument("--slices", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="exercise the RCCL all-reduce hook even with one rank (self-test)")
    ap.add_argument("--cpu-sample-cols", type=int, default=0)
    ap.add_argument("--verbose", type=int, default=0, help="1: residual trajectory of every solve on stderr")
    return ap.parse_args()


class _DevPtr:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    dist = None
    torch = None
    use_dist = world > 1 or a.force_dist
    # RCCL prints a version banner through C stdio on stdout; keep the real stdout for the one
    # JSON line and send everything else (fd 1) to stderr.
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if use_dist:
        # torch first: the library then binds to the same HIP runtime as torch / RCCL
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29511"
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    import numpy as np
    import bigsnpr_amd as ba
    from bigsnpr_amd import _lib
    L = _lib.load()
    _lib.check(L.bsn_set_device(local_rank))
    ba.selftest()

    n, m_total = a.n, a.m
    j0 = (m_total * rank) // world
    j1 = (m_total * (rank + 1)) // world
    m_local = j1 - j0
    t0 = time.time()
    gb = ba.bed.synthetic(n, m_local, seed=20250905, j_begin=j0)
    L.bsn_device_sync()
    gen_s = time.time() - t0

    allreduce = None
    ar_stats = {"calls": 0, "seconds": 0.0, "bytes": 0}
    if use_dist:
        views = {}

        def allreduce(ptr, count):
            t0 = time.perf_counter()
            t = views.get((ptr, count))
            if t is None:  # alias of the library's device buffer, created once per (ptr, count)
                t = views[(ptr, count)] = torch.as_tensor(_DevPtr(ptr, count), device="cuda")
            dist.all_reduce(t)
            torch.cuda.current_stream().synchronize()
            ar_stats["calls"] += 1
            ar_stats["bytes"] += 8 * count
            ar_stats["seconds"] += time.perf_counter() - t0

    def sync():
        L.bsn_device_sync()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        return ba.bed_randomSVD(gb, k=a.k, tol=a.tol, block=a.block, slices=a.slices, allreduce=allreduce,
                                m_total=m_total, return_uv=False, verbose=a.verbose)

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    infos = [step() for _ in range(a.steps)]
    sync()
    wall = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    passes = sum(r["nops"] + 1 for r in infos)               # + colstats pass of fun.scaling
    value = m_total * passes / wall                           # whole job, all ranks
    bytes_per_launch = ((n + 3) // 4) * m_local               # algorithmic: 2-bit payload of the shard
    kern = {}
    for key, name in (("prod", "k_prod (A~ panel, contraction over variants)"),
                      ("cprod", "k_cprod (A~' panel, contraction over samples)")):
        ms = sum(r[key + "_ms"] for r in infos)
        cnt = sum(r["n_" + key] for r in infos)
        kern[key] = dict(name=name, total_ms=ms, launches=cnt, avg_ms=ms / max(cnt, 1))
    dom_key = max(kern, key=lambda k_: kern[k_]["total_ms"])
    dom = kern[dom_key]
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside
    # the process, so this is the committed rocprofv3 measurement of the same workload
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        # the counters were collected on the one-column-block kernels (block x slices <= 16)
        if pm["workload"] == {"n": n, "m_per_gpu": m_local} and infos[-1]["block"] * infos[-1]["slices"] <= 16:
            traffic = pm["kernels"][dom_key]["hbm_read_bytes"]
    except Exception:
        traffic = None
    achieved = bytes_per_launch / (dom["avg_ms"] * 1e-3) / 1e9
    out = {
        "metric": "SNP-cols/sec for bed_randomSVD k=%d (m*passes/wall)" % a.k,
        "value": value, "unit": "SNP-cols/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "i8 MFMA products / f64 panels", "data": "synthetic",
        "config": {"workload": "bed_randomSVD k=%d on synthetic %dx%d 2-bit .bed image resident in HBM"
                               % (a.k, n, m_total),
                   "n": n, "m_total": m_total, "m_per_gpu": m_local, "block": infos[-1]["block"],
                   "slices": infos[-1]["slices"], "tol": a.tol,
                   "parallelism": "columns sharded x%d, n x %d panel all-reduce" % (world, infos[-1]["block"])},
        "passes_per_solve": passes / a.steps,
        "niter": infos[-1]["niter"], "converged": infos[-1]["converged"],
        "end_to_end_cols_per_s": m_total * a.steps / wall,
        "hbm_GBps_whole_solve": passes * ((n + 3) // 4) * m_total / wall / 1e9,
        "sigma": [float(x) for x in infos[-1]["d"][:5]],
        "generate_s": gen_s,
        "allreduce": {"calls": ar_stats["calls"], "ms_per_call": 1e3 * ar_stats["seconds"] / max(ar_stats["calls"], 1),
                      "bytes_per_solve": ar_stats["bytes"] / max(a.steps + a.warmup, 1),
                      "note": "per block step: one n x block panel + one small Gram block"} if use_dist else None,
        "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE x2, bytes per launch)" if traffic else None,
                     "bytes_per_launch": bytes_per_launch, "avg_launch_ms": dom["avg_ms"],
                     "launches": dom["launches"],
                     "other": {k: {"avg_ms": v["avg_ms"], "launches": v["launches"],
                                   "GBps": bytes_per_launch / (v["avg_ms"] * 1e-3) / 1e9}
                               for k, v in kern.items()}},
    }

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ba, gb, n, a.cpu_sample_cols)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stderr.flush()
    if rank == 0:
        real_stdout.write(json.dumps(out) + "\n")   # the ONE stdout line
        real_stdout.flush()


def cpu_baseline(ba, gb, n, sample_cols):
    """The reference's OpenMP kernels restated in C (oracle/bsn_oracle.c: orc_pMatVec4,
    orc_cpMatVec4, orc_bed_colstats) on the first `m_s` columns of the same matrix, all host
    cores.  Reported as SNP-cols/s of one pass averaged over the three pass kinds a solve
    is made of (A x, A' x, colstats)."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    # bounded sample: ~10-30 s of CPU work.  ~0.3 ns per genotype per core-pass on 8 cores.
    m_s = sample_cols or int(max(256, min(gb.ncol, 2.5e9 * cores / 8 / n)))
    payload = download_cols(ba, gb, m_s)
    ob = orc.BedFile.from_payload(payload, n, m_s)
    st = orc.bed_colstats(ob, ncores=cores)
    af = st["sumX"] / (2.0 * st["nb_nona_col"])
    sc = dict(center=2 * af, scale=np.sqrt(2 * af * (1 - af)))
    scale = np.where(sc["scale"] > 0, sc["scale"], 1.0)
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=m_s), rng.normal(size=n)
    t = {}
    t0 = time.perf_counter(); orc.bed_colstats(ob, ncores=cores); t["colstats"] = time.perf_counter() - t0
    t0 = time.perf_counter(); orc.bed_prodVec(ob, x, None, None, sc["center"], scale, cores); t["prodVec"] = time.perf_counter() - t0
    t0 = time.perf_counter(); orc.bed_cprodVec(ob, y, None, None, sc["center"], scale, cores); t["cprodVec"] = time.perf_counter() - t0
    per_pass = (t["prodVec"] + t["cprodVec"]) / 2
    return {"value": m_s / per_pass, "unit": "SNP-cols/s", "cores": cores, "kind": "port",
            "sample": "first %d of the columns (n=%d): one bed_prodVec + one bed_cprodVec pass "
                      "(+ colstats %.2fs), OpenMP %d threads" % (m_s, n, t["colstats"], cores),
            "seconds": t}


def download_cols(ba, gb, m_s):
    """first m_s columns of the device image as a .bed payload"""
    import numpy as np
    full_nbyte = (gb.nrow + 3) // 4
    # read through the public accessor in column chunks (bsn_bed_read would expand to int32;
    # re-generate the same columns in a small image instead: same seed, same bytes)
    small = ba.bed.synthetic(gb.nrow, m_s, seed=20250905, j_begin=0)
    out = small.download()
    small.close()
    assert out.size == full_nbyte * m_s
    return out


if __name__ == "__main__":
    main()
