#!/usr/bin/env python3
"""The per-rank table behind DESIGN.md section 5: from the `bench.py --shard-of N --force-dist` records of one
GPU (profiles/r06_shard_{2,4,8}.json) and the full-size record (profiles/r06_bench_default.json) — streaming kernel averages
on the shard, what a rank of the N-GPU run streams per solve (the SHARDED run does the arithmetic of the unsharded solve:
its block steps and launch counts), the panel algebra divided by N, the exchange by class — and the projected time and
strong-scaling efficiency with every input named.  Prints the table.

    python tools/shard_projection.py profiles/r06_bench_default.json profiles/r06_shard_2.json profiles/r06_shard_4.json profiles/r06_shard_8.json
"""
import json, sys


def load(p):
    return json.loads(open(p).read().strip().splitlines()[-1])


full = load(sys.argv[1])
shards = {int(d["config"]["m_total"] // d["config"]["m_per_gpu"]): d for d in map(load, sys.argv[2:])}
ko = full["roofline"]["other"]
launches = {k: v["launches"] / full["steps"] for k, v in ko.items()}      # per solve at full size: what every rank of a sharded run issues
warm_ms = full["warm_start"]["ms"]
stream1 = sum(v["avg_ms"] * launches[k] for k, v in ko.items()) + warm_ms
T1 = full["ms_per_step"]
other1 = T1 - stream1                 # panel algebra, quantiser / finalize kernels, u / v to the host, host Rayleigh-Ritz
n, k = full["config"]["n"], 20
print("# inputs: full-size solve %.1f ms = %.1f streaming (%s; warm start %.1f) + %.1f everything else" % (
    T1, stream1, ", ".join("%.0f x %s %.2f" % (launches[x], x, ko[x]["avg_ms"]) for x in ko), warm_ms, other1))
print("# exchange model (NOT measured: no N > 1 box): per block step one reduce-scatter of the n x 16 fp64 panel (%.0f MB in all,"
      % (n * 16 * 8 / 1e6))
print("#   (N-1)/N of it crosses a rank's links) overlapped with the product pass except its last segment, one all-gather of the rounded")
print("#   block as int32 / int16 (%.1f / %.1f MB) exposed, two small all-reduces; link rate assumed 60 GB/s per direction and rank"
      % (n * 16 * 4 / 1e6, n * 16 * 2 / 1e6))
print("#   (xGMI: 7 links x ~ 50 GB/s usable, ring over one link pair at a time), 20 us per collective; the 1-rank RCCL timers give the launch cost")
print("%-3s %-44s %-10s %-12s %-10s %-10s %-10s %-9s %-9s" % ("N", "shard kernels ms (wide_prod wide_cprod prod cprod stats)", "TB/s wide", "streaming ms", "other ms", "exchange", "T(N) ms", "speed-up", "efficiency"))
for N in sorted(shards):
    d = shards[N]
    o = d["roofline"]["other"]
    stream = sum(o[x]["avg_ms"] * launches[x] for x in ko) + warm_ms / N
    # n-side and m-side panel algebra, finalize kernels and the v download scale with 1 / N; the host Rayleigh-Ritz step, the
    # all-gathered u (n x k on every rank) and launch gaps do not: 3 ms of the full-size "everything else" are taken as fixed
    other = 3.0 + (other1 - 3.0) / N
    steps = full["niter"]
    rs_bytes = n * 16 * 8 * (N - 1) / N
    ag_bytes = n * 16 * 4 * (N - 1) / N
    link = 60e9
    ex_meas = d["exchange"].get("ms_per_solve") or {}
    launch_cost = sum(ex_meas.values()) * steps / max(1, d["niter"])          # the 1-rank timers, scaled to the sharded solve's steps
    exposed = steps * (0.25 * rs_bytes / link + ag_bytes / link + 4 * 20e-6) * 1e3 + launch_cost
    T = stream + other + exposed
    bytes_wide = d["roofline"]["bytes_per_launch"]
    print("%-3d %-44s %-10.2f %-12.1f %-10.1f %-10.1f %-10.1f %-9.2f %-9.2f" % (
        N, " ".join("%.2f" % o[x]["avg_ms"] for x in ("wide_prod", "wide_cprod", "prod", "cprod", "cprod_stats")),
        bytes_wide / o["wide_prod"]["avg_ms"] / 1e9, stream, other, exposed, T, T1 / T, T1 / T / N))
print("# (shard solved ALONE on one GPU, for reference: %s ms per solve)" % ", ".join("N=%d: %.1f" % (N, shards[N]["ms_per_step"]) for N in sorted(shards)))
