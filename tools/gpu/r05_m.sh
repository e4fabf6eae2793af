#!/bin/bash
# round 5, trip M: shape sweep of the three-column-block kernels (profiling build, BSN_NB3)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05m; mkdir -p $O; : > $O/summary.txt
cat > /tmp/v.py <<'P'
import os, time, numpy as np, bigsnpr_amd as ba
gb = ba.bed.synthetic(400000, 1000000)
r = ba.bed_randomSVD(gb, k=20, return_uv=False)
rs = [ba.bed_randomSVD(gb, k=20, return_uv=False) for _ in range(4)]
f = lambda key, cnt: sum(r[key] for r in rs) / max(1, sum(r[cnt] for r in rs))
print("BSN_NB3=%s" % os.environ.get("BSN_NB3", "0"), "solve %.1f ms" % (sum(r["gpu_ms"] for r in rs) / 4), "wide cprod %.2f ms" % f("wide_cprod_ms", "n_wide_cprod"), "wide prod %.2f ms" % f("wide_prod_ms", "n_wide_prod"),
      "narrow cprod %.2f prod %.2f stats %.2f" % (f("cprod_ms", "n_cprod"), f("prod_ms", "n_prod"), f("cprod_stats_ms", "n_cprod_stats")), "sigma1 %.6f" % rs[0]["d"][0], flush=True)
P
export BSN_LIB_PATH=$GRAFT_REPO_ROOT/bigsnpr_amd/libbigsnpr_hip_abl.so
for v in 0 1 2 3 8 4 0; do BSN_NB3=$v python /tmp/v.py 2>&1 | tail -1 | tee -a $O/summary.txt; done
