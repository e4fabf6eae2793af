#!/bin/bash
# round 6, trip N: the cross-product LD kernel on the FP4 pipe (k_pair_xy_f4) — parity suite, A/B against the int8 kernel, autoSVD at 1M
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ld.py tests/test_gpu_fbm.py tests/test_gpu_sct.py tests/test_gpu_autosvd.py tests/test_gpu_out_of_core.py tests/test_prs_pipeline_golden.py tests/test_gpu_random_shapes.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
python - > $O/ld_complete_ab.txt 2>&1 <<'PY'
import os, time, numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import ld as ldm
n, m, W = 400000, 100000, 2000
gb = ba.bed.synthetic(n, m, seed=5, na16=0)            # complete data: the cross product alone
pos = np.arange(m, dtype=np.float64)
res = {}
for tag, env in (("fp4", None), ("int8", "1")):
    if env: os.environ["BSN_LD_I8"] = env
    else: os.environ.pop("BSN_LD_I8", None)
    for rep in range(3):
        t0 = time.perf_counter(); ld = ba.bed_ld_scores(gb, size=W / 1000.0, infos_pos=pos); dt = time.perf_counter() - t0
    st = ldm.last_stats()
    print("%-5s bed_ld_scores %.1f ms, pair-statistics kernels %.1f ms (%d launches), %s" % (tag, 1e3 * dt, st["stats_ms"], st["launches"], st["kernel"][:60]))
    res[tag] = ld
print("identical:", np.array_equal(res["fp4"], res["int8"]))
PY
cat $O/ld_complete_ab.txt
timeout 600 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_1m.txt 2>&1
tail -4 $O/autosvd_1m.txt | cut -c1-400
