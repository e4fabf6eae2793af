#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05k; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_autosvd.py tests/test_prs_pipeline_golden.py; do
  timeout 900 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -30
python - <<'P' 2>&1 | tee -a $O/summary.txt
import time, numpy as np
from bigsnpr_amd import autosvd as A
rng = np.random.default_rng(0)
for m, k in ((250000, 10), (1000000, 10), (250000, 20)):
    U = rng.normal(size=(m, k)) * rng.uniform(0.5, 2, size=k); U[:500] += 5
    A.dist_ogk(U[:1000], device=True)
    t0 = time.perf_counter(); d1 = A.dist_ogk(U, device=True); t1 = time.perf_counter()
    d0 = A.dist_ogk(U); t2 = time.perf_counter()
    r = A.rollmean(d1, 50); t3 = time.perf_counter(); thr = A.tukey_mc_up(r); t4 = time.perf_counter()
    print("m %d k %d: dist_ogk device %.3f s, host %.3f s (max rel diff %.1e); rollmean %.3f s, tukey_mc_up %.3f s" % (m, k, t1 - t0, t2 - t1, float(np.max(np.abs(d1 / d0 - 1))), t3 - t2, t4 - t3))
P
timeout 900 python tools/probe_autosvd.py 2>&1 | tail -12 | tee -a $O/summary.txt
