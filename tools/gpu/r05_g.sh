#!/bin/bash
# round 5, trip G: the subset path (ind.col = a sorted random 30 % of the variants): gather lists against the compacted copy
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05g; mkdir -p $O; : > $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_svd.py tests/test_gpu_autosvd.py tests/test_gpu_out_of_core.py tests/test_gpu_smaj.py -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed|error' $O/tests.log | tail -1)" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/tests.log | head -20
for mode in compact gather; do
  if [ $mode = gather ]; then export BSN_NO_COMPACT=1; else unset BSN_NO_COMPACT; fi
  timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 --ind-col-fraction 0.3 --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/subset_$mode.json 2> $O/subset_$mode.err
  python - <<P | tee -a $O/summary.txt
import json
d = json.load(open('$O/subset_$mode.json'))
print('$mode: %.2f ms per solve' % d['ms_per_step'], 'niter', d['niter'], 'passes %.2f' % d['passes_per_solve'], d['ind_col'],
      {k: (round(v['avg_ms'], 2), v['launches'], v['column_blocks'], round(v['GBps'])) for k, v in d['roofline']['other'].items()})
P
done
unset BSN_NO_COMPACT
# cost of the pieces of a compaction: allocation, gather, sample-major build
python - <<'P' 2>&1 | tee -a $O/summary.txt
import time, ctypes as C, numpy as np, bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()
hip = C.CDLL("libamdhip64.so")
for gbs in (1, 10, 30):
    p = C.c_void_p(); t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(gbs << 30)); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
    hip.hipMemset(p, 0, C.c_size_t(gbs << 30)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipFree(p); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
    print("hipMalloc %d GB: %.1f ms, first memset %.1f ms, hipFree %.1f ms (rc %d)" % (gbs, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), rc))
gb = ba.bed.synthetic(400000, 1000000)
rng = np.random.default_rng(1)
prev = None
for frac in (0.30, 0.29, 0.28):
    ic = np.sort(rng.choice(1000000, int(frac * 1000000), replace=False))
    t0 = time.perf_counter(); r = ba.bed_randomSVD(gb, ind_col=ic, k=20); t1 = time.perf_counter()
    r2 = ba.bed_randomSVD(gb, ind_col=ic, k=20); t2 = time.perf_counter()
    print("fraction %.2f: first solve %.1f ms (compaction %.1f ms), second %.1f ms" % (frac, 1e3 * (t1 - t0), r["compact_ms"], 1e3 * (t2 - t1)), r["compacted"], r["tiled"], r["niter"])
P
