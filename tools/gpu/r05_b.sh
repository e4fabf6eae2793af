#!/bin/bash
# round 5, trip B: accuracy frontier on the oracle's shapes, warm-start / start-grid variants at C3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd.py tests/test_gpu_smaj.py -m gpu -q -x -s > $O/test_svd.log 2>&1
echo "svd tests rc=$? $(grep -E 'passed|failed|error' $O/test_svd.log | tail -1)" | tee -a $O/summary.txt
grep -A8 "u/v frontier" $O/test_svd.log | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/test_svd.log | head
cat > /tmp/var.py <<'P'
import json, os, sys, time, numpy as np, bigsnpr_amd as ba
n, m, k = 400000, 1000000, 20
gb = ba.bed.synthetic(n, m)
ref = ba.bed_randomSVD(gb, k=k, tol=1e-10, slices=7, block=4)
def angles(a, b):
    s = np.sign((a * b).sum(0)); return np.linalg.norm(a * s - b, axis=0)
for name, kw in json.loads(sys.argv[1]).items():
    r = ba.bed_randomSVD(gb, k=k, **kw); ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = ba.bed_randomSVD(gb, k=k, **kw); ts.append(time.perf_counter() - t0)
    au, av = angles(r["u"], ref["u"]), angles(r["v"], ref["v"])
    print(os.environ.get("BSN_START_SLICES", "-"), name, "%.1f ms" % (1e3 * min(ts)), "niter", r["niter"], "wide", r["wide_steps"], "passes", r["nops"],
          "resid %.1e %.1e" % (r["lead_rel_resid"], r["max_rel_resid"]), "u %.1e %.1e v %.1e %.1e" % (au[:10].max(), au.max(), av[:10].max(), av.max()),
          "stats pass %.2f" % (r["cprod_stats_ms"] / max(1, r["n_cprod_stats"])), flush=True)
P
python /tmp/var.py '{"default": {}, "warm2": {"warm_start": 2}, "warm3": {"warm_start": 3}, "warm1_den8": {"warm_denominator": 8}, "warm2_den8": {"warm_start": 2, "warm_denominator": 8}, "floor 3e-7": {"vec_floor": 3e-7}, "floor 1e-6": {"vec_floor": 1e-6}}' 2>&1 | tee -a $O/summary.txt
BSN_START_SLICES=1 python /tmp/var.py '{"default": {}, "warm2": {"warm_start": 2}}' 2>&1 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -s -k c3 > $O/test_fullsize.log 2>&1
echo "fullsize c3 rc=$? $(grep -E 'passed|failed|error' $O/test_fullsize.log | tail -1)" | tee -a $O/summary.txt
grep "C3 \|^E " $O/test_fullsize.log | tee -a $O/summary.txt
