#!/bin/bash
# round 5, trip C: warm-start / start-grid / floor variants of the default solve at C3; NB=3 against split launches
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05c; mkdir -p $O
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
python /tmp/var.py '{"default": {}, "warm2": {"warm_start": 2}, "warm3": {"warm_start": 3}, "warm1_den8": {"warm_denominator": 8}, "warm2_den8": {"warm_start": 2, "warm_denominator": 8}, "floor 3e-7": {"vec_floor": 3e-7}, "floor 1e-6": {"vec_floor": 1e-6}, "floor 2e-6": {"vec_floor": 2e-6}}' 2>&1 | tee -a $O/summary.txt
BSN_START_SLICES=1 python /tmp/var.py '{"default": {}, "warm2": {"warm_start": 2}}' 2>&1 | tee -a $O/summary.txt
cat > /tmp/ab.py <<'P'
import sys, numpy as np, bigsnpr_amd as ba
gb = ba.bed.synthetic(100000, 300000)
r = ba.bed_randomSVD(gb, k=20, slices=3, block=16)
np.save(sys.argv[1], np.concatenate([r["d"], r["u"].ravel(), r["v"].ravel()]))
print(r["niter"], r["n_wide_cprod"], r["n_wide_prod"], r["n_cprod"], r["n_prod"], r["wide_cprod_ms"], r["wide_prod_ms"])
P
python /tmp/ab.py /tmp/a.npy; BSN_NO_NB3=1 python /tmp/ab.py /tmp/b.npy
python -c "
import numpy as np; a=np.load('/tmp/a.npy'); b=np.load('/tmp/b.npy'); print('NB3 vs split launches: identical =', bool(np.array_equal(a,b)), 'max diff', float(np.abs(a-b).max()))" | tee -a $O/summary.txt
