#!/bin/bash
# round 5, trip E: the default solve with the grid of a new block chosen from the step's own residuals; bench line with
# the accuracy record
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05e; mkdir -p $O; : > $O/summary.txt
cat > /tmp/var.py <<'P'
import json, os, sys, time, numpy as np, bigsnpr_amd as ba
n, m, k = 400000, 1000000, int(os.environ.get("K", "20"))
gb = ba.bed.synthetic(n, m)
ref = ba.bed_randomSVD(gb, k=k, tol=1e-10, slices=7, block=4)
def angles(a, b):
    s = np.sign((a * b).sum(0)); return np.linalg.norm(a * s - b, axis=0)
h = (k + 1) // 2
for name, kw in json.loads(sys.argv[1]).items():
    r = ba.bed_randomSVD(gb, k=k, **kw); ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = ba.bed_randomSVD(gb, k=k, **kw); ts.append(time.perf_counter() - t0)
    au, av = angles(r["u"], ref["u"]), angles(r["v"], ref["v"])
    print("k", k, name, "%.1f ms" % (1e3 * min(ts)), "block", r["block"], "niter", r["niter"], "wide", r["wide_steps"], r["n_wide_cprod"], r["n_wide_prod"], "launches", r["nops"],
          "resid %.1e %.1e" % (r["lead_rel_resid"], r["max_rel_resid"]), "u %.1e %.1e v %.1e %.1e" % (au[:h].max(), au.max(), av[:h].max(), av.max()),
          "stats pass %.2f" % (r["cprod_stats_ms"] / max(1, r["n_cprod_stats"])), "tiled", r["tiled"], flush=True)
ba.bed_randomSVD(gb, k=k, verbose=True)
P
python /tmp/var.py '{"default": {}, "no schedule": {"vec_floor": -1}}' 2> $O/verbose_k20.txt | tee -a $O/summary.txt
grep "bit\|rel resid" $O/verbose_k20.txt | tee -a $O/summary.txt
BSN_NO_SPECULATION=1 python /tmp/var.py '{"default, no speculation at all": {}}' 2>/dev/null | tee -a $O/summary.txt
K=10 python /tmp/var.py '{"default": {}, "no schedule": {"vec_floor": -1}}' 2> $O/verbose_k10.txt | tee -a $O/summary.txt
grep "bit\|rel resid" $O/verbose_k10.txt | tee -a $O/summary.txt
for f in tests/test_gpu_tiled.py tests/test_gpu_smaj.py tests/test_gpu_svd.py; do
  timeout 900 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -30
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ingest > $O/bench_default.json 2> $O/bench_default.err
python - <<'P' | tee -a $O/summary.txt
import json
d = json.load(open('gpurun_out/r05e/bench_default.json'))
print('bench default: %.2f ms' % d['ms_per_step'], 'value %.3e' % d['value'], 'roofline', d['roofline']['bound'], round(d['roofline']['frac'], 3), d['roofline']['kernel'][:12],
      {k: (round(v['avg_ms'], 2), v['launches'], v['column_blocks']) for k, v in d['roofline']['other'].items()})
print('accuracy', json.dumps(d.get('accuracy')))
print('alternatives', json.dumps(d.get('fp64_equivalent')))
P
