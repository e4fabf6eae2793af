#!/bin/bash
# VERDICT r4 #5b: the missing-value plane only for K-steps with a missing code (BSN_NA_SKIP), parity + solve A/B + power rows
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05s; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_na_skip.py tests/test_gpu_smaj.py; do
  timeout 900 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -30
B="python bench.py --gpus 1 --steps 8 --warmup 2 --no-accuracy --no-cpu-baseline --no-ingest --no-wide"
run() {  # tag na16 env...
  tag=$1; na=$2; shift 2
  env "$@" timeout 600 $B --na16 $na > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - $O/bench_$tag.json $tag <<'P' | tee -a $O/summary.txt
import json, sys
r = None
for line in open(sys.argv[1]):
    try: r = json.loads(line)
    except Exception: pass
if r is None: print(sys.argv[2], "no line"); sys.exit()
k = r["roofline"]["other"]
print(sys.argv[2], "ms_per_step %.1f" % r["ms_per_step"], "missing", json.dumps(r.get("missing_values")), "sigma0 %.12g" % r["sigma"][0],
      " ".join("%s=%.2fms x%d" % (n, v["avg_ms"], v["launches"]) for n, v in k.items()))
P
}
run na655_plain 655 BSN_NA_SKIP=0
run na655_skip 655 BSN_NA_SKIP=1
run na6_plain 6 BSN_NA_SKIP=0
run na6_auto 6 X=1
run na66_plain 66 BSN_NA_SKIP=0
run na66_skip 66 BSN_NA_SKIP=1
run na0 0 X=1
# power rows: the 16-vector three-block kernels back to back, 4 s each
for cfg in "655 0" "655 1" "6 0" "6 1"; do
  set -- $cfg
  BSN_NA_SKIP=$2 timeout 300 python tools/probe_power.py --n 400000 --m 500000 --seconds 4 --slices 3 --only16 --na16 $1 --tag "BSN_NA_SKIP=$2" 2>&1 | grep '^{' | tee -a $O/power.txt
done
