#!/bin/bash
# round 5: the whole GPU suite + smoke on the last build (after the driver fix)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05final4; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_*.py tests/test_prs_pipeline_golden.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ingest > $O/bench_default_short.json 2> /dev/null
python -c "
import json; d=json.load(open('$O/bench_default_short.json')); print('default (10 steps): %.1f ms' % d['ms_per_step'], 'roofline', d['roofline']['bound'], round(d['roofline']['frac'],3), 'traffic', d['roofline']['traffic'], 'u lead', d['accuracy']['u_leading_half'])" | tee -a $O/summary.txt
