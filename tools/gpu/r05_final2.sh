#!/bin/bash
# round 5: last trip on the final build — whole GPU suite, smoke, the default bench line, LD and C2 lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05final2; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_*.py tests/test_prs_pipeline_golden.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
BSN_TEST_SEED_OFFSET=2000 timeout 900 python -m pytest tests/test_gpu_random_shapes.py -q -x 2>&1 | tail -1 | sed "s/^/seed offset 2000: /" | tee -a $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<P | tee -a $O/summary.txt
import json
d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('default: %.2f ms' % d['ms_per_step'], 'value %.3e' % d['value'], 'roofline', r['bound'], round(r['frac'],3), 'hbm', round(r['hbm']['frac'],3), 'traffic', r['traffic'], r.get('paced_by'),
      {k:(round(v['avg_ms'],2), v['launches'], v['column_blocks']) for k,v in r['other'].items()})
print('accuracy', {k: d['accuracy'][k] for k in ('u_leading_half','u_all','v_leading_half','v_all','leading_half_within_tolerance')})
P
timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> /dev/null
timeout 600 python bench.py --workload matvec --steps 200 --warmup 20 > $O/c2_matvec.json 2> /dev/null
python -c "
import json
d=json.load(open('$O/ld_bench.json')); print('ld: bed_ld_scores %.1f ms, bed_cor %.1f ms, frac %.3f' % (d['ms_per_step'], d['bed_cor_ms'], d['roofline']['frac']), d['roofline']['kernel'][:20])
d=json.load(open('$O/c2_matvec.json')); print('c2: %.4f ms per call' % d['ms_per_call'], 'cpu baseline', d['cpu_baseline']['value'])" | tee -a $O/summary.txt
