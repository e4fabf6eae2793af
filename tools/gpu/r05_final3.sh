#!/bin/bash
# round 5: records on the build with the NASKIP kernels — whole GPU suite, smoke, PMC passes (traffic record), bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05final3; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_*.py tests/test_prs_pipeline_golden.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
BSN_TEST_SEED_OFFSET=3000 timeout 900 python -m pytest tests/test_gpu_random_shapes.py -q -x 2>&1 | tail -1 | sed "s/^/seed offset 3000: /" | tee -a $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
bash tools/pmc_run.sh r05_block16 --steps 2 --warmup 1 --no-wide --no-accuracy > /dev/null 2>&1
bash tools/pmc_run.sh r05_block8 --steps 2 --warmup 1 --no-wide --no-accuracy --block 8 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r05_block16 > $O/pmc_block16.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r05_block8 > $O/pmc_block8.txt 2>&1
python tools/make_pmc_traffic.py gpurun_out/pmc_r05_block16 gpurun_out/pmc_r05_block8 > $O/pmc_traffic.json 2> $O/pmc_traffic.err
cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<P | tee -a $O/summary.txt
import json
d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('default: %.2f ms' % d['ms_per_step'], 'value %.3e' % d['value'], 'roofline', r['bound'], round(r['frac'],3), 'hbm', round(r['hbm']['frac'],3), 'mfma', {k:(round(v,3) if isinstance(v,float) else v) for k,v in r['mfma'].items()}, 'traffic', r['traffic'], r.get('paced_by'),
      {k:(round(v['avg_ms'],2), v['launches'], v['column_blocks']) for k,v in r['other'].items()}, d['missing_values'])
print('accuracy', {k: d['accuracy'][k] for k in ('u_leading_half','u_all','v_leading_half','v_all','leading_half_within_tolerance')})
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'ingest', d.get('ingest', {}).get('GBps'))
P
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-wide --no-accuracy > /dev/null 2> /tmp/pk.err
cd "$GRAFT_REPO_ROOT"
cp $(find /tmp/pk -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
python tools/trace_gaps.py $(find /tmp/pk -name '*kernel_trace.csv' | head -1) > $O/bench_solve_timeline.txt; tail -3 $O/bench_solve_timeline.txt | cut -c1-250
B="python bench.py --gpus 1 --steps 8 --warmup 2 --no-accuracy --no-cpu-baseline --no-ingest --no-wide"
for na in 66 6; do
  timeout 600 $B --na16 $na > $O/bench_na$na.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_na$na.json')); print('na16=$na (the library\'s own choice): %.1f ms' % d['ms_per_step'], d['missing_values'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})" | tee -a $O/summary.txt
done
timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> /dev/null
timeout 600 python bench.py --workload matvec --steps 200 --warmup 20 > $O/c2_matvec.json 2> /dev/null
python -c "
import json
d=json.load(open('$O/ld_bench.json')); print('ld: bed_ld_scores %.1f ms, bed_cor %.1f ms, frac %.3f' % (d['ms_per_step'], d['bed_cor_ms'], d['roofline']['frac']), d['roofline']['kernel'][:20])
d=json.load(open('$O/c2_matvec.json')); print('c2: %.4f ms per call' % d['ms_per_call'], 'cpu baseline', d['cpu_baseline']['value'])" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 6 --warmup 2 --force-dist --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
python -c "
import json; d=json.load(open('$O/bench_rccl_1rank.json')); print('rccl 1 rank', round(d['ms_per_step'],2), 'ms', json.dumps(d['exchange'])[:500])" | tee -a $O/summary.txt
