#!/bin/bash
# round 5: final trip — whole GPU suite (file by file), random shapes with other seeds, PMC passes for the traffic record, bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05final; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_*.py tests/test_prs_pipeline_golden.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
for off in 1000; do BSN_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random_shapes.py -q -x 2>&1 | tail -1 | sed "s/^/seed offset $off: /" | tee -a $O/summary.txt; done
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
      {k:(round(v['avg_ms'],2), v['launches'], v['column_blocks']) for k,v in r['other'].items()})
print('accuracy', {k: d['accuracy'][k] for k in ('u_leading_half','u_all','v_leading_half','v_all','leading_half_within_tolerance')})
print('alternatives', {k: (round(v['ms'],1), v.get('angles_to_reference', {}).get('u_leading_half')) for k, v in d['fp64_equivalent'].items() if 'ms' in v})
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'ingest', d.get('ingest', {}).get('GBps'))
P
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-wide --no-accuracy > /dev/null 2> /tmp/pk.err
cd "$GRAFT_REPO_ROOT"
cp $(find /tmp/pk -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
python tools/trace_gaps.py $(find /tmp/pk -name '*kernel_trace.csv' | head -1) > $O/bench_solve_timeline.txt; tail -3 $O/bench_solve_timeline.txt | cut -c1-250
timeout 300 python bench.py --steps 6 --warmup 2 --force-dist --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
timeout 300 python bench.py --shard-of 8 --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/shard125k_block16.json 2> /dev/null
timeout 300 python bench.py --k 10 --steps 6 --warmup 2 --no-cpu-baseline --no-ingest --no-wide > $O/bench_k10.json 2> /dev/null
for f in bench_rccl_1rank shard125k_block16 bench_k10; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, json.dumps(d['exchange'])[:600], d.get('accuracy', {}).get('u_leading_half'))" | tee -a $O/summary.txt; done
grep -i "exchange\|abort" $O/bench_rccl_1rank.err | head -5 | tee -a $O/summary.txt
