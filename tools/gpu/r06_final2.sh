#!/bin/bash
# round 6: final trip (second half of the round: after the LD / outlier-step / counts changes; bench line with auto_svd) — whole GPU suite (file by file), smoke, PMC passes for the traffic record, the driver's bench line, kernel stats, shard lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06final2; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_*.py tests/test_prs_pipeline_golden.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
bash tools/pmc_run.sh r06_block16 --steps 2 --warmup 1 --no-wide --no-accuracy --no-cold > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r06_block16 > $O/pmc_block16.txt 2>&1
python tools/make_pmc_traffic.py gpurun_out/pmc_r06_block16 > $O/pmc_traffic.json 2> $O/pmc_traffic.err
python -c "import json; json.load(open('$O/pmc_traffic.json'))" && cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<P | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default: %.2f ms' % d['ms_per_step'], 'value %.3e' % d['value'], 'roofline', r['bound'], round(r['frac'],3), 'hbm', round(r['hbm']['frac'],3), 'traffic', r['traffic'], r.get('paced_by'),
      {k:(round(v['avg_ms'],2), v['launches'], v['column_blocks']) for k,v in r['other'].items()})
print('accuracy', {k: d['accuracy'].get(k) for k in ('u_leading_half','v_leading_half','worse_of_two_matrices','leading_half_within_tolerance')})
c=d.get('cold',{})
print('cold full', {k:c.get('synthetic_full_size',{}).get(k) for k in ('first_solve_ms','warm_solve_ms','first_minus_warm_ms','solve_ms')})
print('cold bed', {k:c.get('real_bed_file',{}).get(k) for k in ('open_ms','first_solve_ms','warm_solve_ms','file')})
print('alternatives', {k: (round(v['ms'],1), v.get('angles_to_reference', {}).get('u_leading_half')) for k, v in d['fp64_equivalent'].items() if 'ms' in v})
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'ingest', d.get('ingest', {}).get('GBps'))
print('auto_svd', d.get('auto_svd'))
P
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-wide --no-accuracy --no-cold --no-autosvd > /dev/null 2> /tmp/pk.err
cd "$GRAFT_REPO_ROOT"
cp $(find /tmp/pk -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
python tools/trace_gaps.py $(find /tmp/pk -name '*kernel_trace.csv' | head -1) > $O/bench_solve_timeline.txt; tail -3 $O/bench_solve_timeline.txt | cut -c1-250
for N in 2 4 8; do
  timeout 300 python bench.py --steps 8 --warmup 2 --shard-of $N --force-dist --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/shard_$N.json 2> $O/shard_$N.err
done
python tools/shard_projection.py $O/bench_default.json $O/shard_2.json $O/shard_4.json $O/shard_8.json | tee $O/shard_projection.txt
timeout 300 python bench.py --steps 6 --warmup 2 --force-dist --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> /dev/null
timeout 300 python bench.py --workload matvec --steps 20 --warmup 3 > $O/c2_matvec.json 2> /dev/null
for f in bench_rccl_1rank ld_bench c2_matvec; do python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), 'ms', d['roofline'].get('frac'))" | tee -a $O/summary.txt; done
