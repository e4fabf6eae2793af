#!/bin/bash
# round 4: final trip — whole GPU suite (file by file), random shapes with other seeds, PMC passes for the traffic record, bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04final; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_*.py tests/test_prs_pipeline_golden.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
for off in 1000 2000; do BSN_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random_shapes.py -q -x 2>&1 | tail -1 | sed "s/^/seed offset $off: /" | tee -a $O/summary.txt; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
bash tools/pmc_run.sh r04_block16 --steps 2 --warmup 1 --no-wide > /dev/null 2>&1
bash tools/pmc_run.sh r04_block8 --steps 2 --warmup 1 --no-wide --block 8 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r04_block16 > $O/pmc_block16.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r04_block8 > $O/pmc_block8.txt 2>&1
python tools/make_pmc_traffic.py gpurun_out/pmc_r04_block16 gpurun_out/pmc_r04_block8 > $O/pmc_traffic.json 2> $O/pmc_traffic.err
cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<P
import json
d=json.load(open('$O/bench_default.json')); print('default:', round(d['ms_per_step'],2),'ms value %.3e' % d['value'], 'frac', round(d['roofline']['frac'],3), 'traffic', d['roofline']['traffic'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, 'smaj build', d['sample_major_copy_build_s'])
P
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-wide > /dev/null 2> /tmp/pk.err
cd "$GRAFT_REPO_ROOT"
cp $(find /tmp/pk -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
python tools/trace_gaps.py $(find /tmp/pk -name '*kernel_trace.csv' | head -1) > $O/bench_solve_timeline.txt; tail -3 $O/bench_solve_timeline.txt | cut -c1-250
timeout 300 python bench.py --steps 6 --warmup 2 --force-dist --no-cpu-baseline --no-ingest --no-wide > $O/bench_rccl_1rank.json 2> /dev/null
timeout 300 python bench.py --shard-of 8 --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-ingest --no-wide > $O/shard125k_block16.json 2> /dev/null
for f in bench_rccl_1rank shard125k_block16; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})"; done
