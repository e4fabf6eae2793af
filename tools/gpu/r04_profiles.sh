#!/bin/bash
# round 4: the measurement trip behind profiles/r04_* (final build of the round)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04prof; mkdir -p $O
# 1. the driver's command line
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<P
import json
d=json.load(open('$O/bench_default.json')); print('default:', round(d['ms_per_step'],2),'ms value %.3e' % d['value'], 'passes', d['passes_per_solve'], 'frac', round(d['roofline']['frac'],3), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, 'cpu', d.get('cpu_baseline',{}).get('value'))
P
# 2. kernel trace + stats of the same command, cut into solves
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-wide > /dev/null 2> /tmp/pk.err
cd "$GRAFT_REPO_ROOT"
cp $(find /tmp/pk -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
python tools/trace_gaps.py $(find /tmp/pk -name '*kernel_trace.csv' | head -1) > $O/bench_solve_timeline.txt; tail -3 $O/bench_solve_timeline.txt | cut -c1-250
# 3. PMC passes (each on its own, kernel-trace only): default solve and the 8-vector solve
bash tools/pmc_run.sh r04_block16 --steps 2 --warmup 1 --no-wide > /dev/null 2>&1
bash tools/pmc_run.sh r04_block8 --steps 2 --warmup 1 --no-wide --block 8 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r04_block16 > $O/pmc_block16.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r04_block8 > $O/pmc_block8.txt 2>&1
python tools/make_pmc_traffic.py gpurun_out/pmc_r04_block16 gpurun_out/pmc_r04_block8 > $O/pmc_traffic.json 2> $O/pmc_traffic.err
rm -rf gpurun_out/pmc_r04_block16/*/ gpurun_out/pmc_r04_block8/*/ 2>/dev/null
head -c 1500 $O/pmc_traffic.json
# 4. other configurations
timeout 300 python bench.py --steps 6 --warmup 2 --block 8 --no-cpu-baseline --no-ingest --no-wide > $O/bench_block8.json 2> /dev/null
timeout 300 python bench.py --steps 6 --warmup 2 --force-dist --no-cpu-baseline --no-ingest --no-wide > $O/bench_rccl_1rank.json 2> /dev/null
BSN_NO_SMAJ=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ingest --no-wide > $O/bench_nosmaj.json 2> /dev/null
timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> /dev/null
timeout 300 python bench.py --workload matvec --steps 200 --warmup 20 > $O/c2_matvec.json 2> /dev/null
for f in bench_block8 bench_rccl_1rank bench_nosmaj ld_bench c2_matvec; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],3), 'ms value %.3e' % d['value'])"; done
timeout 600 python tools/probe_ooc.py --gb 8 --budget-gb 2 2>&1 | grep "^{" > $O/ooc.json; python -c "
import json; d=json.load(open('$O/ooc.json')); print({k:(round(v['out_of_core']['GBps'],1), v['identical']) for k,v in d.items() if isinstance(v,dict)})"
