#!/bin/bash
# round 6, second session, the last build: the GPU suite in ONE pytest process as the driver runs it, smoke(), the three bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06final8; mkdir -p $O; : > $O/summary.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/suite_one_process.log 2>&1
echo "pytest tests/ -x -q -m gpu (one process): rc=$? $(grep -E 'passed|failed' $O/suite_one_process.log | tail -1) wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/suite_one_process.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> /dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
for f in ld_bench bench_default; do python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), 'ms', d['roofline'].get('frac'), d.get('auto_svd',{}).get('second_call_s'))" | tee -a $O/summary.txt; done
timeout 900 python tools/probe_autosvd.py --m 1000000 --bed > $O/autosvd_bed.txt 2>&1
grep -i "third call\|inside" $O/autosvd_bed.txt | tail -2 | cut -c1-250 | tee -a $O/summary.txt
