#!/bin/bash
# round 6, second session: the GPU suite in ONE pytest process, exactly as the driver runs it (cross-file effects: memory held
# by earlier tests, environment switches); the out-of-core fuzz (tools/fuzz_out_of_core.py) on 40 further draws; the driver's
# bench line with the five-solve cold record
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06final4; mkdir -p $O; : > $O/summary.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/suite_one_process.log 2>&1
echo "pytest tests/ -x -q -m gpu (one process): rc=$? $(tail -1 $O/suite_one_process.log) wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/suite_one_process.log | head -20
timeout 1500 python tools/fuzz_out_of_core.py 100 40 > $O/fuzz_out_of_core.txt 2>&1
echo "fuzz_out_of_core 100..139: rc=$? $(tail -1 $O/fuzz_out_of_core.txt)" | tee -a $O/summary.txt
grep -B1 -A3 MISMATCH $O/fuzz_out_of_core.txt | head -60
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<P | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default: %.2f ms' % d['ms_per_step'], 'value %.3e' % d['value'], 'roofline', r['bound'], round(r['frac'],3), 'hbm', round(r['hbm']['frac'],3))
c=d.get('cold',{}).get('synthetic_full_size',{})
print('cold full', [round(x,1) for x in c.get('solve_ms',[])], [s['image_layout_at_exit'] for s in c.get('solves',[])])
c=d.get('cold',{}).get('real_bed_file',{})
print('cold bed', [round(x,1) for x in c.get('solve_ms',[])], c.get('open_ms'))
P
