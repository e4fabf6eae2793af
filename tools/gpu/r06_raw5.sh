#!/bin/bash
# round 6, second session: the M plane taken from the code plane (c & (c << 1): 96 VALU per 48 MFMA in the hot loop, 110 before) — LD
# tests, the C5 line twice (the look-up kernel once as the box's reference), clumping at C5 size, then the suite in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06raw5; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_ld.py tests/test_gpu_complete_data.py tests/test_gpu_out_of_core.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
for tag in raw lut raw; do
  if [ $tag = lut ]; then export BSN_LD_LUT=1; else unset BSN_LD_LUT; fi
  timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_$tag.json 2> /dev/null
  python -c "
import json; d=json.loads(open('$O/ld_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']
print('C5 $tag: %.1f ms per bed_ld_scores' % d['ms_per_step'], 'kernels %.1f ms' % r['ms_all_launches'], 'frac', round(r['frac'],3), {k:round(v,1) for k,v in d.items() if 'cor' in k})" | tee -a $O/summary.txt
done
unset BSN_LD_LUT
cp $O/ld_raw.json $O/ld_bench.json
timeout 600 python tools/probe_clump.py 100000 2>&1 | tee -a $O/summary.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/suite_one_process.log 2>&1
echo "pytest tests/ -x -q -m gpu (one process): rc=$? $(grep -E 'passed|failed' $O/suite_one_process.log | tail -1) wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/suite_one_process.log | head -20
