#!/bin/bash
# round 6, second session: s_setprio level around the MFMA groups of k_pair_stats_f4<., RAW> (0 / 1 / 2 / 3, profiling build), the same
# on the four-product form (bed clumping at 400K x 100K) and on k_quad_xy_f4 (complete data); LD tests on the product build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06raw4; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_ld.py tests/test_gpu_complete_data.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
export BSN_LIB_PATH=$GRAFT_REPO_ROOT/bigsnpr_amd/libbigsnpr_hip_abl.so
for rep in 1 2; do
  for pr in 0 1 2 3; do
    export BSN_LD_RAW_PRIO=$pr
    timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_prio${pr}_$rep.json 2> /dev/null
    python -c "
import json; d=json.loads(open('$O/ld_prio${pr}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('C5 prio $pr rep $rep: %.1f ms per bed_ld_scores' % d['ms_per_step'], 'kernels %.1f ms' % r['ms_all_launches'], 'frac', round(r['frac'],3), {k:round(v,1) for k,v in d.items() if 'cor' in k})" | tee -a $O/summary.txt
  done
done
for pr in 0 2 0 2; do
  export BSN_LD_RAW_PRIO=$pr
  timeout 600 python tools/probe_clump.py 100000 2>&1 | sed "s/^/clumping, prio $pr: /" | tee -a $O/summary.txt
done
unset BSN_LD_RAW_PRIO
for q in off on off on; do
  if [ $q = on ]; then export BSN_LD_QUAD_PRIO=1; else unset BSN_LD_QUAD_PRIO; fi
  timeout 300 python tools/probe_ld_complete.py 2>&1 | grep "na16=0" | tail -1 | sed "s/^/quad prio $q: /" | cut -c1-200 | tee -a $O/summary.txt
done
