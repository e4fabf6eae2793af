#!/bin/bash
# round 6, second session: s_setprio in the look-up FP4 kernel (BSN_LD_LUT=1: what 1.86M < n <= 4.19M samples take) and in the int8 kernel
# (BSN_LD_I8=1: n > 4.19M) against BSN_LD_NOPRIO=1, profiling build, C5, same box alternating; LD tests with each kernel family
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06raw8; mkdir -p $O; : > $O/summary.txt
for fam in "" BSN_LD_LUT BSN_LD_I8; do
  [ -n "$fam" ] && export $fam=1
  timeout 1500 python -m pytest tests/test_gpu_ld.py -m gpu -q -x -k "not raw_plane" > $O/test_gpu_ld_$fam.log 2>&1
  echo "tests/test_gpu_ld.py [$fam] rc=$? $(grep -E 'passed|failed|error' $O/test_gpu_ld_$fam.log | tail -1)" | tee -a $O/summary.txt
  [ -n "$fam" ] && unset $fam
done
grep -n "FAILED\|^E " $O/*.log | head -20
export BSN_LIB_PATH=$GRAFT_REPO_ROOT/bigsnpr_amd/libbigsnpr_hip_abl.so
for rep in 1 2; do
  for fam in BSN_LD_LUT BSN_LD_I8; do
    for pr in prio noprio; do
      unset BSN_LD_LUT BSN_LD_I8 BSN_LD_NOPRIO
      export $fam=1
      [ $pr = noprio ] && export BSN_LD_NOPRIO=1
      timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_${fam}_${pr}_$rep.json 2> /dev/null
      python -c "
import json; d=json.loads(open('$O/ld_${fam}_${pr}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('C5 $fam $pr rep $rep: %.1f ms per bed_ld_scores' % d['ms_per_step'], 'kernels %.1f ms' % r['ms_all_launches'], r['kernel'][:34])" | tee -a $O/summary.txt
    done
  done
done
