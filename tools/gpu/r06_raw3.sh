#!/bin/bash
# round 6, second session: k_pair_stats_f4<., RAW> without the keep-mask when every sample is selected (MASK = false) and with
# s_setprio around its MFMA groups (PRIO) — same-box A/B at C5 on the profiling build (the switches are abl_getenv), LD tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06raw3; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_ld.py tests/test_gpu_out_of_core.py tests/test_gpu_autosvd.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
export BSN_LIB_PATH=$GRAFT_REPO_ROOT/bigsnpr_amd/libbigsnpr_hip_abl.so
for rep in 1 2; do
  for tag in nomask mask nomask_prio; do
    unset BSN_LD_RAW_MASK BSN_LD_RAW_PRIO
    [ $tag = mask ] && export BSN_LD_RAW_MASK=1
    [ $tag = nomask_prio ] && export BSN_LD_RAW_PRIO=1
    timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_${tag}_$rep.json 2> /dev/null
    python -c "
import json; d=json.loads(open('$O/ld_${tag}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('C5 $tag $rep: %.1f ms per bed_ld_scores' % d['ms_per_step'], 'kernels %.1f ms over %d launches' % (r['ms_all_launches'], r['launches']), 'frac', round(r['frac'],3), {k:round(v,1) for k,v in d.items() if 'cor' in k})" | tee -a $O/summary.txt
  done
done
