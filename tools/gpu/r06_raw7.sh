#!/bin/bash
# round 6, second session: one launch per run for the fused FP4 LD kernels against launches of 4 096 blocks (BSN_LD_BATCH, profiling build),
# same box, alternating; LD tests on the product build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06raw7; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_ld.py tests/test_gpu_out_of_core_random.py tests/test_gpu_autosvd.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
export BSN_LIB_PATH=$GRAFT_REPO_ROOT/bigsnpr_amd/libbigsnpr_hip_abl.so
for rep in 1 2; do
  for tag in one 4096 3840 7680; do
    unset BSN_LD_BATCH
    [ $tag != one ] && export BSN_LD_BATCH=$tag
    timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_${tag}_$rep.json 2> /dev/null
    python -c "
import json; d=json.loads(open('$O/ld_${tag}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('C5 batch $tag rep $rep: %.1f ms per bed_ld_scores' % d['ms_per_step'], 'kernels %.1f ms over %d launches' % (r['ms_all_launches'], r['launches']), 'frac', round(r['frac'],3), {k:round(v,1) for k,v in d.items() if 'cor' in k})" | tee -a $O/summary.txt
  done
done
unset BSN_LD_BATCH BSN_LIB_PATH
timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> /dev/null
timeout 600 python tools/probe_clump.py 100000 2>&1 | tee -a $O/summary.txt
