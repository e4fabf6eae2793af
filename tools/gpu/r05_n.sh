#!/bin/bash
# round 5, trip N: the six LD products on the FP4 matrix pipe (k_pair_stats_f4) against the int8 kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05n; mkdir -p $O; : > $O/summary.txt
for t in "tests/test_gpu_ld.py" "tests/test_gpu_fullsize.py -k c5" "tests/test_gpu_random_shapes.py -k 'correlations'" "tests/test_gpu_sct.py" "tests/test_gpu_autosvd.py -k 'not dist_ogk and not medcouple'"; do
  tag=$(echo "$t" | tr ' /' '__' | tr -d "'")
  timeout 1500 bash -c "python -m pytest $t -m gpu -q -x" > $O/$tag.log 2>&1
  echo "$t rc=$? $(grep -E 'passed|failed|error' $O/$tag.log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
for v in f4 i8 f4 i8; do
  if [ $v = i8 ]; then export BSN_LD_I8=1; else unset BSN_LD_I8; fi
  timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_$v.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/ld_$v.json')); print('$v', 'bed_ld_scores %.1f ms' % d['ms_per_step'], 'bed_cor %.1f ms' % d['bed_cor_ms'], 'kernel', d['roofline']['kernel'][:30], 'int8-equivalent TOP/s %.0f' % d['roofline']['achieved'], 'launches ms', round(d['roofline']['ms_all_launches'],1))" | tee -a $O/summary.txt
done
