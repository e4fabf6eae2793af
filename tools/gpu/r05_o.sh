#!/bin/bash
# round 5, trip O: FP4 LD kernels — 64 x 64 block (default) against 128 x 32 and the int8 kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05o; mkdir -p $O; : > $O/summary.txt
for t in "tests/test_gpu_ld.py" "tests/test_gpu_fullsize.py -k c5" "tests/test_gpu_random_shapes.py -k 'correlations'" "tests/test_gpu_sct.py" "tests/test_gpu_fbm.py"; do
  tag=$(echo "$t" | tr ' /' '__' | tr -d "'")
  timeout 1500 bash -c "python -m pytest $t -m gpu -q -x" > $O/$tag.log 2>&1
  echo "$t rc=$? $(grep -E 'passed|failed|error' $O/$tag.log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
for v in 2 1 0 2; do
  BSN_LD_F4=$v timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_$v.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/ld_$v.json')); print('BSN_LD_F4=$v', 'bed_ld_scores %.1f ms' % d['ms_per_step'], 'bed_cor %.1f ms' % d['bed_cor_ms'], 'kernel', d['roofline']['kernel'][:22], 'useful TOP/s %.0f' % d['roofline']['achieved'], 'frac', round(d['roofline']['frac'],3), 'launches ms', round(d['roofline']['ms_all_launches'],1))" | tee -a $O/summary.txt
done
