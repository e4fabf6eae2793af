#!/bin/bash
# round 6, second session: the six pairwise-complete sums from look-up-free planes (k_pair_stats_f4<., RAW>): the LD test files,
# then the same-box A/B against the look-up kernel (BSN_LD_LUT=1) at C5 (bed_ld_scores / bed_cor, 1 % missing values), the
# bed clumping band at 400K x 100K and bed_autoSVD at 400K x 1M with 1 % missing values
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06raw; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_ld.py tests/test_gpu_complete_data.py tests/test_gpu_out_of_core.py tests/test_gpu_autosvd.py tests/test_gpu_sct.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -30
for rep in 1 2; do
  for tag in raw lut; do
    if [ $tag = lut ]; then export BSN_LD_LUT=1; else unset BSN_LD_LUT; fi
    timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_${tag}_$rep.json 2> /dev/null
    python -c "
import json; d=json.loads(open('$O/ld_${tag}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('C5 $tag $rep: %.1f ms per bed_ld_scores' % d['ms_per_step'], 'kernels %.1f ms over %d launches' % (r['ms_all_launches'], r['launches']), 'frac', round(r['frac'],3), {k:v for k,v in d.items() if 'cor' in k}, r['kernel'][:40])" | tee -a $O/summary.txt
  done
done
unset BSN_LD_LUT
for tag in raw lut; do
  if [ $tag = lut ]; then export BSN_LD_LUT=1; else unset BSN_LD_LUT; fi
  timeout 600 python tools/probe_clump.py 100000 2>&1 | sed "s/^/$tag: /" | tee -a $O/summary.txt
  timeout 900 python tools/probe_autosvd.py --m 1000000 --bed > $O/autosvd_bed_$tag.txt 2>&1
  grep -i "total\|call\|clump" $O/autosvd_bed_$tag.txt | tail -8 | sed "s/^/$tag: /" | cut -c1-250 | tee -a $O/summary.txt
done
