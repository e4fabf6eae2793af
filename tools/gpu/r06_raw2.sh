#!/bin/bash
# round 6, second session: the LD test file with the raw-plane A/B test, the per-counter record of k_pair_stats_f4<., RAW> at C5
# (four --pmc passes, kernel-trace only, as tools/gpu/r06_q.sh), the C5 bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06raw2; mkdir -p $O; : > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_ld.py -m gpu -q -x > $O/test_gpu_ld.log 2>&1
echo "tests/test_gpu_ld.py rc=$? $(grep -E 'passed|failed|error' $O/test_gpu_ld.log | tail -1)" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/test_gpu_ld.log | head -20
timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> /dev/null
python -c "
import json; d=json.loads(open('$O/ld_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('C5: %.1f ms per bed_ld_scores' % d['ms_per_step'], 'kernels %.1f ms over %d launches' % (r['ms_all_launches'], r['launches']), 'frac', round(r['frac'],3), {k:v for k,v in d.items() if 'cor' in k})" | tee -a $O/summary.txt
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc -o g$i -- python $R/tools/probe_ld_complete.py > $O/pmc_g$i.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc "k_pair|k_ld_sum" > $O/ld_pmc_summary.txt 2>&1
rm -rf $O/pmc
head -30 $O/ld_pmc_summary.txt | cut -c1-400
