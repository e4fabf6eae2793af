#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05p; mkdir -p $O; : > $O/summary.txt
for v in lut fromx i8 lut; do
  unset BSN_LD_X2_FROM_X BSN_LD_I8
  [ $v = fromx ] && export BSN_LD_X2_FROM_X=1
  [ $v = i8 ] && export BSN_LD_I8=1
  timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_$v.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/ld_$v.json')); print('$v', 'bed_ld_scores %.1f ms' % d['ms_per_step'], 'bed_cor %.1f ms' % d['bed_cor_ms'], 'kernel', d['roofline']['kernel'][:22], 'useful TOP/s %.0f' % d['roofline']['achieved'], 'frac', round(d['roofline']['frac'],3), 'of int8 peak', round(d['roofline']['frac_of_int8_peak'],3), 'launches ms', round(d['roofline']['ms_all_launches'],1))" | tee -a $O/summary.txt
done
unset BSN_LD_X2_FROM_X BSN_LD_I8
timeout 900 python -m pytest tests/test_gpu_ld.py tests/test_gpu_fbm.py -m gpu -q -x 2>&1 | tail -1 | tee -a $O/summary.txt
# counters of the FP4 kernel
cd /tmp && for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1)); timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r05_ld -o g$i -- python $GRAFT_REPO_ROOT/tools/probe_ld.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py gpurun_out/pmc_r05_ld 2>/dev/null | grep -A16 "k_pair_stats_f4" | head -40 | tee $O/pmc_ld.txt | head -20
