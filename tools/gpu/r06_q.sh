#!/bin/bash
# round 6, trip Q: (1) the per-counter record of the LD kernels at C5 (k_pair_stats_f4 with missing values, k_pair_xy_f4 on
# complete data) -> profiles/r06_ld_pmc.txt; (2) the random-shape parity suite on three other draws (the driver changed this round)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06q; mkdir -p $O
for off in 1 2 3; do
  BSN_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random_shapes.py -x -q -m gpu 2>&1 | tail -5 > $O/random_shapes_offset_$off.txt
done
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
python tools/pmc_summary.py $O/pmc "k_pair|k_ld_sum|k_colstats" > $O/ld_pmc_summary.txt 2>&1
rm -f $O/pmc/*kernel_trace.csv.bak
du -sh $O/pmc
cat $O/random_shapes_offset_*.txt; head -50 $O/ld_pmc_summary.txt
