#!/bin/bash
# round 6, trip C: the FP6 question closed with counters — all-zero digit panels (no operand toggling) and PMC passes
# (clock from GRBM_GUI_ACTIVE, matrix-pipe busy, VALU per MFMA, LDS) over six skeleton configurations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c; mkdir -p $O
( cd tools/ubench && timeout 200 ./fp6_parts 12 1 1 ) > $O/fp6_zero_digits.txt 2>&1
( cd tools/ubench && timeout 200 ./fp6_parts 12 0 1 ) > $O/fp6_random_digits.txt 2>&1
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc -o g$i -- $R/tools/ubench/fp6_parts 2 0 1 > $O/pmc_g$i.log 2>&1
done
ls -R $O | head -30
