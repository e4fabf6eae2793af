#!/bin/bash
# round 6, trip P: the FP6 skeleton with the column blocks DIVIDED BETWEEN WAVES (every wave keeps <= 48 accumulators, four
# waves per SIMD stay resident) against the int8 shapes — times, then the counter groups of trip C
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p; mkdir -p $O
( cd tools/ubench && timeout 300 ./fp6_parts 12 0 1 ) > $O/fp6_roles.txt 2>&1
( cd tools/ubench && timeout 300 ./fp6_parts 12 1 1 ) > $O/fp6_roles_zero_digits.txt 2>&1
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc -o g$i -- $R/tools/ubench/fp6_parts 2 0 1 > $O/pmc_g$i.log 2>&1
done
cat $O/fp6_roles.txt
