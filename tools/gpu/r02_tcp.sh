#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r02tc; mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/$O/pmc -o g -- python $R/tools/probe_tcross.py --n 16384 --m 65536 --reps 1 > $R/$O/pmc.log 2>&1)
python tools/pmc_summary.py $O/pmc "k_tcross" 2>&1 | tail -12
