#!/bin/bash
# round 4, trip C: operand roles of k_cprod (digits as the A operand of the MFMA), 50 GB shard
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for t in 0 124 125 126 0 124 137; do
  echo "BSN_TUNE=$t"; BSN_TUNE=$t timeout 120 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 16 --slices 2 --reps 8 2>&1 | grep '"cprod"' | tee -a $O/ablation.txt
done
echo "== one column block (8 vectors x 2 slices)"
for t in 0 141 142 0 141; do
  echo "BSN_TUNE=$t"; BSN_TUNE=$t timeout 120 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8 --slices 2 --reps 8 2>&1 | grep '"cprod"' | tee -a $O/ablation8.txt
done
echo "== single vector, 7 slices"
for t in 0 141; do
  echo "BSN_TUNE=$t"; BSN_TUNE=$t timeout 120 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 1 --slices 7 --reps 8 2>&1 | grep '"cprod"' | tee -a $O/ablation1.txt
done
