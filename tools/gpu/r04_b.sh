#!/bin/bash
# round 4, trip B: k_cprod32 (32 x 32 x 32 MFMA) against k_cprod<2>, operand-role swap of k_prod<2>; 50 GB shard, 16 x 2
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for t in 0 131 132 133 134 135 136 137 122 174 0 131; do
  echo "BSN_TUNE=$t"; BSN_TUNE=$t timeout 120 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 16 --slices 2 --reps 8 2>&1 | grep '"cprod"\|"prod"' | tee -a $O/ablation.txt
done
echo "== full size"
for t in 0 131 132; do
  echo "BSN_TUNE=$t"; BSN_TUNE=$t timeout 200 python tools/probe_matvec.py --n 400000 --m 1000000 --nvecs 16 --slices 2 --reps 6 2>&1 | grep '"cprod"\|"prod"' | tee -a $O/full.txt
done
