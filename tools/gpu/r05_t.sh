#!/bin/bash
# VERDICT r4 #5b, the two digit-layout experiments (profiling build, timing only) + the NA-skip test again
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05t; mkdir -p $O; : > $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_na_skip.py -m gpu -q -x > $O/test_gpu_na_skip.log 2>&1
echo "na_skip rc=$? $(grep -E 'passed|failed|error' $O/test_gpu_na_skip.log | tail -1)" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/*.log | head
export BSN_LIB_PATH=$GRAFT_REPO_ROOT/bigsnpr_amd/libbigsnpr_hip_abl.so
for rep in 1 2; do
for d in 0 1 2; do
  BSN_DIGITS=$d timeout 300 python tools/probe_power.py --n 400000 --m 500000 --seconds 4 --slices 3 --only16 --tag "BSN_DIGITS=$d" 2>&1 | grep '^{' | tee -a $O/power_digits.txt
done
done
