#!/bin/bash
# round 6, second session: the GPU suite in ONE pytest process as the driver runs it (after the fix of the new random
# out-of-core test's budget rule), and the out-of-core fuzz on 60 further draws
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06final5; mkdir -p $O; : > $O/summary.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/suite_one_process.log 2>&1
echo "pytest tests/ -x -q -m gpu (one process): rc=$? $(grep -E 'passed|failed' $O/suite_one_process.log | tail -1) wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/suite_one_process.log | head -20
timeout 1500 python tools/fuzz_out_of_core.py 200 60 > $O/fuzz_out_of_core.txt 2>&1
echo "fuzz_out_of_core 200..259: rc=$? $(tail -1 $O/fuzz_out_of_core.txt)" | tee -a $O/summary.txt
grep -A3 "MISMATCH\|itself failed" $O/fuzz_out_of_core.txt | cut -c1-400 | head -60
