#!/bin/bash
# round 6, second session: the new streamed wide-band test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06raw6; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_out_of_core_random.py -m gpu -q -x > $O/test.log 2>&1
echo "rc=$? $(grep -E 'passed|failed|error' $O/test.log | tail -1)" | tee $O/summary.txt
grep -n "FAILED\|^E " $O/test.log | head -20
