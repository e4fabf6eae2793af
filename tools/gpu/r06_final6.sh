#!/bin/bash
# round 6, second session: after the raw-plane LD kernel — the GPU suite in ONE pytest process as the driver runs it, smoke(),
# three further draws of the random-shape suite (LD / clumping cases go through the new kernel where the band is large enough)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06final6; mkdir -p $O; : > $O/summary.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/suite_one_process.log 2>&1
echo "pytest tests/ -x -q -m gpu (one process): rc=$? $(grep -E 'passed|failed' $O/suite_one_process.log | tail -1) wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/suite_one_process.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
for off in 13 14 15; do
  BSN_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random_shapes.py tests/test_gpu_out_of_core_random.py -x -q -m gpu 2>&1 | tail -3 > $O/random_shapes_offset_$off.txt
  echo "random shapes + out-of-core draws, offset $off: $(tail -1 $O/random_shapes_offset_$off.txt)" | tee -a $O/summary.txt
done
