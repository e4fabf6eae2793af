#!/bin/bash
# round 6, second session: ten further draws of the random-shape parity suite and a hundred of the out-of-core fuzz on the last build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06draws; mkdir -p $O; : > $O/summary.txt
for off in 16 17 18 19 20 21 22 23 24 25; do
  BSN_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random_shapes.py -x -q -m gpu 2>&1 | tail -3 > $O/random_shapes_offset_$off.txt
  echo "random shapes, offset $off: $(tail -1 $O/random_shapes_offset_$off.txt)" | tee -a $O/summary.txt
done
timeout 2400 python tools/fuzz_out_of_core.py 300 100 > $O/fuzz_out_of_core.txt 2>&1
echo "fuzz_out_of_core 300..399: rc=$? $(tail -1 $O/fuzz_out_of_core.txt)" | tee -a $O/summary.txt
grep -A3 "MISMATCH\|itself failed" $O/fuzz_out_of_core.txt | cut -c1-400 | head -40
