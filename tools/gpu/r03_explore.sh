#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/tests
for off in "$@"; do
  BSN_TEST_SEED_OFFSET=$off timeout 600 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q -k "not many" > gpurun_out/tests/explore_$off.log 2>&1
  echo "offset $off: $(tail -1 gpurun_out/tests/explore_$off.log)"
  grep -E "^FAILED|^ERROR" gpurun_out/tests/explore_$off.log | head -10
done
