#!/bin/bash
# the driver fix for fewer samples than the basis holds + more random-shape draws
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05v; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_edge_cases.py tests/test_gpu_svd.py tests/test_gpu_random_shapes.py; do
  timeout 900 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
for off in 3000 4000 5000 6000; do
  BSN_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random_shapes.py -q > $O/seed$off.log 2>&1
  echo "seed offset $off: $(tail -1 $O/seed$off.log)" | tee -a $O/summary.txt
  grep -n "^FAILED\|^E  " $O/seed$off.log | head -20
done
