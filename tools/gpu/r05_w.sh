#!/bin/bash
# more random-shape draws (exploration)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05w; mkdir -p $O; : > $O/summary.txt
for off in ${OFFSETS:-7000 8000 9000 10000 11000 12000}; do
  BSN_TEST_SEED_OFFSET=$off timeout 600 python -m pytest tests/test_gpu_random_shapes.py -q > $O/seed$off.log 2>&1
  echo "seed offset $off: $(tail -1 $O/seed$off.log)" | tee -a $O/summary.txt
  grep -n "^FAILED\|^E  " $O/seed$off.log | head -12
done
