#!/bin/bash
# thick restart on an inexact exhaustion: the SVD files of the suite + two random-shape offsets
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05x; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_svd.py tests/test_gpu_edge_cases.py tests/test_gpu_random_shapes.py tests/test_gpu_complete_data.py tests/test_gpu_sharded_svd.py; do
  timeout 900 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
BSN_TEST_SEED_OFFSET=18000 timeout 600 python -m pytest tests/test_gpu_random_shapes.py -q > $O/seed18000.log 2>&1
echo "seed offset 18000: $(tail -1 $O/seed18000.log)" | tee -a $O/summary.txt
grep -n "^FAILED\|^E  " $O/seed18000.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
