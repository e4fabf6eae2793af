#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04tests; mkdir -p $O
for f in tests/test_gpu_fullsize.py tests/test_gpu_plink_io.py tests/test_gpu_tiled.py tests/test_gpu_smaj.py tests/test_gpu_comm.py tests/test_gpu_sharded_svd.py; do
  timeout 1200 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary2.txt
done
