#!/bin/bash
# the remaining files that run the SVD driver, on the last sources
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05y; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_autosvd.py tests/test_gpu_fullsize.py tests/test_gpu_out_of_core.py tests/test_gpu_na_skip.py tests/test_gpu_r_shim.py tests/test_gpu_fused_scaling.py tests/test_gpu_pcadapt.py tests/test_gpu_tiled.py tests/test_gpu_smaj.py; do
  timeout 600 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
