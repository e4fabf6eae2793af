#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05last; mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_svd.py -m gpu -q -x > $O/edge_svd.log 2>&1
echo "edge+svd rc=$? $(tail -1 $O/edge_svd.log)"
grep -n "^E  \|^FAILED" $O/edge_svd.log | head -12
