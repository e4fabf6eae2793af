#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05z; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_bench_launch.py -m gpu -q -x > $O/test_gpu_bench_launch.log 2>&1
echo "bench_launch rc=$? $(tail -1 $O/test_gpu_bench_launch.log)"
grep -n "^E  \|FAILED" $O/test_gpu_bench_launch.log | head -20
