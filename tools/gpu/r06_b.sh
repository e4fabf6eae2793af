#!/bin/bash
# round 6, trip B: FP6 micro-benchmark (second structure), vector-accuracy probe, the regression tests of this morning's fixes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
( cd tools/ubench && timeout 300 ./fp6_parts 12 ) > $O/fp6_parts.txt 2>&1
timeout 900 python tools/probe_vec_c3.py > $O/vec_c3.txt 2> $O/vec_c3.err
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -m gpu -k "regather or compacted" > $O/pytest_svd.txt 2>&1
tail -5 $O/pytest_svd.txt
