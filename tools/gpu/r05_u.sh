#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05u; mkdir -p $O
BSN_TEST_SEED_OFFSET=3000 timeout 900 python -m pytest tests/test_gpu_random_shapes.py -q -x > $O/seed3000.log 2>&1
tail -80 $O/seed3000.log
