#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/tests
timeout 1200 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q -x "$@" > gpurun_out/tests/rand.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/tests/rand.log | tail -40
