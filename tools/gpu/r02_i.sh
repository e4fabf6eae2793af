#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $O/pytest.log; tail -25 $O/pytest.log
