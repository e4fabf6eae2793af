#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02q
timeout 900 python tools/probe_rows.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r02q/rows.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
