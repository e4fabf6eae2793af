#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02s
timeout 600 python -m pytest tests/test_gpu_pcadapt.py tests/test_gpu_prs_tcrossprod.py tests/test_gpu_matvec.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
timeout 900 python tools/probe_rows.py --skip-fbm 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r02s/rows.txt
