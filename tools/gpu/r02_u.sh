#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ld.py tests/test_gpu_sct.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 600 python tools/probe_tcross.py --n 8192 --m 32768 2>&1 | grep entry
timeout 600 python tools/probe_tcross.py --n 16384 --m 65536 --reps 1 2>&1 | grep entry
