#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
BSN_TIMING=1 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ingest > $O/b1.json 2> $O/b1.err; grep "host wall\|timed" $O/b1.err | cut -c1-230
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
