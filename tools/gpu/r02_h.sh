#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
BSN_TIMING=1 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ingest 2>&1 >$O/b.json | grep "host wall\|timed" | sed -e 's/.*solve/solve/' | cut -c1-140
timeout 600 python -m pytest tests/test_gpu_fused_scaling.py tests/test_gpu_svd.py tests/test_gpu_complete_data.py -q 2>&1 | tail -2
