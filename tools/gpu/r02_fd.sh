#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python bench.py --steps 2 --warmup 1 --force-dist --no-cpu-baseline --no-ingest --variants 250000 2>/tmp/e.txt | python -c "
import json,sys; d=json.load(sys.stdin); print('force-dist:', round(d['ms_per_step'],1), d['config']['parallelism'][:70])"
grep -i "error\|fall\|Trace" /tmp/e.txt | head -3
timeout 900 python -m pytest tests/test_gpu_bench_launch.py tests/test_gpu_comm.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
