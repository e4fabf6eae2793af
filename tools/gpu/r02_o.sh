#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ld.py tests/test_gpu_svd.py tests/test_gpu_sharded_svd.py -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -2
for t in 0 24; do BSN_LD_TILE=$t timeout 300 python bench.py --workload ld --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('tile $t: ld', d['ms_per_step'], 'cor', d['bed_cor_ms'], d['roofline']['frac'])"; done
BSN_LD_TILE=24 timeout 600 python -m pytest tests/test_gpu_ld.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -2
