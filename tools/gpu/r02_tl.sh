#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_abi.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25
for t in 0 1; do
if [ $t = 1 ]; then export BSN_NO_TILED=1; fi
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('no_tiled=$t:', round(d['ms_per_step'],1),'ms', 'frac', round(d['roofline']['frac'],4), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['image_layout'][:40], d['sigma'][:2])"
done
