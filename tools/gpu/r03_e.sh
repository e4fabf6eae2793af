#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "svd or comm or fused or complete or tiled or fullsize or bench" > $O/gpu.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/gpu.log | tail -4
BSN_TIMING=1 timeout 300 python bench.py --variants 125000 --steps 6 --warmup 2 --no-cpu-baseline --no-ingest > $O/b125.json 2> $O/b125.err
python - <<'P'
import json; d=json.load(open('gpurun_out/r03e/b125.json')); print('m=125000:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})
P
grep "host wall" $O/b125.err | tail -1
bash tools/gpu/r03_d.sh
cp gpurun_out/r03d/* $O/
