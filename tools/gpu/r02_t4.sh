#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_svd.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for rep in 1 2; do
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ingest 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('tiled:', round(d['ms_per_step'],1),'ms', 'frac', round(d['roofline']['frac'],4), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})"
BSN_NO_TILED=1 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ingest 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('plain:', round(d['ms_per_step'],1),'ms', 'frac', round(d['roofline']['frac'],4), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})"
done
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
P="python tools/probe_matvec.py --n 400000 --m 1000000 --nvecs 8 --slices 2 --reps 12"
for t in 42 47 48; do BSN_PROBE_TILE=1 BSN_TUNE=$t timeout 300 $P 2>&1 | grep '"cprod"' | sed "s/^/tiled tune $t: /"; done
