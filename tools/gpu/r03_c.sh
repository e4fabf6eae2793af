#!/bin/bash
# round 3: tests, the 125K shard after the MFMA panel product, shape sweeps (ablation build) of the counting pass and the two-block kernels
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/gpu.log | tail -4
one() { # label, args...
  l=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-ingest > $O/$l.json 2> $O/$l.err
  python - <<P
import json
try:
  d=json.load(open('$O/$l.json')); print('$l:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['sigma'][:1])
except Exception as e: print('$l: FAILED', e)
P
}
BSN_TIMING=1 one b125 --variants 125000 --steps 6 --warmup 2
grep "host wall" $O/b125.err | tail -1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p125 -o st -- python $GRAFT_REPO_ROOT/bench.py --variants 125000 --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > /dev/null 2> /tmp/p125.err
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/p125 -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f > $O/timeline_125k.txt; tail -3 $O/timeline_125k.txt
one b8 --steps 4 --warmup 1
one b16 --block 16 --steps 4 --warmup 1
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for t in 41 43 45; do BSN_TUNE=$t one stats_t$t --steps 3 --warmup 1; done
for t in 0 91 92 93 94 95 96 97 98; do BSN_TUNE=$t one nb2_t$t --block 16 --steps 3 --warmup 1; done
