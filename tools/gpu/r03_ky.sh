#!/bin/bash
# new K-split rule of k_prod: one-shot calls at C2, default solve at C3, 125K shard
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03ky; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_edge_cases.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -2
for r in 1 2; do timeout 600 python bench.py --workload matvec --steps 50 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('C2: %.3f ms per call' % d['ms_per_call'])"; done
one() { l=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-ingest > $O/$l.json 2> $O/$l.err
  python - <<P
import json
try:
  d=json.load(open('$O/$l.json')); print('$l:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],2), 'niter', d['niter'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})
except Exception as e: print('$l: FAILED', e)
P
}
one default --steps 5 --warmup 1
one b8 --block 8 --steps 4 --warmup 1
one shard --m 125000 --steps 8 --warmup 2
