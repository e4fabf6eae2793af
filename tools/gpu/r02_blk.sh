#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for b in 8 16 12 10; do
timeout 600 python bench.py --block $b --steps 4 --warmup 1 --no-cpu-baseline --no-ingest 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('block $b:', round(d['ms_per_step'],1),'ms', 'passes', d['passes_per_solve'], 'niter', d['niter'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['sigma'][:2])"
done
