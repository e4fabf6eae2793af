#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for m in 125000 250000 500000; do
BSN_TIMING=1 timeout 600 python bench.py --variants $m --steps 5 --warmup 2 --no-cpu-baseline --no-ingest 2> /tmp/err.txt | python -c "
import json,sys; d=json.load(sys.stdin); print('m=$m:', round(d['ms_per_step'],2),'ms', 'passes', round(d['passes_per_solve'],2), 'niter', d['niter'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, 'stream-only ms', round(sum(v['avg_ms']*v['launches'] for v in d['roofline']['other'].values())/d['steps'],1))"
grep -i "phase\|timing\|alloc" /tmp/err.txt | tail -3
done
