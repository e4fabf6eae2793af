#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for t in 8 16 4 32; do
BSN_COPY_THREADS=$t timeout 300 python tools/probe_tcross.py --n 16384 --m 65536 --reps 2 2>&1 | grep entry | sed "s/^/threads $t: /" | cut -c1-110
BSN_COPY_THREADS=$t timeout 300 python bench.py --workload ld --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('threads $t: ld', round(d['ms_per_step'],1), 'cor', round(d['bed_cor_ms'],1))"
done
