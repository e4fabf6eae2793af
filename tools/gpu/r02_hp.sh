#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cat /sys/kernel/mm/transparent_hugepage/enabled
for h in 0 1; do
if [ $h = 1 ]; then export BSN_NO_HUGEPAGE_HINT=1; fi
timeout 300 python tools/probe_tcross.py --n 16384 --m 65536 --reps 2 2>&1 | grep entry | sed "s/^/no_hint=$h: /" | cut -c1-100
timeout 300 python bench.py --workload ld --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('no_hint=$h: ld', round(d['ms_per_step'],1), 'cor', round(d['bed_cor_ms'],1))"
done
