#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "1 8" "2 8" "1 4" "2 4" "1 32" "1 2"; do set -- $cfg; echo "== warm $1 den $2"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --verbose 1 --warm-start $1 --warm-den $2 2>&1 >/tmp/b.json | grep "step [3-7] " | tail -4 | sed -e 's/.*step/step/'; python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['ms_per_step'], d['niter'], round(d['passes_per_solve'],3), d['warm_start']['ms'])"; done
for k in 10 20 40; do for w in -1 0; do echo "== k $k warm $w n=100000 m=600000"; timeout 300 python bench.py --n 100000 --m 600000 --k $k --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --warm-start $w 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['niter'], round(d['passes_per_solve'],3))"; done; done
