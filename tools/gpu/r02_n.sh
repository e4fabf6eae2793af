#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "0 16" "2 4" "3 4" "4 4" "3 8" "4 8" "4 2" "3 2"; do set -- $cfg; echo "== thin $1 den $2"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --verbose 1 --thin-steps $1 --warm-den $2 2>&1 >/tmp/b.json | grep "step [1-9] " | tail -7 | sed -e 's/.*step/step/' | cut -c1-60 | tr '\n' ';'; echo; python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['ms_per_step'], d['niter'], round(d['passes_per_solve'],3), d['converged'], d['sigma'][0])"; done
