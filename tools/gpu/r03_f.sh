#!/bin/bash
# round 3: residual trajectories of the block-16 solve under warm-start variants
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
for cfg in "16 1 16" "16 1 8" "16 1 4" "16 2 16" "16 2 8" "16 2 4" "16 3 8" "12 1 8" "8 1 16" "8 2 8"; do
set -- $cfg
timeout 300 python bench.py --block $1 --warm-start $2 --warm-den $3 --steps 2 --warmup 1 --verbose 1 --no-cpu-baseline --no-ingest > $O/o.json 2> $O/o.err
python - <<P
import json
d=json.load(open('$O/o.json')); print('block $1 warm $2 den $3:', round(d['ms_per_step'],2),'ms passes', round(d['passes_per_solve'],3), 'niter', d['niter'], 'conv', d['converged'])
P
grep "max rel resid" $O/o.err | tail -6 | awk '{printf "%s ", $(NF-2)} END{print ""}'
done
