#!/bin/bash
# round 6, trip A: (1) FP6 x FP4 micro-benchmark, (2) what limits u / v at C3, (3) the cold first call, (4) per-rank shard lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
( cd tools/ubench && timeout 300 ./fp6_parts 12 ) > $O/fp6_parts.txt 2>&1
timeout 600 python tools/probe_vec_c3.py > $O/vec_c3.txt 2> $O/vec_c3.err
BSN_TIMING=1 BSN_ALLOC_TRACE=1 timeout 300 python tools/probe_cold.py > $O/cold.txt 2> $O/cold.err
BSN_NO_SMAJ=1 timeout 300 python tools/probe_cold.py > $O/cold_nosmaj.txt 2> $O/cold_nosmaj.err
for N in 2 4 8; do
  timeout 300 python bench.py --steps 8 --warmup 2 --shard-of $N --force-dist --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/shard_$N.json 2> $O/shard_$N.err
done
tail -3 $O/fp6_parts.txt; tail -2 $O/cold.txt
