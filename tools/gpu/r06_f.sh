#!/bin/bash
# round 6, trip F: where do the 2.5 s of the first solve go when the copy is made beside it?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
for rep in 1 2; do
BSN_TIMING=1 BSN_ALLOC_TRACE=1 timeout 300 python tools/probe_cold.py > $O/cold_async_$rep.txt 2> $O/cold_async_$rep.err
BSN_SMAJ_SYNC=1 BSN_TIMING=1 BSN_ALLOC_TRACE=1 timeout 300 python tools/probe_cold.py > $O/cold_sync_$rep.txt 2> $O/cold_sync_$rep.err
BSN_NO_SMAJ=1 timeout 300 python tools/probe_cold.py > $O/cold_nosmaj_$rep.txt 2> $O/cold_nosmaj_$rep.err
done
grep -h "solve_ms" $O/*.txt | cut -c1-200
grep -h "helper thread\|host wall" $O/*.err | cut -c1-250
