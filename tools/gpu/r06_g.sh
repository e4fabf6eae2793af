#!/bin/bash
# round 6, trip G: out-of-core clumping / conversions / list solve / autoSVD; the copy behind the first solve; cold timings with allocation times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_out_of_core.py tests/test_gpu_smaj.py tests/test_gpu_ld.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for rep in 1 2; do
BSN_TIMING=1 BSN_ALLOC_TRACE=1 timeout 300 python tools/probe_cold.py > $O/cold_behind_$rep.txt 2> $O/cold_behind_$rep.err
done
grep -h "solve_ms" $O/cold_*.txt | cut -c1-200
grep -h "helper thread\|host wall" $O/cold_*.err | cut -c1-250
