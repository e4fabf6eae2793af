#!/bin/bash
# round 6, trip I: out-of-core tests; cold timings after the pool / staging changes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_out_of_core.py tests/test_gpu_smaj.py tests/test_gpu_ld.py tests/test_gpu_sct.py tests/test_gpu_plink_io.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for rep in 1 2 3; do
BSN_TIMING=1 BSN_ALLOC_TRACE=1 timeout 300 python tools/probe_cold.py > $O/cold_$rep.txt 2> $O/cold_$rep.err
done
grep -h "solve_ms" $O/cold_*.txt | cut -c1-200
grep -h "helper thread\|host wall" $O/cold_*.err | cut -c1-250
