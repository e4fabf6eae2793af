#!/bin/bash
# round 6, trip Y: the four-product variant of k_pair_stats_f4 for the bed clumping formula: tests, then bed_autoSVD at 400K x 1M
# with the previous library (tools/ab/libbigsnpr_hip_old.so, by hand from the previous commit) and the current one, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06y; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ld.py tests/test_gpu_complete_data.py tests/test_gpu_random_shapes.py tests/test_gpu_autosvd.py tests/test_gpu_out_of_core.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export BSN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libbigsnpr_hip_old.so; else unset BSN_LIB_PATH; fi
    timeout 900 python tools/probe_autosvd.py --m 1000000 --bed > $O/bed_autosvd_${lib}_$rep.txt 2>&1
    echo "$lib $rep: $(grep -A1 'third call' $O/bed_autosvd_${lib}_$rep.txt | cut -c1-300 | tr '\n' ' ')"
  done
done
