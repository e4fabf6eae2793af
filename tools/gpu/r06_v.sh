#!/bin/bash
# round 6, trip V: A/B on ONE box — the library with round 6's first k_pair_xy_f4 (tools/ab/libbigsnpr_hip_old.so, built from
# the previous commit's ld.hip) against the current one, alternating: C5 on complete data and snp_autoSVD at 400K x 1M
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06v; mkdir -p $O
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export BSN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libbigsnpr_hip_old.so; else unset BSN_LIB_PATH; fi
    timeout 600 python tools/probe_ld_complete.py > $O/ld_complete_${lib}_$rep.txt 2>&1
    echo "$lib $rep: $(grep na16 $O/ld_complete_${lib}_$rep.txt | tr '\n' ' ')"
    timeout 900 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_${lib}_$rep.txt 2>&1
    echo "$lib $rep: $(grep -A1 'third call' $O/autosvd_${lib}_$rep.txt | cut -c1-300 | tr '\n' ' ')"
  done
done
