#!/bin/bash
# round 6, second session: s_setprio around the fp64 MFMAs of k_tcross (profiling build, BSN_TCROSS_PRIO=1), n = 16384, m = 65536, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT" BSN_LIB_PATH=$GRAFT_REPO_ROOT/bigsnpr_amd/libbigsnpr_hip_abl.so
O=gpurun_out/r06tcross; mkdir -p $O; : > $O/summary.txt
for t in off on off on; do
  if [ $t = on ]; then export BSN_TCROSS_PRIO=1; else unset BSN_TCROSS_PRIO; fi
  timeout 300 python tools/probe_tcross.py --n 16384 --m 65536 2>/dev/null | tail -1 | sed "s/^/prio $t: /" | tee -a $O/summary.txt
done
