#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for m in 125000 500000; do for rep in 1 2 3; do for ky in 9 11; do
  BSN_KY=$ky timeout 300 python tools/probe_matvec.py --n 400000 --m $m --nvecs 8 --slices 2 --reps 8 2>&1 | grep '"prod"' | sed "s/^/m $m ky $ky: /"
done; done; done
