#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for ky in 0 5 8 11 13 16 21 32 48; do
  if [ $ky = 0 ]; then unset BSN_KY; else export BSN_KY=$ky; fi
  timeout 300 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8 --slices 2 --reps 8 2>&1 | grep '"prod"' | sed "s/^/ky $ky: /"
done
