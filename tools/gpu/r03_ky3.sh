#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for rep in 1 2; do for ky in 4 5 9 11; do
  BSN_KY=$ky timeout 300 python tools/probe_matvec.py --n 400000 --m 125000 --nvecs 16 --slices 2 --reps 8 2>&1 | grep '"prod"' | sed "s/^/NB2 m 125000 ky $ky: /"
done; done
for rep in 1 2; do for ky in 5 9 11 13; do
  BSN_KY=$ky timeout 300 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 16 --slices 2 --reps 6 2>&1 | grep '"prod"' | sed "s/^/NB2 m 500000 ky $ky: /"
done; done
