#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
P="python tools/probe_matvec.py --n 400000 --m 1000000 --nvecs 8 --slices 2 --reps 12"
for rep in 1 2; do
BSN_PROBE_TILE=1 BSN_TUNE=0 timeout 300 $P 2>&1 | grep '"prod"' | sed "s/^/tiled ky 11: /"
BSN_PROBE_TILE=1 BSN_TUNE=0 BSN_KY=16 timeout 300 $P 2>&1 | grep '"prod"' | sed "s/^/tiled ky 16: /"
BSN_PROBE_TILE=1 BSN_TUNE=77 BSN_KY=16 timeout 300 $P 2>&1 | grep '"prod"' | sed "s/^/tiled ky 16 xcd map: /"
BSN_PROBE_TILE=1 BSN_TUNE=77 BSN_KY=8 timeout 300 $P 2>&1 | grep '"prod"' | sed "s/^/tiled ky 8 xcd map: /"
done
