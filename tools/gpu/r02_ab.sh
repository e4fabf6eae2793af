#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
P="python tools/probe_matvec.py --n 400000 --m 1000000 --nvecs 8 --slices 2 --reps 12"
for rep in 1 2 3; do
for t in 0 42 46; do BSN_PROBE_TILE=1 BSN_TUNE=$t timeout 300 $P 2>&1 | grep '"cprod"' | sed "s/^/tiled tune $t: /"; done
done
