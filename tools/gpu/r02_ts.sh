#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
P="python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8 --slices 2 --reps 8"
$P 2>&1 | grep '"cprod"\|"prod"' | tr '\n' ' ' | sed "s/^/plain: /"; echo
for t in 0 41 42 43 44 45 46 0; do
  BSN_PROBE_TILE=1 BSN_TUNE=$t timeout 300 $P 2>&1 | grep '"cprod"\|"prod"' | tr '\n' ' ' | sed "s/^/tiled tune $t: /"; echo
done
