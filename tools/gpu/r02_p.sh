#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02p
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for t in 0 31 32 33 81 82 83 0; do
  BSN_TUNE=$t timeout 300 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8 --slices 2 --reps 8 2>&1 | grep kernel | grep -v counts | tr '\n' ' ' | sed "s/^/tune $t: /"; echo
done | tee gpurun_out/r02p/nt.txt
unset BSN_LIB_PATH
timeout 900 python tools/probe_rows.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r02p/rows.txt
