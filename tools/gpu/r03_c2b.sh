#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
run() { timeout 300 python tools/probe_calls.py --reps 30 2>/dev/null | tr '\n' ' '; echo; }
echo "base:"; run
for t in 21 24 25 26; do echo "BSN_TUNE=$t (cprod shape):"; BSN_TUNE=$t run; done
for k in 12 16 24 32 48; do echo "BSN_KY=$k (prod K split):"; BSN_KY=$k run; done
