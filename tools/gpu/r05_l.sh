#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python tools/probe_autosvd.py 2>&1 | tail -14 | tee $O/autosvd.txt
timeout 900 python tools/probe_autosvd.py --m 1000000 2>&1 | tail -3 | tee -a $O/autosvd.txt
