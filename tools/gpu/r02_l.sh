#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02l
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02l/pytest.log 2>&1
grep -n "Fatal\|File \"/tmp/code\|File \"/root/repo\|passed\|failed\|Error" gpurun_out/r02l/pytest.log | head -20
