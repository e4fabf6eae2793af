#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1200 python tools/probe_clump_lazy.py 2>&1 | grep "bed_clumping\|Error\|error"
