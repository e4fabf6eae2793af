#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
BSN_ALLOC_TRACE=1 timeout 300 python tools/probe_first_calls.py 2>&1 | grep "prod\|alloc" | tail -60
