#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 300 python tools/probe_first_calls.py 2>&1 | grep "prod"
