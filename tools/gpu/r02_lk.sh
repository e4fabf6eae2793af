#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python tools/probe_leak.py 2>&1 | grep "iteration"
