#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 300 tools/ubench/shape 2>&1 | tail -20
