#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python tools/probe_matvec.py --dosage --n 50000 --m 200000 --nvecs 1,8 --slices 2 --reps 8 2>&1 | grep "kernel\|generate"
timeout 600 python tools/probe_matvec.py --dosage --n 400000 --m 100000 --nvecs 8 --slices 2 --reps 4 2>&1 | grep "kernel\|generate"
