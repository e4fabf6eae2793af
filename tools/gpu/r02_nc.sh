#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python tools/probe_matvec.py --n 400000 --m 250000 --nvecs 8 --slices 2 --reps 8 2>&1 | grep '"cprod"\|"prod"'
timeout 600 python tools/probe_matvec.py --n 400000 --m 250000 --nvecs 8 --slices 2 --reps 8 --subset 0.5 2>&1 | grep '"cprod"\|"prod"\|subset'
timeout 600 python tools/probe_matvec.py --n 400000 --m 250000 --nvecs 8 --slices 2 --reps 8 --subset 0.15 2>&1 | grep '"cprod"\|"prod"\|subset'
