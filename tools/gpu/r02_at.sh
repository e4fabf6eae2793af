#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
BSN_ALLOC_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-ingest --verbose 0 2> /tmp/at.txt > /dev/null
grep -n "bsn alloc\|warmup done\|timed solves" /tmp/at.txt | awk '/warmup done/{f=1} f' | head -20
echo "allocs before warmup done: $(awk '/warmup done/{exit} /bsn alloc/{c++} END{print c+0}' /tmp/at.txt)"
