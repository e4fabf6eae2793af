#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
nproc; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for t in default 1 3 5 7 default; do
  if [ $t = default ]; then unset BSN_COPY_THREADS; else export BSN_COPY_THREADS=$t; fi
  timeout 600 python bench.py --workload matvec --steps 50 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('copy threads $t: %.3f ms per call' % d['ms_per_call'])"
done
