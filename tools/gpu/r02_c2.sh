#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02c2
timeout 600 python bench.py --workload matvec --steps 50 --warmup 3 > gpurun_out/r02c2/c2.json 2> gpurun_out/r02c2/c2.err; echo rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r02c2/c2.json')); print(round(d['ms_per_call'],3),'ms per call', '%.3g'%d['value'], d['unit'], 'frac', round(d['roofline']['frac'],3)); c=d['cpu_baseline']; print('cpu', '%.3g'%c['value'], c['cores'], c['seconds'], 'ratio', round(d['gpu_over_cpu']))"
tail -3 gpurun_out/r02c2/c2.err
