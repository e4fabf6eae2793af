#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/tests gpurun_out/r03final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests/gpu.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/tests/gpu.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > gpurun_out/r03final/ld.json 2> gpurun_out/r03final/ld.err
python -c "
import json; d=json.load(open('gpurun_out/r03final/ld.json')); print('ld_scores %.1f ms, bed_cor %.1f ms, frac %.3f' % (d['ms_per_step'], d['bed_cor_ms'], d['roofline']['frac']))"
