#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r03cor
timeout 900 python -m pytest tests/test_gpu_ld.py tests/test_gpu_fbm.py tests/test_gpu_r_shim.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 600 python bench.py --workload ld --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03cor/ld.json 2> gpurun_out/r03cor/ld.err
python -c "
import json; d=json.load(open('gpurun_out/r03cor/ld.json')); print('ld_scores %.1f ms, bed_cor %.1f ms, frac %.3f' % (d['ms_per_step'], d['bed_cor_ms'], d['roofline']['frac']))"
