#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03ld; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ld.py tests/test_gpu_sct.py tests/test_gpu_fullsize.py tests/test_gpu_complete_data.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for v in shared old; do
  if [ $v = old ]; then export BSN_LD_NO_SHARED_DECODE=1; else unset BSN_LD_NO_SHARED_DECODE; fi
  timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_$v.json 2> $O/ld_$v.err
  python - <<P
import json; d=json.load(open('$O/ld_$v.json')); print('$v: ld_scores %.1f ms, cor %.1f ms, frac %.3f, kernel ms %.1f / %d launches; %s' % (d['ms_per_step'], d['bed_cor_ms'], d['roofline']['frac'], d['roofline']['ms_all_launches'], d['roofline']['launches'], d['roofline']['kernel'][:40]))
P
done
