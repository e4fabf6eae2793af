#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02l
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02l/pytest.log 2>&1
grep -n "Fatal\|File \"/tmp/code\|File \"/root/repo\|passed\|failed" gpurun_out/r02l/pytest.log | head -20
timeout 300 python bench.py --workload ld --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('ld', d['ms_per_step'], 'cor', d['bed_cor_ms'], d['roofline']['frac'])"
for args in "" "--force-dist" "--m 125000 --force-dist"; do timeout 300 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --no-ingest 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$args', d['ms_per_step'], d['niter'], round(d['passes_per_solve'],3), d['value'], round(d['hbm_frac_whole_solve'],4), round(d['roofline']['frac'],4), d['warm_start'])"; done
