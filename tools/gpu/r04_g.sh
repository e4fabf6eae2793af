#!/bin/bash
# round 4, trip G: chunk-major sample-major copy: parity, bench with / without, kernel-trace stats of the default line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_smaj.py -x -q 2>&1 | grep -v "^RCCL" | tail -15 | tee $O/tests.txt
for e in "" "BSN_NO_SMAJ=1"; do env $e timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ingest --no-wide 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['sigma'][0], d['roofline']['kernels_launched'].get('prod'))"; done | tee $O/bench.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --no-wide > /dev/null 2> /tmp/pk.err
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/pk -name '*kernel_stats.csv' | head -1); head -14 $f | cut -c1-200 | tee $O/kernel_stats_head.txt; cp $f $O/kernel_stats.csv
