#!/bin/bash
# round 4, trip E: LD band in blocks of columns, then the whole GPU suite; per-rank configuration of an 8-GPU run on one GPU
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ld.py -x -q 2>&1 | grep -v "^RCCL" | tail -25 | tee $O/tests_ld.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL" | tail -8 | tee $O/tests_all.txt
# the TRUE per-rank configuration of the 8-GPU run: 400K x 125K shard, m_total = 1e6 (warm start, 16-vector block), 1-rank RCCL
timeout 300 python bench.py --shard-of 8 --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-ingest --no-wide > $O/shard125k_block16.json 2> $O/shard.err
python - <<P
import json
d=json.load(open('$O/shard125k_block16.json')); print('shard-of-8:', round(d['ms_per_step'],2),'ms passes', d['passes_per_solve'], 'niter', d['niter'], 'block', d['config']['block'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})
P
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p125 -o st -- python $GRAFT_REPO_ROOT/bench.py --shard-of 8 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --no-wide > /dev/null 2> /tmp/p125.err
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/p125 -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f > $O/shard125k_block16_timeline.txt; tail -4 $O/shard125k_block16_timeline.txt
