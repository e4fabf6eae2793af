#!/bin/bash
# round 4, trip I: cost of running the product pass in segments (1-rank RCCL, full size) — overlap / one stream / whole pass
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
for e in "" "BSN_NO_OVERLAP=1" "BSN_NO_SEGMENTS=1" "" "BSN_NO_SEGMENTS=1"; do env $e timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-wide 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$e]', round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['sigma'][0])"; done | tee $O/segments_1rank.txt
for e in "" "BSN_NO_SEGMENTS=1"; do env $e timeout 300 python bench.py --shard-of 8 --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-ingest --no-wide 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shard-of-8 [$e]', round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['sigma'][0])"; done | tee -a $O/segments_1rank.txt
