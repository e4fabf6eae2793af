#!/bin/bash
# round 3: the collectives one sharded solve issues (RCCL's entry points served by tests/native/mock_rccl.cpp, two ranks on one GPU)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export BSN_RCCL_LIBRARY=$PWD/tests/native/libmock_rccl.so MOCK_RCCL_TRACE=1
for blk in 0 8; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 1 --warmup 0 --samples 400000 --variants 600000 --block $blk --no-uv --no-cpu-baseline --no-ingest 2> /tmp/tr.txt | python -c "
import json,sys; d=json.load(sys.stdin); print('block arg $blk -> 2 ranks (mock transport, one GPU):', d['niter'], 'block steps, block', d['config']['block'], ',', d['passes_per_solve'], 'passes;', d['config']['parallelism'][:40], '; sigma1 %.6f' % d['sigma'][0])"
grep "mock rccl" /tmp/tr.txt | sed 's/ stream.*//' | sort | uniq -c | sort -rn | head -20
echo "streams used:"; grep "mock rccl" /tmp/tr.txt | sed 's/.* stream //' | sort | uniq -c
done
