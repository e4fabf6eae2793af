#!/bin/bash
# the driver's N = 8 launch line at FULL size (400 000 x 1 000 000) with the stand-in transport on one GPU: does the sharded solve take the
# same number of block steps as the single-GPU solve?  (times mean nothing: eight ranks share one GPU, the stand-in copies through the host)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -c "
import sys; sys.path.insert(0,'tests/native'); import build_native; print(build_native.build_mock_rccl())" > /tmp/mock.path
A="--steps 1 --warmup 1 --no-cpu-baseline --no-ingest --no-wide --verbose 1"
BSN_RCCL_LIBRARY=$(cat /tmp/mock.path) timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 8 $A 2>/tmp/b8.err | python -c "import json,sys; d=json.load(sys.stdin); print('8 ranks: niter', d['niter'], 'block', d['config']['block'], 'passes %.3f' % d['passes_per_solve'], 'conv', d['converged'], 'sigma1 %.9f' % d['sigma'][0], 'sigma5 %.9f' % d['sigma'][4], '| exchange', d['exchange'], '| n_gpus', d['n_gpus'])"
grep -i "resid\|step" /tmp/b8.err | tail -12 | cut -c1-200
grep -i "error\|fall" /tmp/b8.err | head -5
