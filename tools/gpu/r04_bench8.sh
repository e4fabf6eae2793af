#!/bin/bash
# the driver's N = 8 launch line with the stand-in transport on one GPU: 8 ranks x 50 000 of 400 000 variants (40 000 samples: sample blocks of 5 120 rows, product pass in segments; k = 20 -> 16 vectors,
# warm start on, as in the real C4 run), against the single-rank solve of the same matrix
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -c "
import sys; sys.path.insert(0,'tests/native'); import build_native; print(build_native.build_mock_rccl())" > /tmp/mock.path
A="--samples 40000 --variants 400000 --steps 2 --warmup 1 --no-cpu-baseline --no-ingest"
timeout 600 python bench.py --gpus 1 $A 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('1 rank : niter', d['niter'], 'block', d['config']['block'], 'passes %.3f' % d['passes_per_solve'], 'conv', d['converged'], 'sigma1 %.9f' % d['sigma'][0], 'sigma5 %.9f' % d['sigma'][4])"
BSN_RCCL_LIBRARY=$(cat /tmp/mock.path) timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 8 $A 2>/tmp/b8.err | python -c "import json,sys; d=json.load(sys.stdin); print('8 ranks: niter', d['niter'], 'block', d['config']['block'], 'passes %.3f' % d['passes_per_solve'], 'conv', d['converged'], 'sigma1 %.9f' % d['sigma'][0], 'sigma5 %.9f' % d['sigma'][4], '|', d['config']['parallelism'][:60], '| exchange', d['exchange'], '| scaling', d['scaling'], 'n_gpus', d['n_gpus'])"
grep -i "error\|fall" /tmp/b8.err | head -5
