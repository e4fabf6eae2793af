#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02fb
timeout 600 python bench.py --steps 3 --warmup 1 --force-dist --no-cpu-baseline --no-ingest --m 250000 2> gpurun_out/r02fb/fd.err | python -c "
import json,sys; d=json.load(sys.stdin); print('force-dist:', round(d['ms_per_step'],1), d['config']['parallelism'][:60])"
tail -3 gpurun_out/r02fb/fd.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --variants 250000 2> gpurun_out/r02fb/two.err | python -c "
import json,sys; d=json.load(sys.stdin); print('2 ranks on one GPU:', round(d['ms_per_step'],1), d['n_gpus'], d['niter'], d['converged'], d['config']['parallelism'])"
grep -i "fall\|error\|Traceback" gpurun_out/r02fb/two.err | head -5
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --m 250000 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('1 rank:', round(d['ms_per_step'],1), d['niter'], d['sigma'][:2])"
