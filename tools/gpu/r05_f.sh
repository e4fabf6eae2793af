#!/bin/bash
# round 5, trip F: first-contact negotiation of the exchange, watchdog, timers — through the stand-in transport
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05f; mkdir -p $O; : > $O/summary.txt
for t in "tests/test_gpu_comm.py -k first_contact" "tests/test_gpu_bench_launch.py" "tests/test_gpu_comm.py -k 'not first_contact'" "tests/test_gpu_sharded_svd.py"; do
  tag=$(echo "$t" | tr ' /' '__' | tr -d "'")
  timeout 1500 bash -c "python -m pytest $t -m gpu -q -x" > $O/$tag.log 2>&1
  echo "$t rc=$? $(grep -E 'passed|failed|error' $O/$tag.log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -40
# the driver's 8-rank launch line at full size through the stand-in (one GPU: 8 shards of 125 000 variants)
MOCK=$(python -c "import sys; sys.path.insert(0,'tests/native'); import build_native; print(build_native.build_mock_rccl())")
BSN_RCCL_LIBRARY=$MOCK timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 3 --warmup 1 --no-uv > $O/bench_8ranks_standin.json 2> $O/bench_8ranks_standin.err
echo "8 ranks rc=$?" | tee -a $O/summary.txt
python - <<'P' | tee -a $O/summary.txt
import json
try:
    d = json.load(open('gpurun_out/r05f/bench_8ranks_standin.json'))
    print('8 ranks (stand-in, one GPU): %.1f ms' % d['ms_per_step'], 'niter', d['niter'], 'converged', d['converged'], 'sigma1', d['sigma'][0])
    print(json.dumps(d['exchange'])[:3000])
except Exception as e:
    print('no bench line', e)
P
tail -5 $O/bench_8ranks_standin.err | cut -c1-300
