#!/bin/bash
# round 6, trip E: the copy beside the first solve + the cold record, the slab rule on the shards, the changed tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_smaj.py tests/test_gpu_svd.py tests/test_gpu_comm.py tests/test_gpu_tiled.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-wide > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06e/bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'cold', json.dumps(d.get('cold'))[:1500])
PY
for N in 2 4 8; do
  timeout 300 python bench.py --steps 8 --warmup 2 --shard-of $N --force-dist --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/shard_$N.json 2> $O/shard_$N.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r06e/shard_$N.json').read().strip().splitlines()[-1])
print($N, d['ms_per_step'], d['niter'], {k:round(v['avg_ms'],3) for k,v in d['roofline']['other'].items()}, d['exchange'].get('ms_per_solve'))
PY
done
