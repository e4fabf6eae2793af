#!/bin/bash
# round 6, trip M: the whole GPU suite + the driver's bench line + shard lines + autoSVD at 1M on the build of the second batch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.txt 2>&1
tail -12 $O/pytest_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06m/bench_default.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
print('accuracy', d.get('accuracy',{}).get('u_leading_half'), d.get('accuracy',{}).get('v_leading_half'))
c=d.get('cold',{})
print('cold full', {k:c.get('synthetic_full_size',{}).get(k) for k in ('first_solve_ms','warm_solve_ms','first_minus_warm_ms','solve_ms')})
PY
for N in 2 4 8; do
  timeout 300 python bench.py --steps 8 --warmup 2 --shard-of $N --force-dist --no-cpu-baseline --no-ingest --no-wide --no-accuracy > $O/shard_$N.json 2> $O/shard_$N.err
done
timeout 600 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_1m.txt 2>&1
tail -4 $O/autosvd_1m.txt | cut -c1-400
