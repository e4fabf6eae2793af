#!/bin/bash
# round 6, trip K: the whole GPU suite on the current build + the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06k; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.txt 2>&1
tail -25 $O/pytest_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06k/bench_default.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
print('accuracy', d.get('accuracy',{}).get('u_leading_half'), d.get('accuracy',{}).get('v_leading_half'))
c=d.get('cold',{})
print('cold full', {k:c.get('synthetic_full_size',{}).get(k) for k in ('first_solve_ms','warm_solve_ms','first_minus_warm_ms','handle_s','library_and_runtime_s','solve_ms')})
print('cold bed', {k:c.get('real_bed_file',{}).get(k) for k in ('open_ms','first_solve_ms','warm_solve_ms','first_minus_warm_ms','file')})
print('cpu', d.get('cpu_baseline',{}).get('value'), 'ingest', d.get('ingest',{}).get('GBps'))
PY
timeout 600 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_1m.txt 2>&1
tail -6 $O/autosvd_1m.txt | cut -c1-400
