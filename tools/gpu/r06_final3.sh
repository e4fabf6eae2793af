#!/bin/bash
# round 6, second session: the whole GPU suite at HEAD (file by file, as tools/gpu/r06_final2.sh), smoke, the driver's own
# bench invocation with its wall time, six further draws of the random-shape parity suite, and the ordered kernel list of
# one traced solve (what the non-streaming 15 ms of a solve are)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r06final3; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_*.py tests/test_prs_pipeline_golden.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
t0=$(date +%s.%N)
timeout 900 python bench.py > $O/bench_driver_style.json 2> $O/bench_driver_style.err
t1=$(date +%s.%N)
python - <<P | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_driver_style.json').read().strip().splitlines()[-1]); r=d['roofline']
print('python bench.py (no flags): wall %.1f s' % ($t1 - $t0), 'steps', d['steps'], 'warmup', d['warmup'], '%.2f ms' % d['ms_per_step'], 'value %.3e' % d['value'],
      'roofline', r['bound'], round(r['frac'],3), 'hbm', round(r['hbm']['frac'],3))
print('accuracy', {k: d['accuracy'].get(k) for k in ('u_leading_half','v_leading_half','leading_half_within_tolerance')})
c=d.get('cold',{})
print('cold full', {k:c.get('synthetic_full_size',{}).get(k) for k in ('first_solve_ms','warm_solve_ms','first_minus_warm_ms')})
P
for off in 7 8 9 10 11 12; do
  BSN_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random_shapes.py -x -q -m gpu 2>&1 | tail -3 > $O/random_shapes_offset_$off.txt
  echo "random shapes, offset $off: $(tail -1 $O/random_shapes_offset_$off.txt)" | tee -a $O/summary.txt
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --no-wide --no-accuracy --no-cold --no-autosvd > /dev/null 2> /tmp/pk.err
cd "$GRAFT_REPO_ROOT"
python - <<'P' > $O/one_solve_kernel_list.txt
import csv, glob, re
f = glob.glob('/tmp/pk/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void bsn::", "")) for r in csv.DictReader(open(f)))
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_random")]
s = rows[starts[-1]:]
prev = s[0][0]
for st, en, name in s:
    print("%9.3f ms  gap %7.3f  dur %8.3f  %s" % ((st - s[0][0]) / 1e6, (st - prev) / 1e6, (en - st) / 1e6, name[:110]))
    prev = en
P
tail -2 $O/one_solve_kernel_list.txt
