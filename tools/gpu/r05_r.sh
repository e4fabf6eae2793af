#!/bin/bash
# the 8-wave LD kernel (both operands through LDS, BSN_LD_F4W=1) against the default, same box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05r; mkdir -p $O; : > $O/summary.txt
for v in "" 1; do
  tag=${v:+f4w}; tag=${tag:-default}
  for f in tests/test_gpu_ld.py tests/test_gpu_fbm.py; do
    env ${v:+BSN_LD_F4W=1} timeout 900 python -m pytest $f -m gpu -q -x > $O/${tag}_$(basename $f .py).log 2>&1
    echo "$tag $f rc=$? $(grep -E 'passed|failed|error' $O/${tag}_$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
  done
  env ${v:+BSN_LD_F4W=1} timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/${tag}_ld_bench.json 2> $O/${tag}_ld_bench.err
  python - "$O/${tag}_ld_bench.json" "$tag" <<'P' | tee -a $O/summary.txt
import json, sys
for line in open(sys.argv[1]):
    try: r = json.loads(line)
    except Exception: continue
    print(sys.argv[2], r.get("metric"), r.get("ms_per_step"), json.dumps(r.get("roofline"))[:300])
P
done
grep -n "FAILED\|^E " $O/*.log | head -20
