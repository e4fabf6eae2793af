#!/bin/bash
# round 5, trip I: out-of-core bed_randomSVD (tests + the 8-GB file against a 2-GB budget), R shim, compaction, whole suite
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05i; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_out_of_core.py tests/test_gpu_r_shim.py tests/test_gpu_edge_cases.py tests/test_gpu_svd.py tests/test_gpu_fused_scaling.py; do
  timeout 900 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -30
timeout 900 python tools/probe_ooc.py > $O/out_of_core.json 2> $O/out_of_core.err
python -c "
import json; d=json.load(open('$O/out_of_core.json')); print({k: (v if not isinstance(v, dict) else {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b, dict)}) for k,v in d.items()})" | tee -a $O/summary.txt
tail -3 $O/out_of_core.err
