#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
df -h /tmp | tail -1; free -g | head -2; nproc
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep "^\[bench" $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r02e/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","passes_per_solve","niter")}); print(d["roofline"]["frac"], d["roofline"]["other"]); print(d.get("cpu_baseline")); print(d.get("ingest"))
PY
timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_matvec.py -q 2>&1 | tail -3
