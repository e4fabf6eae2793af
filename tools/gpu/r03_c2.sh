#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_edge_cases.py tests/test_gpu_comm.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 600 python bench.py --workload matvec --steps 50 --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python - <<'P'
import json; d=json.load(open('gpurun_out/r03c2/c2.json')); print('C2 matvec: %.3f ms per call, whole-call frac %.3f' % (d['ms_per_call'], d['roofline']['frac']))
P
BSN_COPY_THREADS=1 timeout 600 python bench.py --workload matvec --steps 50 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('   one copy thread: %.3f ms per call' % d['ms_per_call'])"
