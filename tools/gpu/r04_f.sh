#!/bin/bash
# round 4, trip F: the sample-major copy + k_prodT: parity, build time, kernel time, default bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_smaj.py tests/test_gpu_ld.py -x -q 2>&1 | grep -v "^RCCL" | tail -15 | tee $O/tests.txt
BSN_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.load(open('$O/bench.json')); print(round(d['ms_per_step'],2),'ms passes', d['passes_per_solve'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}); print(d['image_layout']); print(d['roofline']['kernels_launched'].get('prod')); print(d['sigma'][:3])
P
BSN_NO_SMAJ=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-wide > $O/bench_nosmaj.json 2> $O/bench_nosmaj.err
python - <<P
import json
d=json.load(open('$O/bench_nosmaj.json')); print('no smaj:', round(d['ms_per_step'],2),'ms', {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}, d['sigma'][:3])
P
cat > /tmp/smaj_t.py <<'P'
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()
gb = ba.bed.synthetic(400000, 1000000)
L.bsn_device_sync()
t0 = time.perf_counter(); ok = gb.sample_major(); L.bsn_device_sync(); t1 = time.perf_counter()
print("sample-major copy of 400000 x 1000000 built:", ok, "%.1f ms" % ((t1 - t0) * 1e3))
P
python /tmp/smaj_t.py 2>&1 | grep built | tee $O/build_time.txt
for ky in 13 17 21; do BSN_KY=$ky BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --no-wide 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BSN_KY=$ky', round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})"; done | tee $O/ky.txt
