#!/bin/bash
# round 4, trip D: parity of the pipelined k_cprod<2> in every variant + readbina, the default bench line, one-block pipeline timing
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r_shim.py tests/test_gpu_plink_io.py tests/test_gpu_matvec.py tests/test_gpu_svd.py tests/test_gpu_fused_scaling.py tests/test_gpu_complete_data.py tests/test_gpu_tiled.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.load(open('$O/bench.json')); print(round(d['ms_per_step'],2),'ms passes', d['passes_per_solve'], 'conv', d['converged'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()}); print(d['roofline']['traffic'], d['roofline']['traffic_source']); print(d['roofline']['kernels_launched']); print(d.get('fp64_equivalent'))
P
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for t in 0 143 0 143; do
  echo "BSN_TUNE=$t"; BSN_TUNE=$t timeout 120 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8 --slices 2 --reps 8 2>&1 | grep '"cprod"' | tee -a $O/nb1.txt
done
for t in 0 114; do
  echo "BSN_TUNE=$t"; BSN_TUNE=$t timeout 120 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 16 --slices 2 --reps 8 2>&1 | grep '"cprod"' | tee -a $O/nb2.txt
done
