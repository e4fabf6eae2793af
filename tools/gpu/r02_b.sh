#!/bin/bash
# round 2, trip B: full parity suite, workgroup-shape variants of the two streaming kernels
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
P="python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8 --slices 2 --reps 8"
timeout 300 $P > $O/probe_0.log 2>&1
for t in 21 22 23 24 25 26 27 71 72 73; do
  BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so BSN_TUNE=$t timeout 300 $P > $O/probe_tune$t.log 2>&1
done
timeout 300 $P > $O/probe_0b.log 2>&1
for f in $O/probe_*.log; do echo $f; grep -h '"cprod"\|"prod"' $f; done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verbose 1 > $O/bench.json 2> $O/bench.err; tail -c 1200 $O/bench.json
