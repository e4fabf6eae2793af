#!/bin/bash
# round 6, trip W: the whole GPU suite after the LD / robust / counts changes, file by file; the first call of the outlier step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06w; mkdir -p $O
BSN_ALLOC_TRACE=1 timeout 600 python tools/probe_first_outlier.py > $O/first_outlier.txt 2>&1
grep -v "^\[alloc\] .* 0\.0[0-9]* ms" $O/first_outlier.txt | tail -40
: > $O/summary.txt
for f in tests/test_gpu_*.py; do
  timeout 1500 python -m pytest $f -x -q -m gpu > $O/$(basename $f .py).txt 2>&1
  echo "$f rc=$? $(tail -1 $O/$(basename $f .py).txt)" | tee -a $O/summary.txt
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.txt)" | tee -a $O/summary.txt
