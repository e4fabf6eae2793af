#!/bin/bash
# the whole GPU suite, file by file (a crash in one file does not hide the others)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04tests; mkdir -p $O; : > $O/summary.txt
for f in tests/test_gpu_*.py tests/test_prs_pipeline_golden.py; do
  timeout 1500 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
for f in $O/*.log; do if grep -q "Fatal Python error\|core dumped\|Segmentation\|Aborted" $f; then echo "== crash in $f"; grep -n "Fatal Python error" -A12 $f | head -20; fi; done | tee $O/crash.txt
grep -n "FAILED\|^E " $O/*.log | head -30
