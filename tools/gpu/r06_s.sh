#!/bin/bash
# round 6, trip S: the outlier step on the device + the counts a handle remembers + order(S) of all chromosomes in one device
# call: the tests that cover them, then snp_autoSVD at 400K x 1M
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06s; mkdir -p $O
for t in autosvd edge_cases ld fbm out_of_core complete_data sct pcadapt; do
  timeout 900 python -m pytest tests/test_gpu_$t.py -x -q -m gpu 2>&1 | tail -12 > $O/pytest_$t.txt
  echo "$t: $(tail -1 $O/pytest_$t.txt)"
  grep -q "failed\|error" $O/pytest_$t.txt && cat $O/pytest_$t.txt
done
timeout 900 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_1m.txt 2>&1
grep "call\|inside" $O/autosvd_1m.txt | cut -c1-330
