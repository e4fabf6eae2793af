#!/bin/bash
# round 6, trip U: k_pair_xy_f4 at three waves per SIMD without register-file shuffles, mask skipped when all samples are selected
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06u; mkdir -p $O
for t in ld complete_data fbm random_shapes autosvd; do
  timeout 1200 python -m pytest tests/test_gpu_$t.py -x -q -m gpu 2>&1 | tail -12 > $O/pytest_$t.txt
  echo "$t: $(tail -1 $O/pytest_$t.txt)"
  grep -q "failed\|error" $O/pytest_$t.txt && cat $O/pytest_$t.txt
done
timeout 600 python tools/probe_ld_complete.py > $O/ld_complete.txt 2>&1; cat $O/ld_complete.txt
timeout 900 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_1m.txt 2>&1
grep "call\|inside" $O/autosvd_1m.txt | cut -c1-330
