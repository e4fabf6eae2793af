#!/bin/bash
# round 6, trip R: the outlier step of snp_autoSVD on the device end to end (dist_ogk fused, rollmean, sort, medcouple window):
# its tests, then snp_autoSVD at 400K x 1M
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_autosvd.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_autosvd.txt
cat $O/pytest_autosvd.txt
timeout 900 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_1m.txt 2>&1
grep -v "^Discarding\|^$\|Iteration\|Computing\|outlier\|Converged\|Phase" $O/autosvd_1m.txt | tail -8
