#!/bin/bash
# round 6, trip X: bed_autoSVD at 400K x 1M (1 % missing values: the clumping band on the six-product FP4 kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06x; mkdir -p $O
timeout 1200 python tools/probe_autosvd.py --m 1000000 --bed > $O/bed_autosvd_1m.txt 2>&1
grep "call\|inside\|Error\|error" $O/bed_autosvd_1m.txt | cut -c1-360; tail -3 $O/bed_autosvd_1m.txt | cut -c1-300
