#!/bin/bash
# round 6, trip T: all chromosomes of a clumping in one device call — tests, then snp_autoSVD at 400K x 1M (and with the
# per-chromosome loop for comparison)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06t; mkdir -p $O
for t in ld autosvd edge_cases fbm out_of_core random_shapes; do
  timeout 1200 python -m pytest tests/test_gpu_$t.py -x -q -m gpu 2>&1 | tail -12 > $O/pytest_$t.txt
  echo "$t: $(tail -1 $O/pytest_$t.txt)"
  grep -q "failed\|error" $O/pytest_$t.txt && cat $O/pytest_$t.txt
done
timeout 900 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_1m.txt 2>&1
grep "call\|inside" $O/autosvd_1m.txt | cut -c1-330
BSN_CLUMP_PER_CHR=1 timeout 900 python tools/probe_autosvd.py --m 1000000 > $O/autosvd_1m_per_chr.txt 2>&1
grep "call\|inside" $O/autosvd_1m_per_chr.txt | cut -c1-330
