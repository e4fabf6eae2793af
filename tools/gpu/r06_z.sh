#!/bin/bash
# round 6, trip Z: the random-shape parity suite (products, statistics, LD, clumping, SVD against the oracle) on three more draws
# after the LD / counts / clumping changes; the kernel trace of the bench line without the auto_svd record (a clean timeline)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06z; mkdir -p $O
for off in 4 5 6; do
  BSN_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random_shapes.py -x -q -m gpu 2>&1 | tail -3 > $O/random_shapes_offset_$off.txt
  echo "offset $off: $(tail -1 $O/random_shapes_offset_$off.txt)"
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-wide --no-accuracy --no-cold --no-autosvd > /dev/null 2> /tmp/pk.err
cd "$GRAFT_REPO_ROOT"
cp $(find /tmp/pk -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
python tools/trace_gaps.py $(find /tmp/pk -name '*kernel_trace.csv' | head -1) > $O/bench_solve_timeline.txt; tail -3 $O/bench_solve_timeline.txt | cut -c1-250
head -4 $O/bench_kernel_stats.csv | cut -c1-50,140-250
