#!/bin/bash
# round 2, trip A: parity after the image recode + fused statistics, first timings, ablations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verbose 1 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
for blk in 12 16; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --block $blk --slices 2 > $O/bench_b$blk.json 2> $O/bench_b$blk.err
done
timeout 300 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8,16 --slices 2 --reps 5 > $O/probe.log 2>&1
for t in 11 12 13 17 19 61 62 63 64; do
  BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so BSN_TUNE=$t timeout 300 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8 --slices 2 --reps 5 > $O/probe_tune$t.log 2>&1
done
grep -h kernel $O/probe*.log | head -60
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; find $O/prof -type f ! -name "*stats*" -delete 2>/dev/null
head -12 $O/kernel_stats.csv
