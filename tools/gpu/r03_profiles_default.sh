#!/bin/bash
# the default (16-vector) configuration again after the tiled copy was limited to one-block solves
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03prof3; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > $R/$O/kt.log 2>&1)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
T=$(find $O/kt -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T > $O/bench_solve_timeline.txt; rm -rf $O/kt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc -o g -- python $R/bench.py --block 16 --steps 1 --warmup 0 --no-cpu-baseline --no-ingest > $R/$O/pmc.log 2>&1)
python tools/pmc_summary.py $O/pmc "k_prod|k_cprod" > $O/pmc_fetch_block16.txt 2>&1; rm -rf $O/pmc
head -c 600 $O/bench_default.json; echo; grep "==" $O/pmc_fetch_block16.txt
