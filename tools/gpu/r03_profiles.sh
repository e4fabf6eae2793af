#!/bin/bash
# round 3: the measurements behind profiles/r03_* (run from the repo root on the GPU box)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03prof; mkdir -p $O
R=$GRAFT_REPO_ROOT
# 1. the driver's command line (library default: 16 vectors per pass at k = 20) and the 8-vector configuration
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --block 8 --steps 10 --warmup 3 --no-cpu-baseline --no-ingest > $O/bench_block8.json 2> $O/bench_block8.err
# 2. kernel trace of the same workload: stats + per-solve timeline; and of the 125 000-variant shard (the per-GPU work of an 8-GPU run)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > $R/$O/kt.log 2>&1)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
T=$(find $O/kt -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T > $O/bench_solve_timeline.txt; rm -rf $O/kt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt8 -- python $R/bench.py --block 8 --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > $R/$O/kt8.log 2>&1)
find $O/kt8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_block8_kernel_stats.csv; rm -rf $O/kt8
timeout 300 python bench.py --variants 125000 --steps 10 --warmup 3 --no-uv --no-cpu-baseline --no-ingest > $O/shard125k.json 2> $O/shard125k.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kts -- python $R/bench.py --variants 125000 --steps 3 --warmup 1 --no-uv --no-cpu-baseline --no-ingest > $R/$O/kts.log 2>&1)
T=$(find $O/kts -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T > $O/shard125k_timeline.txt; rm -rf $O/kts
# 3. HBM traffic (PMC pass on its own: kernel-trace only) and clock; SQ counters — both kernel families
for b in 16 8; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc -o g -- python $R/bench.py --block $b --steps 1 --warmup 0 --no-cpu-baseline --no-ingest > $R/$O/pmc.log 2>&1)
python tools/pmc_summary.py $O/pmc "k_prod|k_cprod" > $O/pmc_fetch_block$b.txt 2>&1; rm -rf $O/pmc
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $R/$O/pmc2 -o g -- python $R/bench.py --block $b --steps 1 --warmup 0 --no-cpu-baseline --no-ingest > $R/$O/pmc2.log 2>&1)
python tools/pmc_summary.py $O/pmc2 "k_prod|k_cprod" > $O/pmc_sq_block$b.txt 2>&1; rm -rf $O/pmc2
done
# 4. LD (config C5)
timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld.json 2> $O/ld.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ldkt -- python $R/bench.py --workload ld --steps 2 --warmup 1 > $R/$O/ldkt.log 2>&1)
find $O/ldkt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/ld_kernel_stats.csv; rm -rf $O/ldkt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $R/$O/ldpmc -o g -- python $R/bench.py --workload ld --steps 1 --warmup 0 > $R/$O/ldpmc.log 2>&1)
python tools/pmc_summary.py $O/ldpmc "k_pair|k_ld|k_cor|k_band" > $O/ld_pmc.txt 2>&1; rm -rf $O/ldpmc
# 5. config C2: one-shot calls
timeout 600 python bench.py --workload matvec --steps 50 > $O/c2_matvec.json 2> $O/c2_matvec.err
# 6. through the RCCL communicator with one rank
timeout 600 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline --no-ingest > $O/bench_comm1.json 2> $O/bench_comm1.err
# 7. the collectives of a sharded solve (stand-in transport, two ranks on this GPU)
bash tools/gpu/r03_tr.sh > $O/collectives_trace_raw.txt 2>&1
ls -la $O; head -c 1800 $O/bench_default.json; echo; cat $O/pmc_fetch_block16.txt | head -30
