#!/bin/bash
# round 2: the measurements behind profiles/r02_* (run from the repo root on the GPU box)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02prof; mkdir -p $O
R=$GRAFT_REPO_ROOT
# 1. the driver's command line
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
# 2. kernel trace of the same workload
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > $R/$O/kt.log 2>&1)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
T=$(find $O/kt -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T > $O/bench_solve_timeline.txt; rm -rf $O/kt
# 3. HBM traffic (PMC pass on its own: kernel-trace only) and clock
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc -o g -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ingest > $R/$O/pmc.log 2>&1)
python tools/pmc_summary.py $O/pmc "k_prod|k_cprod" > $O/pmc_fetch.txt 2>&1; rm -rf $O/pmc
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $R/$O/pmc2 -o g -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ingest > $R/$O/pmc2.log 2>&1)
python tools/pmc_summary.py $O/pmc2 "k_prod|k_cprod" > $O/pmc_sq.txt 2>&1; rm -rf $O/pmc2
# 4. LD (config C5)
timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld.json 2> $O/ld.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ldkt -- python $R/bench.py --workload ld --steps 2 --warmup 1 > $R/$O/ldkt.log 2>&1)
find $O/ldkt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/ld_kernel_stats.csv; rm -rf $O/ldkt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/$O/ldpmc -o g -- python $R/bench.py --workload ld --steps 1 --warmup 0 > $R/$O/ldpmc.log 2>&1)
python tools/pmc_summary.py $O/ldpmc "k_pair|k_ld|k_cor|k_band" > $O/ld_pmc.txt 2>&1; rm -rf $O/ldpmc
# 5. through the RCCL communicator with one rank
timeout 600 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline --no-ingest > $O/bench_comm1.json 2> $O/bench_comm1.err
# 6. ablations of the two streaming kernels on a 50 GB shard
P="python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 8 --slices 2 --reps 8"
$P > $O/probe_product.log 2>&1
for t in 11 12 13 15 17 19 61 62 63 64; do BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so BSN_TUNE=$t $P > $O/probe_tune$t.log 2>&1; done
for f in $O/probe_*.log; do echo "$f $(grep -h '"cprod"\|"prod"' $f | tr '\n' ' ')"; done > $O/ablation_raw.txt
ls -la $O; head -c 1500 $O/bench_default.json; echo; cat $O/pmc_fetch.txt; head -8 $O/ld_kernel_stats.csv
