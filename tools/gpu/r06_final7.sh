#!/bin/bash
# round 6, second session, last trip: the GPU suite in ONE pytest process on the final build, smoke(), the C5 line and the
# per-counter record of the shipped LD kernel (k_pair_stats_f4<true, true, false, 2>), the driver's bench line, C2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06final7; mkdir -p $O; : > $O/summary.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/suite_one_process.log 2>&1
echo "pytest tests/ -x -q -m gpu (one process): rc=$? $(grep -E 'passed|failed' $O/suite_one_process.log | tail -1) wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
grep -n "FAILED\|^E " $O/suite_one_process.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> /dev/null
timeout 300 python bench.py --workload matvec --steps 20 --warmup 3 > $O/c2_matvec.json 2> /dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
for f in ld_bench c2_matvec bench_default; do python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), 'ms', d['roofline'].get('frac'), d.get('auto_svd',{}).get('second_call_s'))" | tee -a $O/summary.txt; done
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc -o g$i -- python $R/tools/probe_ld_complete.py > $O/pmc_g$i.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc "k_pair|k_quad" > $O/ld_pmc_summary.txt 2>&1
rm -rf $O/pmc
head -24 $O/ld_pmc_summary.txt | cut -c1-300
