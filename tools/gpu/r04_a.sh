#!/bin/bash
# round 4, trip A (diagnostics): power / clock under the streaming kernels, grid-fill staircase, the transposed shape (what a
# sample-major k_prod would cost), ablations + sched_group_barrier / s_setprio variants of the two-block kernels, K-split sweep
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
rocm-smi --showpower --showclocks > $O/smi_idle.txt 2>&1
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $O/hwmon_ls.txt 2>&1
echo "== power" ; 
timeout 200 python tools/probe_power.py --n 400000 --m 500000 --seconds 4 2>&1 | grep -v "^RCCL\|^HIP ver" | tee $O/power_normal.txt
timeout 200 python tools/probe_power.py --n 400000 --m 500000 --seconds 4 --xkind ones 2>&1 | grep "kernel" | tee $O/power_ones.txt
echo "== staircase"
for m in 917504 1000000 1048576; do
  echo "m=$m"; timeout 200 python tools/probe_matvec.py --n 400000 --m $m --nvecs 16 --slices 2 --reps 6 2>&1 | grep '"kernel"' | tee -a $O/staircase.txt
done
echo "== transposed shape"
timeout 200 python tools/probe_matvec.py --n 1000000 --m 400000 --nvecs 16,8 --slices 2 --reps 6 2>&1 | grep '"kernel"' | tee $O/transposed.txt
echo "== ablations (50 GB shard, 16 vectors x 2 slices)"
export BSN_LIB_PATH=$PWD/bigsnpr_amd/libbigsnpr_hip_abl.so
for t in 0 111 112 113 117 119 121 122 123 161 162 163 164 171 172 173 0; do
  echo "BSN_TUNE=$t"; BSN_TUNE=$t timeout 120 python tools/probe_matvec.py --n 400000 --m 500000 --nvecs 16 --slices 2 --reps 6 2>&1 | grep '"cprod"\|"prod"' | tee -a $O/ablation.txt
done
echo "== K split of k_prod<2> at 1M variants"
for ky in 11 13 17 26; do
  echo "BSN_KY=$ky"; BSN_KY=$ky timeout 200 python tools/probe_matvec.py --n 400000 --m 1000000 --nvecs 16 --slices 2 --reps 6 2>&1 | grep '"prod"' | tee -a $O/ky.txt
done
