#!/bin/bash
# round 4: Infinity-Cache residency micro-benchmark (tools/ubench/l3.hip) + the clock while it runs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04l3; mkdir -p $O
(while true; do rocm-smi --showclocks 2>/dev/null | grep "sclk" | sed 's/.*(\([0-9]*Mhz\)).*/\1/' | tr '\n' ' '; date +%s.%N | cut -c1-14; sleep 0.25; done) > $O/sclk.txt &
SMI=$!
./tools/ubench/l3 ${L3_GB:-100} | tee $O/l3.txt
kill $SMI
awk '{print $1}' $O/sclk.txt | sort | uniq -c | sort -rn | head -8 > $O/sclk_hist.txt; cat $O/sclk_hist.txt
