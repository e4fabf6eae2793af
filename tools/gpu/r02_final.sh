#!/bin/bash
# round 2, second half: measurements of the secondary rows (profiles/r02_rows_probe.txt and friends)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python tools/probe_rows.py 2>&1 | grep -v "$F" > $O/rows.txt
timeout 300 python tools/probe_calls.py 2>&1 | grep entry > $O/calls_c2.txt
{ timeout 300 python tools/probe_tcross.py --n 2000 --m 100000 --reps 2; timeout 300 python tools/probe_tcross.py --n 8192 --m 32768;
  timeout 600 python tools/probe_tcross.py --n 16384 --m 65536 --reps 1; } 2>&1 | grep entry > $O/tcross.txt
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python tools/probe_tcross.py --n 16384 --m 65536 --reps 1 > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -4 "$f" > $O/tcross_kernel_stats.csv
timeout 900 python tools/probe_clump_lazy.py 2>&1 | grep bed_clumping > $O/clump_lazy.txt
timeout 600 python tools/probe_autosvd.py 2>&1 | grep "total" > $O/autosvd.txt
{ timeout 300 python tools/probe_matvec.py --dosage --n 50000 --m 200000 --nvecs 1,8 --slices 2 --reps 8; } 2>&1 | grep "kernel" > $O/byte_kernels.txt
{ timeout 300 python tools/probe_matvec.py --n 400000 --m 250000 --nvecs 8 --slices 2 --reps 8 --subset 0.5; } 2>&1 | grep "kernel\|subset" > $O/subset_kernels.txt
for b in 8 16; do timeout 600 python bench.py --block $b --steps 4 --warmup 1 --no-cpu-baseline --no-ingest 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('block $b:', round(d['ms_per_step'],1),'ms per solve, passes', round(d['passes_per_solve'],3), 'block steps', d['niter'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})"; done > $O/block_sweep.txt
tail -n +1 $O/*.txt | head -120
