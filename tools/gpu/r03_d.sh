#!/bin/bash
# round 3: kernel traces of the 125K shard solve and of the full-size solve (csv kept for analysis)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p125 -o st -- python $GRAFT_REPO_ROOT/bench.py --variants 125000 --steps 2 --warmup 1 --no-cpu-baseline --no-ingest $BENCH_EXTRA > /dev/null 2> /tmp/p125.err
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/p125 -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f > $O/timeline_125k.txt; tail -3 $O/timeline_125k.txt
python - $f > $O/trace_125k_last_solve.csv <<'P'
import csv,sys,re
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),re.sub(r"\(.*","",r["Kernel_Name"]).replace("void bsn::","").replace("bsn::","")[:60], r.get("Grid_Size_X","") , r.get("Grid_Size_Y","")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx=[i for i,r in enumerate(rows) if r[2].startswith("k_random")]
last=rows[idx[-1]:]
t0=last[0][0]
for st,en,nm,gx,gy in last: print("%.1f,%.1f,%s,%s,%s" % ((st-t0)/1e3,(en-st)/1e3,nm,gx,gy))
P
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pfull -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest $BENCH_EXTRA > /dev/null 2> /tmp/pfull.err
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/pfull -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f > $O/timeline_full.txt; tail -3 $O/timeline_full.txt
