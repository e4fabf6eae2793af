#!/bin/bash
# round 4: the library's roctx ranges in a rocprofv3 marker trace of two solves
cd /tmp; export TMPDIR=/tmp
BSN_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/pm -o mk -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --no-wide > /dev/null 2> /tmp/pm.err
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04roctx; mkdir -p $O
ls /tmp/pm/* | head; f=$(find /tmp/pm -name '*marker_api_trace.csv' | head -1); echo "marker file: $f"
python3 - "$f" > $O/roctx_ranges.txt <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --marker-trace of `BSN_ROCTX=1 python bench.py --steps 2 --warmup 1`: ranges pushed by libbigsnpr_hip (host-side spans, ms)")
print("# columns of the trace:", list(rows[0].keys()) if rows else None)
agg = collections.OrderedDict()
for r in rows:
    name = r.get("Function") or r.get("Name") or r.get("Marker_Name") or str(r)
    try:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    except Exception:
        continue
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += d
for k, (n, t) in agg.items():
    print("%-60s %4d ranges  %9.2f ms total  %8.3f ms avg" % (k[:60], n, t, t / n))
P
cat $O/roctx_ranges.txt | head -30
