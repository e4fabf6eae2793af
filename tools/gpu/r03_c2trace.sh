#!/bin/bash
# device-side timeline of one-shot calls at C2: kernels and copies of the last prodVec / cprodVec call
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c2t; mkdir -p $O
cat > /tmp/c2one.py <<'P'
import sys, time, numpy as np
import os; sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()
n, m = 50000, 200000
gb = ba.bed.synthetic(n, m, seed=9)
sc = ba.bed_scaleBinom(gb)
rng = np.random.default_rng(0)
x, y = rng.normal(size=m), rng.normal(size=n)
for _ in range(6):
    ba.bed_prodVec(gb, x, center=sc["center"], scale=sc["scale"]); ba.bed_cprodVec(gb, y, center=sc["center"], scale=sc["scale"])
L.bsn_device_sync()
for fn, nm in ((lambda: ba.bed_prodVec(gb, x, center=sc["center"], scale=sc["scale"]), "prodVec"), (lambda: ba.bed_cprodVec(gb, y, center=sc["center"], scale=sc["scale"]), "cprodVec")):
    time.sleep(0.01)
    t0 = time.perf_counter_ns(); fn(); t1 = time.perf_counter_ns()
    print("HOST %s wall %.1f us" % (nm, (t1 - t0) / 1e3))
P
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/c2tr -o t -- python /tmp/c2one.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
cd "$GRAFT_REPO_ROOT"
python - <<'P'
import csv, glob, re
ev = []
for f in glob.glob('/tmp/c2tr/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void bsn::", "").replace("bsn::", "")[:60]))
for f in glob.glob('/tmp/c2tr/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?")))))
ev.sort()
# the last two calls: split on gaps > 5 ms
groups, cur = [], []
for e in ev:
    if cur and e[0] - cur[-1][1] > 5_000_000:
        groups.append(cur); cur = []
    cur.append(e)
groups.append(cur)
out = open('gpurun_out/r03c2t/timeline.txt', 'w')
for g in groups[-2:]:
    t0 = g[0][0]
    out.write("---- call: span %.1f us\n" % ((g[-1][1] - t0) / 1e3))
    for st, en, nm in g:
        out.write("%8.1f .. %8.1f  (%7.1f us)  %s\n" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, nm))
out.close()
print(open('gpurun_out/r03c2t/timeline.txt').read())
P
