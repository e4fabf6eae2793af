#!/usr/bin/env python3
"""Prints VGPR/AGPR/SGPR/LDS/scratch/occupancy per kernel of a .hip file (gfx950)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "bigsnpr_amd/csrc"),
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
keys = ["VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPRs Spill"]
print("%-52s %s" % ("kernel", " ".join("%9s" % k.split()[0][:9] for k in keys)))
for n, r in rows.items():
    print("%-52s %s" % (n[:52], " ".join("%9s" % r.get(k, "?") for k in keys)))
