#!/usr/bin/env python3
"""Out-of-core handle at size: writes a synthetic .bed of --gb gigabytes (page cache warm), opens it with
BSN_IMAGE_BUDGET = --budget-gb so that it is walked in slabs, and times bed_counts / bed_cprodVec / bed_prodVec against
the resident handle of the same file (run on the GPU box)."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400000)
ap.add_argument("--gb", type=float, default=8.0)
ap.add_argument("--budget-gb", type=float, default=2.0)
a = ap.parse_args()
nb = (a.n + 3) // 4
m = int(a.gb * 1e9 / nb)
d = tempfile.mkdtemp(prefix="bsn_ooc_")
path = os.path.join(d, "big.bed")
gb = ba.bed.synthetic(a.n, m, seed=7)
payload = gb.download()
with open(path, "wb") as f:
    f.write(bytes([0x6C, 0x1B, 0x01]))
    f.write(payload.tobytes())
open(path[:-4] + ".fam", "w").write("\n".join("f i 0 0 0 -9" for _ in range(a.n)) + "\n")
open(path[:-4] + ".bim", "w").write("\n".join("1 s 0 %d A T" % j for j in range(m)) + "\n")
del payload
res = ba.bed(path)
os.environ["BSN_IMAGE_BUDGET"] = str(int(a.budget_gb * 1e9))
ooc = ba.bed(path)
del os.environ["BSN_IMAGE_BUDGET"]
assert ooc.streamed and not res.streamed
rng = np.random.default_rng(0)
y, x = rng.normal(size=a.n), rng.normal(size=m)
sc = ba.bed_scaleBinom(res)
sa = np.where(sc["scale"] > 0, sc["scale"], 1.0)
out = {"n": a.n, "m": m, "file_GB": nb * m / 1e9, "budget_GB": a.budget_gb}
for name, fn in (("counts", lambda h: ba.bed_counts(h)), ("cprodVec", lambda h: ba.bed_cprodVec(h, y, None, None, sc["center"], sa)),
                 ("prodVec", lambda h: ba.bed_prodVec(h, x, None, None, sc["center"], sa))):
    r = {}
    for tag, h in (("resident", res), ("out_of_core", ooc)):
        fn(h)
        t0 = time.perf_counter(); v = fn(h); t = time.perf_counter() - t0
        r[tag] = {"s": t, "GBps": nb * m / t / 1e9}
        r[tag + "_val"] = v
    same = np.array_equal(r["resident_val"], r["out_of_core_val"])
    close = float(np.max(np.abs(np.asarray(r["resident_val"], dtype=float) - np.asarray(r["out_of_core_val"], dtype=float))) /
                  max(1e-300, float(np.max(np.abs(np.asarray(r["resident_val"], dtype=float))))))
    out[name] = {"resident": r["resident"], "out_of_core": r["out_of_core"], "identical": bool(same), "max_rel_diff": close}
# (round 5) bed_randomSVD on the out-of-core handle: every pass walks the file over PCIe
if "--svd" in sys.argv or True:
    k = 10
    t0 = time.perf_counter(); r0 = ba.bed_randomSVD(res, k=k); t_res = time.perf_counter() - t0
    t0 = time.perf_counter(); r1 = ba.bed_randomSVD(ooc, k=k); t_ooc = time.perf_counter() - t0
    passes = r1["nops"]
    out["bed_randomSVD k=10"] = {"resident_s": t_res, "out_of_core_s": t_ooc, "block_steps": [r0["niter"], r1["niter"]],
                                 "file_walks": passes, "GBps_over_the_walks": passes * nb * m / t_ooc / 1e9,
                                 "d_max_rel_diff": float(np.max(np.abs(r1["d"] / r0["d"] - 1))), "out_of_core": r1["out_of_core"],
                                 "center_identical": bool(np.array_equal(r0["center"], r1["center"]))}
print(json.dumps(out))
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
