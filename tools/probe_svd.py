#!/usr/bin/env python3
import argparse, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400000)
ap.add_argument("--m", type=int, default=100000)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--torch-first", action="store_true")
ap.add_argument("--verbose", action="store_true")
a = ap.parse_args()
if a.torch_first:
    import torch
    print("torch", torch.__version__, torch.cuda.is_available())
import numpy as np
import bigsnpr_amd as ba
ba.selftest()
hip = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
print("hip runtime(s):", sorted(set(hip)))
t0 = time.time(); gb = ba.bed.synthetic(a.n, a.m); print("gen %.2fs" % (time.time() - t0))
for rep in range(2):
    t0 = time.time()
    res = ba.bed_randomSVD(gb, k=a.k, verbose=a.verbose and rep == 0, return_uv=False)
    wall = time.time() - t0
    print(json.dumps(dict(wall_s=wall, gpu_ms=res["gpu_ms"], niter=res["niter"], nops=res["nops"], basis=res["basis"],
                          converged=res["converged"], resid=res["max_rel_resid"], d=list(np.round(res["d"][:5], 3)),
                          snp_cols_per_s=a.m * (res["nops"] + 1) / wall)), flush=True)
