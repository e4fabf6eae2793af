import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bigsnpr_amd as ba
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 2**20
ba.selftest()
rng = np.random.default_rng(0)
base = None
for it in range(6):
    for _ in range(50):
        gb = ba.bed.synthetic(3000, 2000, seed=it + 1)
        sc = ba.bed_scaleBinom(gb)
        ba.bed_prodVec(gb, rng.normal(size=2000), center=sc["center"], scale=sc["scale"])
        ba.bed_cprodVec(gb, rng.normal(size=3000), center=sc["center"], scale=sc["scale"])
        ba.bed_randomSVD(gb, k=3)
        ba.bed_ld_scores(gb, size=50)
        ba.bed_clumping(gb, infos_chr=np.ones(2000, dtype=int), infos_pos=1000.0 * np.arange(2000))
        gb.close()
    f = free_mb()
    base = f if base is None else base
    print("iteration block %d: free %.0f MB (delta %.0f MB)" % (it, f, f - base), flush=True)
