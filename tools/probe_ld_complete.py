import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bigsnpr_amd as ba
n, m, W = 400000, 100000, 2000
pos = np.arange(m, dtype=np.float64)
for na16 in (655, 0):
    gb = ba.bed.synthetic(n, m, na16=na16)
    ba.bed_ld_scores(gb, size=W / 1000.0, infos_pos=pos)
    t0 = time.perf_counter(); ld = ba.bed_ld_scores(gb, size=W / 1000.0, infos_pos=pos); t = time.perf_counter() - t0
    print("na16=%d bed_ld_scores %.3f s  mean %.6f" % (na16, t, ld.mean()), flush=True)
    gb.close()
