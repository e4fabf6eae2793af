import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()          # library loaded, no handle opened yet
pids = []
for w in range(2):
    pid = os.fork()
    if pid == 0:
        gb = ba.bed.synthetic(1000, 500, seed=w + 1)
        y = ba.bed_prodVec(gb, np.ones(500))
        print("child", w, "ok", float(y.sum()), flush=True)
        os._exit(0)
    pids.append(pid)
ok = all(os.waitpid(p, 0)[1] == 0 for p in pids)
gb = ba.bed.synthetic(1000, 500, seed=1)
print("parent ok", ok, float(ba.bed_prodVec(gb, np.ones(500)).sum()))
