import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bigsnpr_amd as ba
n, m = 400000, 80000          # 8 GB payload
nb = (n + 3) // 4
rng = np.random.default_rng(0)
payload = rng.integers(0, 256, size=nb * m, dtype=np.uint8)
ba.selftest()
for rep in range(2):
    t0 = time.perf_counter(); gb = ba.bed.from_payload(payload, n, m); t1 = time.perf_counter()
    print("from_host: %.2f s  %.2f GB/s" % (t1 - t0, payload.nbytes / (t1 - t0) / 1e9), flush=True)
    gb.close()
path = "/tmp/probe.bed"
with open(path, "wb") as f:
    f.write(bytes([0x6C, 0x1B, 0x01])); f.write(payload.tobytes())
open("/tmp/probe.bim", "w").write("".join("1 s%d 0 %d A C\n" % (j, j) for j in range(m)))
open("/tmp/probe.fam", "w").write("".join("f i%d 0 0 0 -9\n" % i for i in range(n)))
for rep in range(2):
    t0 = time.perf_counter(); gb = ba.bed(path); t1 = time.perf_counter()
    print("open(file, page cache warm): %.2f s  %.2f GB/s" % (t1 - t0, payload.nbytes / (t1 - t0) / 1e9), flush=True)
    gb.close()
