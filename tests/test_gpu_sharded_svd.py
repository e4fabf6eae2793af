"""The column-sharded bed_randomSVD on real kernels: two ranks hold half of the
variants each and sum the n x b panel through the `allreduce` hook (gloo on a host copy here,
RCCL on the device buffer in bench.py).  The result must equal the unsharded solve up to the
24-bit fixed-point image of the panels (each shard scales its part of Z by its own maximum):
same number of steps, d within 1e-7 (north_star bar: 1e-6), u / v within 1e-5."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,m,k,tol", [(1500, 2200, 6, 1e-9), (1200, 280000, 5, 1e-4)])
def test_two_shards_equal_one(tmp_path, n, m, k, tol):
    """the second shape is wide enough for the warm start on a variant subset (>= 262 144 variants over
    all ranks): its two launches run through the same sample-block collectives"""
    import bigsnpr_amd as ba
    out = str(tmp_path / "sharded.json")
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "tests", "helpers", "sharded_svd_worker.py"), str(n), str(m), str(k), out, str(tol)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out))
    assert got["same"], "ranks diverged"
    gb = ba.bed.synthetic(n, m, seed=31)
    ref = ba.bed_randomSVD(gb, k=k, tol=tol)
    assert got["niter"] == ref["niter"] and got["warm_launches"] == ref["warm_launches"] == (4 if m >= 262144 else 0)
    np.testing.assert_allclose(got["d"], ref["d"], rtol=1e-7 if tol < 1e-8 else 1e-6)
    if tol > 1e-8:
        return                      # vectors are only compared on the tight solve
    v = np.asarray(got["v"])
    sgn = np.sign((v * ref["v"]).sum(0))
    np.testing.assert_allclose(v * sgn, ref["v"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(np.asarray(got["u0"]) * sgn[0], ref["u"][:, 0], rtol=0, atol=1e-5)
