"""The in-library RCCL path of the column-sharded solve (include/bigsnpr_hip.h, bsn_comm_*).

A single-rank communicator exercises the real thing on the one-GPU test box: RCCL is dlopen'ed,
ncclCommInitRank / ncclReduceScatter / ncclAllReduce / ncclAllGather run on the library's streams over
the sample-block layout, and the solve must reproduce the plain single-GPU solve bit for bit (one rank:
every collective is the identity).  With two or more GPUs visible the two-rank run checks the sharded
solve against the unsharded one (skipped otherwise — RCCL refuses two ranks on one device; the two-rank
logic on one GPU is covered through the hook by tests/test_gpu_sharded_svd.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def test_single_rank_communicator_is_the_identity(ba):
    comm = ba.Comm(ba.Comm.unique_id(), 0, 1)
    try:
        x = np.arange(1000, dtype=np.float64) * 0.5
        d = ba.DeviceArray.from_numpy(x)
        comm.allreduce(d)
        np.testing.assert_array_equal(d.to_numpy().ravel(), x)
        n, m, k = 2501, 3000, 6
        gb = ba.bed.synthetic(n, m, seed=12)
        plain = ba.bed_randomSVD(gb, k=k)
        viacomm = ba.bed_randomSVD(gb, k=k, comm=comm, m_total=m)
        assert viacomm["niter"] == plain["niter"]
        for key in ("d", "u", "v"):
            np.testing.assert_array_equal(viacomm[key], plain[key])
        # the one-shot product of a column shard: partial vectors summed over the ranks on the device
        xv = np.random.default_rng(1).normal(size=m)
        np.testing.assert_array_equal(ba.bed_prodVec(gb, xv, center=plain["center"], scale=plain["scale"], comm=comm),
                                      ba.bed_prodVec(gb, xv, center=plain["center"], scale=plain["scale"]))
    finally:
        comm.close()


def _mock_rccl():
    sys.path.insert(0, os.path.join(ROOT, "tests", "native"))
    import build_native
    return build_native.build_mock_rccl()


@pytest.mark.parametrize("transport,world,n", [("mock", 2, 3000), ("mock", 3, 3001), ("mock", 4, 1234), ("mock", 7, 2050), ("mock", 8, 3005), ("rccl", 2, 3000)])
def test_ranks_over_the_collective_calls(tmp_path, ba, transport, world, n):
    """several ranks through comm.hip's collective calls: reduce-scatter of the panel by sample blocks, Gram
    all-reduces, all-gather of the basis block, and the sharded one-shot product.  `rccl` needs two GPUs; `mock`
    runs the same library code on the one GPU of the test box with RCCL's entry points served by a shared-memory
    stand-in (tests/native/mock_rccl.cpp, loaded through BSN_RCCL_LIBRARY)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if transport == "rccl":
        if ba.device_count() < 2:
            pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    else:
        env["BSN_RCCL_LIBRARY"] = _mock_rccl()
    m, k = 4400, 6                      # 3 ranks x 3001 samples: the last sample block is padded
    out = str(tmp_path / "rccl.json")
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "tests", "helpers", "rccl_svd_worker.py"), str(n), str(m), str(k), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out))
    assert got["same"], "ranks diverged"
    gb = ba.bed.synthetic(n, m, seed=31)
    ref = ba.bed_randomSVD(gb, k=k, tol=1e-9)
    np.testing.assert_allclose(got["d"], ref["d"], rtol=1e-7)
    x = np.random.default_rng(7).normal(size=m)
    yref = ba.bed_prodVec(gb, x, center=ref["center"], scale=ref["scale"])
    np.testing.assert_allclose(got["y"], yref, rtol=0, atol=1e-9 * np.abs(yref).max())


@pytest.mark.parametrize("world,n", [(2, 9000), (3, 13001), (8, 33601)])
def test_product_pass_in_segments_with_overlapped_reduce_scatter(tmp_path, ba, world, n):
    """Round 4: with 16-vector blocks (k_prodT on the ranks' sample-major copies) and sample blocks of whole 512-sample
    workgroup blocks the product pass of a sharded solve runs in segments whose reduce-scatters are queued on a second
    stream behind each segment.  Three runs of the same solve through the stand-in transport — segments + second stream
    (default), segments on the solve's own stream (BSN_NO_OVERLAP=1), the whole pass followed by one reduce-scatter
    (BSN_NO_SEGMENTS=1) — must agree bit for bit (the sums over the ranks are the same per element), and with the
    unsharded solve to 1e-6 (both at the default tolerance 1e-4)."""
    m, k = world * 5000 + 96, 20
    runs = {}
    for tag, extra in (("overlap", {}), ("one_stream", {"BSN_NO_OVERLAP": "1"}), ("whole", {"BSN_NO_SEGMENTS": "1"}),
                       ("fp64_gather", {"BSN_NO_COMPACT_GATHER": "1"})):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", BSN_RCCL_LIBRARY=_mock_rccl(), BSN_TEST_BLOCK="16", BSN_TEST_TOL="1e-4",
                   **extra)
        out = str(tmp_path / ("seg_%s.json" % tag))
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = str(sock.getsockname()[1])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", port,
               os.path.join(ROOT, "tests", "helpers", "rccl_svd_worker.py"), str(n), str(m), str(k), out]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[tag] = json.load(open(out))
        assert runs[tag]["same"], "ranks diverged"
    assert runs["overlap"]["segmented_passes"] > 0 and runs["one_stream"]["segmented_passes"] > 0
    assert runs["whole"]["segmented_passes"] == 0 and runs["overlap"]["tiled"] == 2
    # ... and the all-gather of every finished basis block as the 16-bit integers it is rounded to (default) against fp64
    assert runs["overlap"]["compact_gathers"] > 0 and runs["fp64_gather"]["compact_gathers"] == 0
    for tag in ("one_stream", "whole", "fp64_gather"):
        assert runs[tag]["d"] == runs["overlap"]["d"] and runs[tag]["niter"] == runs["overlap"]["niter"]
        assert runs[tag]["usum"] == runs["overlap"]["usum"] and runs[tag]["vsum"] == runs["overlap"]["vsum"]
        assert runs[tag]["y"] == runs["overlap"]["y"]
    gb = ba.bed.synthetic(n, m, seed=31)
    ref = ba.bed_randomSVD(gb, k=k, tol=1e-4, block=16)      # (default tolerance: 16-bit panels, 16 vectors = one launch)
    np.testing.assert_allclose(runs["overlap"]["d"], ref["d"], rtol=1e-6)


def _negotiate(tmp_path, world, tag, **extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BSN_RCCL_LIBRARY=_mock_rccl(), **extra)
    out = str(tmp_path / ("neg_%s.json" % tag))
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    n, m, k = world * 4608, world * 5000 + 96, 20
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "tests", "helpers", "negotiate_worker.py"), str(n), str(m), str(k), "2500", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out)), r.stderr


@pytest.mark.parametrize("world", [2, 3])
def test_first_contact_negotiation(tmp_path, world):
    """Round 5: bigsnpr_amd.comm.negotiate on a healthy transport picks the overlapped exchange; on one
    that never completes a reduce-scatter issued on the second stream (MOCK_RCCL_STALL=second_stream) the watchdog of
    the miniature solve aborts the communicator after its 2.5 s, every rank gets an error instead of a hang, they agree,
    start over with a fresh communicator and settle on the one-stream exchange.  The solve that follows runs in the
    chosen mode and finds the same numbers bit for bit either way; its collectives are timed by class."""
    good, _ = _negotiate(tmp_path, world, "good")
    assert good["report"]["mode"] == "overlap" and good["mode"].endswith("second stream")
    assert [t["mode"] for t in good["report"]["tried"]] == ["whole", "overlap"] and all(t["ok"] for t in good["report"]["tried"])
    assert good["env"] == {"BSN_NO_OVERLAP": None, "BSN_NO_SEGMENTS": None}
    assert good["n_exchange"]["reduce_scatter"] > 0 and good["n_exchange"]["exposed_wait"] > 0
    assert good["exchange_ms"]["reduce_scatter"] > 0 and good["n_exchange"]["small"] > 0 and good["n_exchange"]["all_gather"] > 0
    bad, err = _negotiate(tmp_path, world, "stall", MOCK_RCCL_STALL="second_stream")
    tried = {t["mode"]: t for t in bad["report"]["tried"]}
    assert bad["report"]["mode"] == "one_stream" and bad["mode"] == "segments, one stream"
    assert tried["whole"]["ok"] and tried["one_stream"]["ok"] and not tried["overlap"]["ok"]
    assert "did not finish within 2500 ms" in tried["overlap"]["error"] and 2500 <= tried["overlap"]["ms"] < 20000
    assert bad["env"] == {"BSN_NO_OVERLAP": "1", "BSN_NO_SEGMENTS": None}
    assert bad["n_exchange"]["exposed_wait"] == 0
    assert bad["d"] == good["d"] and bad["usum"] == good["usum"]
