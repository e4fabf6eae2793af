"""bench.py as the driver launches it for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`), on the one GPU of the test box: two ranks share device 0 and RCCL refuses that.  The script then
takes — on every rank together — the host all-reduce hook over gloo and SAYS SO in its one line (`fallback`,
`config.parallelism`: not an RCCL number, but a run that ends with a line; round 5); with `--no-fallback` it exits
non-zero on every rank instead.  What is checked is
the launch path around the solver: rendezvous, column sharding, the one JSON line on rank 0's stdout, whole-job
aggregation, and that the sharded solve finds the singular values of the single-rank one."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "2", "--warmup", "1", "--samples", "20000", "--variants", "60000", "--k", "5",
        "--no-cpu-baseline", "--no-ingest"]


def _run(cmd, **extra_env):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", **extra_env))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # the ONE line of the bench contract
    return json.loads(lines[0]), r.stderr


def _port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])


def test_two_ranks_without_rccl_do_not_fall_back_silently():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", _port(), "bench.py", "--gpus", "2", "--no-fallback"] + ARGS,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode != 0
    assert "not falling back" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]   # no bench line


def test_two_ranks_on_one_gpu_fall_back_and_agree():
    one, _ = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS)
    port = _port()
    two, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                     "--master-addr", "127.0.0.1", "--master-port", port, "bench.py", "--gpus", "2"]
                    + ARGS)
    assert "falling back to the host all-reduce hook" in err
    assert two["fallback"] and two["fallback"]["reason"] and one["fallback"] is None
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["warmup"] == 1
    assert two["config"]["m_total"] == 60000 and two["config"]["m_per_gpu"] == 30000
    assert "FALLBACK" in two["config"]["parallelism"]
    assert two["converged"] and one["converged"]
    np.testing.assert_allclose(two["sigma"], one["sigma"], rtol=1e-6)
    for rec in (one, two):
        assert rec["value"] > 0 and rec["unit"] == "SNP-cols/s" and rec["roofline"]["bound"] in ("hbm", "mfma")
        # whole-job value = total columns x passes / wall
        np.testing.assert_allclose(rec["value"],
                                   rec["config"]["m_total"] * rec["passes_per_solve"] / (rec["ms_per_step"] * 1e-3),
                                   rtol=1e-9)


def test_two_ranks_through_the_in_library_collectives():
    """the same launch with RCCL's entry points served by the shared-memory stand-in (tests/native/mock_rccl.cpp):
    bench.py then takes its normal N > 1 route — communicator, self-test all-reduce, in-library panel exchange —
    instead of the fallback, and must find the single-rank singular values with the same number of block steps"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "native"))
    import build_native
    mock = build_native.build_mock_rccl()
    one, _ = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    two, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                     "--master-addr", "127.0.0.1", "--master-port", port, "bench.py", "--gpus", "2"] + ARGS,
                    BSN_RCCL_LIBRARY=mock)
    assert "falling back" not in err
    assert two["n_gpus"] == 2 and "in-library RCCL" in two["config"]["parallelism"]
    assert two["converged"] and two["niter"] == one["niter"]
    np.testing.assert_allclose(two["sigma"], one["sigma"], rtol=1e-6)
    # round 5: the line explains its exchange — the mode the first-contact probe chose (all three passed: the fastest),
    # HIP-event time per collective class, exposed against hidden, per-rank wall times and the slowest rank's kernels
    ex = two["exchange"]
    assert ex["first_contact"]["mode"] == "overlap" and [t["mode"] for t in ex["first_contact"]["tried"]] == ["whole", "overlap"]
    assert all(t["ok"] for t in ex["first_contact"]["tried"])
    assert set(ex["ms_per_solve"]) == {"reduce_scatter", "all_gather", "small", "exposed_wait"}
    assert ex["collectives_per_solve"]["reduce_scatter"] >= two["niter"] and ex["ms_per_solve"]["reduce_scatter"] > 0
    assert ex["exposed_ms_per_solve"] > 0 and len(ex["per_rank"]["ms_per_step"]) == 2
    assert ex["per_rank"]["ms_per_step_max"] >= ex["per_rank"]["ms_per_step_min"] > 0 and ex["per_rank"]["slowest_rank_kernels_avg_ms"]


def test_a_stalled_exchange_stream_is_survived():
    """the overlapped exchange meets a transport that never completes a reduce-scatter on the second stream
    (the stand-in's MOCK_RCCL_STALL=second_stream).  The first-contact probe's watchdog ends the miniature solve with an
    error instead of a hang, the ranks agree over gloo, a fresh communicator is made and the run goes on in the
    one-stream exchange — same singular values, and the JSON says what happened."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "native"))
    import build_native
    mock = build_native.build_mock_rccl()
    one, _ = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS)
    two, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                     "--master-addr", "127.0.0.1", "--master-port", _port(), "bench.py", "--gpus", "2",
                     "--exchange-timeout-ms", "3000"] + ARGS, BSN_RCCL_LIBRARY=mock, MOCK_RCCL_STALL="second_stream")
    fc = two["exchange"]["first_contact"]
    assert fc["mode"] == "one_stream"
    tried = {t["mode"]: t for t in fc["tried"]}
    assert tried["whole"]["ok"] and not tried["overlap"]["ok"] and tried["one_stream"]["ok"]
    assert "did not finish within 3000 ms" in tried["overlap"]["error"] and tried["overlap"]["ms"] >= 3000
    assert "communicator aborted" in err
    assert two["converged"] and two["niter"] == one["niter"]
    np.testing.assert_allclose(two["sigma"], one["sigma"], rtol=1e-6)


def test_other_workloads_print_one_line():
    """`--workload matvec` (config C2) and `--workload ld` (config C5) at toy sizes: one JSON line each with the
    contract's keys"""
    for args in (["--workload", "matvec", "--samples", "3000", "--variants", "5000", "--steps", "3", "--no-cpu-baseline"],
                 ["--workload", "ld", "--samples", "3000", "--variants", "4000", "--window", "200", "--steps", "1",
                  "--warmup", "1"]):
        rec, _ = _run([sys.executable, "bench.py"] + args)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in rec, key
        assert rec["value"] > 0 and rec["n_gpus"] == 1 and "workload" in rec["config"]


def test_single_gpu_line_carries_the_autosvd_record():
    """the driver's command shape at a toy size: one line, with the objects measured outside the timed region — among them
    `auto_svd` (round 6: snp_autoSVD twice on the timed image, stage by stage)"""
    rec, _ = _run([sys.executable, "bench.py"] + ARGS + ["--no-cold", "--no-wide"])
    for key in ("metric", "value", "roofline", "accuracy", "auto_svd"):
        assert key in rec, key
    av = rec["auto_svd"]
    assert "error" not in av, av
    assert av["second_call_s"] > 0 and 0 < av["kept_variants"] <= av["variants"] == 60000
    assert {"snp_clumping", "big_randomSVD", "dist_ogk"} <= set(av["stages_second_call_s"])
