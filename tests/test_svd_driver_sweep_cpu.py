"""A FIXED sweep of the backend-independent SVD driver (bigsnpr_amd/csrc/svd_driver.hpp + orth_small.hpp + dense_small.hpp,
the sources the product instantiates with the HIP backend) on the CPU harness of tests/native — VERDICT r5 #9: the next change
of the driver is certified by `pytest -m "not gpu"`, not by hand-run sweeps.  Seeded, so the same cases every run:

* the HIP wrapper's AUTOMATIC mode replayed (16-bit products, precision schedule, second solve on 56-bit products when the
  driver reports an inexact exhaustion or triplets below the products' resolution): genotype-like matrices with and without
  missing values, tiny and small shapes, every block size, both orthonormalisation paths — singular values to 1e-6 of numpy's;
* degenerate matrices (exact low rank, duplicated columns, zero rows, repeated singular values): converged and right, or
  reported as not converged — never converged and wrong;
* the five defect seeds of round 5 are pinned one by one in tests/test_svd_driver_cpu.py.

2 000 cases, about a minute; a wall-clock cap keeps a slow host from stalling the suite (the sweep then must still have
covered at least half of its cases)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "native"))

CAP_S = 240.0


@pytest.fixture(scope="module")
def nt():
    import build_native
    return C.CDLL(build_native.build())


def _auto_solve(nt, T, A, k, block, seed):
    """svd.hip's automatic mode: 16-bit base + schedule; 56-bit products when the driver says so"""
    b = block if block else 8
    nt.nt_set_slices(2)
    nt.nt_set_schedule(C.c_double(1e-7), 3, 1)
    r = T.host_svd(nt, A, k, tol=1e-4, block=b, seed=seed)
    if r["refused"]:
        return r
    if r["resolve"]:
        nt.nt_set_slices(7)
        nt.nt_set_schedule(C.c_double(0.0), 0, 0)
        r = T.host_svd(nt, A, k, tol=1e-4, block=min(b, 4), seed=seed)
    return r


def _reset(nt):
    nt.nt_set_slices(0)
    nt.nt_set_schedule(C.c_double(0.0), 0, 0)
    nt.nt_set_fused(0)


def test_automatic_mode_on_1600_genotype_like_matrices(nt):
    import test_svd_driver_cpu as T
    rng = np.random.default_rng(60601)
    t0, done, bad = time.time(), 0, []
    try:
        for trial in range(1600):
            if time.time() - t0 > CAP_S:
                break
            tiny = trial % 2 == 0
            n, m = (int(rng.integers(5, 48)), int(rng.integers(6, 70))) if tiny else (int(rng.integers(30, 300)), int(rng.integers(40, 500)))
            f = rng.uniform(0.05, 0.5, size=m)
            G = rng.binomial(2, f, size=(n, m)).astype(float)
            if rng.random() < 0.5:
                G[rng.random(G.shape) < 0.05] = np.nan
            with np.errstate(all="ignore"):
                mu = np.nanmean(G, axis=0)
            p = mu / 2
            sd = np.sqrt(2 * p * (1 - p))
            keep = np.isfinite(sd) & (sd > 0)
            if keep.sum() < 3:
                continue
            A = np.where(np.isnan(G), 0.0, (G - mu) / np.where(keep, sd, 1.0))[:, keep]
            n, m = A.shape
            kmax = min(n, m) - 1
            if kmax < 1:
                continue
            k = int(rng.integers(1, (kmax if tiny else min(kmax, 25)) + 1))
            block = int(rng.choice([0, 1, 2, 8, 16]))
            nt.nt_set_fused(int(rng.integers(0, 2)))
            r = _auto_solve(nt, T, A, k, block, trial + 1)
            d_true = np.linalg.svd(A, compute_uv=False)[:k]
            done += 1
            tag = (trial, n, m, k, block)
            if r["refused"] or not r["converged"]:
                bad.append(("not converged / refused",) + tag + (r["resid"], r["restarts"]))
            elif not (block == 1 and k > 4 and not tiny):     # (single-vector Lanczos and clusters: DESIGN.md section 6 ii)
                if not np.allclose(r["d"], d_true, rtol=1e-6, atol=1e-6 * d_true[0]):
                    bad.append(("wrong",) + tag + (float(np.abs(r["d"] - d_true).max() / d_true[0]), r["exhausted"]))
    finally:
        _reset(nt)
    assert not bad, bad[:10]
    assert done >= 800, "only %d cases inside the time cap" % done


def test_degenerate_matrices_are_right_or_reported(nt):
    import test_svd_driver_cpu as T
    rng = np.random.default_rng(60602)
    t0, done, bad = time.time(), 0, []
    try:
        for trial in range(400):
            if time.time() - t0 > CAP_S / 2:
                break
            n, m = int(rng.integers(6, 90)), int(rng.integers(6, 140))
            kind = trial % 4
            if kind == 0:      # exact low rank
                r0 = int(rng.integers(1, min(n, m)))
                A = rng.normal(size=(n, r0)) @ rng.normal(size=(r0, m))
            elif kind == 1:    # duplicated columns and zero rows
                A = rng.normal(size=(n, m))
                A[:, rng.integers(0, m, size=m // 3)] = A[:, [0]]
                A[rng.random(n) < 0.2] = 0
            elif kind == 2:    # a geometric spectrum over eight decades
                q = min(n, m)
                U, _ = np.linalg.qr(rng.normal(size=(n, q)))
                V, _ = np.linalg.qr(rng.normal(size=(m, q)))
                A = (U * np.logspace(0, -8, q)) @ V.T
            else:              # real but small trailing values (must not be vouched for on 16-bit products)
                q = min(n, m)
                U, _ = np.linalg.qr(rng.normal(size=(n, q)))
                V, _ = np.linalg.qr(rng.normal(size=(m, q)))
                s = np.r_[np.linspace(100, 40, min(4, q)), np.linspace(1e-2, 5e-3, max(0, q - 4))][:q]
                A = (U * s) @ V.T
            kmax = min(n, m) - 1
            k = int(rng.integers(1, min(kmax, 12) + 1))
            block = int(rng.choice([0, 2, 4, 8, 16]))
            nt.nt_set_fused(int(rng.integers(0, 2)))
            r = _auto_solve(nt, T, A, k, block, trial + 7)
            d_true = np.linalg.svd(A, compute_uv=False)[:k]
            done += 1
            if r["refused"] or not r["converged"]:
                continue                      # reported: the caller gets a warning / an error, not a wrong answer
            sig = d_true > 1e-7 * d_true[0]   # (values below the hard zero of the driver, 1e-5 sigma_1, come back as ~ 0)
            big = d_true > 2e-5 * d_true[0]
            if not np.allclose(r["d"][big], d_true[big], rtol=2e-6) or np.any(r["d"][~sig] > 1e-4 * d_true[0]):
                bad.append((trial, kind, n, m, k, block, r["d"].tolist(), d_true.tolist()))
    finally:
        _reset(nt)
    assert not bad, bad[:3]
    assert done >= 200
