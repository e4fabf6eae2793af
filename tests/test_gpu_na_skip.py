"""The kernels that issue the missing-value plane only for K-steps with a missing code (k_cprod / k_prodT NASKIP,
round 5): the same integer sums as the plain kernels -> bit-identical panels and solves, on nearly complete data, at
1 % scattered missing values (where no step is free), and on data whose missing values come in batches; the sampled
share of free steps and the choice the library makes from it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def batch_structured(n, m, seed=3):
    """CODE_012 bytes with missing values only in some (512 samples x 512 variants) blocks, 5 % inside them"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 3, size=(n, m), dtype=np.uint8)
    blocks = rng.random(size=((n + 511) // 512, (m + 511) // 512)) < 0.2
    inside = np.repeat(np.repeat(blocks, 512, axis=0), 512, axis=1)[:n, :m]
    g[inside & (rng.random(size=(n, m)) < 0.05)] = 3
    return g


def handles(ba):
    yield "1e-4 scattered", ba.bed.synthetic(3001, 5003, seed=11, na16=6)
    yield "1e-3 scattered", ba.bed.synthetic(2500, 4096, seed=12, na16=66)
    yield "1 % scattered", ba.bed.synthetic(1800, 3100, seed=13)
    yield "batches", ba.bed.from_fbm(batch_structured(2100, 2900))


@pytest.mark.parametrize("slices", [2, 3])
def test_panels_are_bit_identical(ba, monkeypatch, slices):
    from bigsnpr_amd import _lib
    sync = _lib.load().bsn_device_sync
    rng = np.random.default_rng(slices)
    for name, gb in handles(ba):
        n, m = gb.nrow, gb.ncol
        sc = ba.bed_scaleBinom(gb)
        keep = sc["scale"] > 0
        ce, sa = np.where(keep, sc["center"], 0.0), np.where(keep, sc["scale"], 1.0)
        assert gb.sample_major()
        lists = (None, np.arange(512, m - 7), np.sort(rng.choice(m, size=m // 2, replace=False)))
        for ic in lists:   # whole image / chunk-aligned range (k_prodT) / a gather list (k_cprod without buffer loads)
            mm = m if ic is None else ic.size
            cc, ss = (ce, sa) if ic is None else (ce[ic], sa[ic])
            op = ba.ScaledOp(gb, None, ic, cc, ss, slices=slices)
            X = ba.DeviceArray.from_numpy(rng.normal(size=(mm, 16)))
            Q = ba.DeviceArray.from_numpy(rng.normal(size=(n, 16)))
            Y, Z = ba.DeviceArray(n, 16), ba.DeviceArray(mm, 16)
            got = {}
            for force in ("0", "1"):
                monkeypatch.setenv("BSN_NA_SKIP", force)
                op.prod(X, Y)
                op.cprod(Q, Z)
                sync()
                got[force] = (Y.to_numpy().copy(), Z.to_numpy().copy())
            monkeypatch.delenv("BSN_NA_SKIP")
            np.testing.assert_array_equal(got["1"][0], got["0"][0], err_msg="prod, %s" % name)
            np.testing.assert_array_equal(got["1"][1], got["0"][1], err_msg="cprod, %s" % name)
            assert np.abs(got["0"][0]).max() > 0 and np.abs(got["0"][1]).max() > 0
            op.close()
        gb.close()


def test_the_solve_chooses_from_the_sampled_share(ba, monkeypatch):
    """nearly complete data: ~90 % of the K-steps hold no missing code, the passes end on the skipping kernels, and
    d / u / v are those of the plain kernels; at 1 % scattered missing values no step is free and nothing changes"""
    n, m, k = 3072, 120000, 20   # (a whole number of 256-byte row segments: the pad samples of a ragged row are free steps too)
    gb = ba.bed.synthetic(n, m, seed=5, na16=6)
    monkeypatch.setenv("BSN_NA_SKIP", "0")
    ref = ba.bed_randomSVD(gb, k=k, block=16)
    assert ref["na_skip"] == 0
    monkeypatch.delenv("BSN_NA_SKIP")
    r = ba.bed_randomSVD(gb, k=k, block=16)
    assert 0.85 < r["na_free_steps"][0] < 0.95 and 0.85 < r["na_free_steps"][1] < 0.95, r["na_free_steps"]   # 0.9999^1024 = 0.903
    assert r["na_skip"] == 3
    for key in ("d", "u", "v"):
        np.testing.assert_array_equal(r[key], ref[key])
    monkeypatch.setenv("BSN_NA_SKIP_MIN", "0.99")   # the threshold is the caller's to move
    assert ba.bed_randomSVD(gb, k=k, block=16)["na_skip"] == 0
    monkeypatch.delenv("BSN_NA_SKIP_MIN")
    gb.close()
    gb = ba.bed.synthetic(n, m, seed=5)
    r = ba.bed_randomSVD(gb, k=k, block=16)
    r = ba.bed_randomSVD(gb, k=k, block=16)
    assert r["na_free_steps"][0] < 0.01 and r["na_free_steps"][1] < 0.01 and r["na_skip"] == 0
    gb.close()


def test_batches_of_missing_values_are_seen(ba):
    g = batch_structured(4096, 6000)
    gb = ba.bed.from_fbm(g)
    r = ba.bed_randomSVD(gb, k=10, block=16)
    r = ba.bed_randomSVD(gb, k=10, block=16)
    # about 80 % of the 512 x 512 blocks are complete (96 blocks drawn), and a K-step lies inside one block
    share = 1.0 - np.mean([(g[i:i + 512, j:j + 512] == 3).any() for i in range(0, 4096, 512) for j in range(0, 6000, 512)])
    assert abs(r["na_free_steps"][0] - share) < 0.05 and abs(r["na_free_steps"][1] - share) < 0.05, (r["na_free_steps"], share)
    assert r["na_skip"] == 3
    gb.close()
