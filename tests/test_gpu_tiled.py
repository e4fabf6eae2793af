"""The streaming-layout copy of the image (bsn_bed_tile: tiles of 64 variants x 1024 samples) must not change a
single bit of any result: the streaming kernels do the same integer arithmetic on the same genotypes, only the
addresses differ.  Shapes are chosen off every alignment (samples not a multiple of 1024 or 4, variants not a
multiple of 64), with missing values, for one and two MFMA column blocks, full and sub-range operators, the
warm-start launches, and the solve with the fused scaling statistics."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def test_products_bit_identical_on_the_tiled_copy(ba, orc, monkeypatch):
    n, m = 3001, 5003
    ob = orc.fake_bed(n, m, seed=5, na16=1300)
    gb = ba.bed.from_payload(ob.payload, n, m)
    sc = orc.bed_scaleBinom(ob)
    ok = sc["scale"] > 0
    ce, sa = np.where(ok, sc["center"], 0.0), np.where(ok, sc["scale"], 1.0)
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=m), rng.normal(size=n)
    views = [np.arange(m), np.arange(128, 128 + 3000), np.arange(7, 7 + 2000), np.arange(4992, m)]

    def run():
        out = []
        for ic in views:
            out.append(ba.bed_prodVec(gb, x[ic], None, ic, ce[ic], sa[ic]))
            out.append(ba.bed_cprodVec(gb, y, None, ic, ce[ic], sa[ic]))
        V = rng.normal(size=(m, 5))
        rng2 = np.random.default_rng(9)
        out.extend(ba.prod_and_rowSumsSq(gb, ba.rows_along(gb), ba.cols_along(gb), ce, sa, rng2.normal(size=(m, 5))))
        out.append(ba.multLinReg(gb, None, None, rng2.normal(size=(n, 3))))
        return out

    plain = run()
    ref = orc.bed_prodVec(ob, x, None, np.arange(m), ce, sa, 4)
    assert np.abs(plain[0] - ref).max() <= 1e-9 * np.abs(ref).max()
    assert gb.tile() is True
    tiled = run()
    for a, b in zip(plain, tiled):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ba.bed_counts(gb), orc.bed_col_counts(ob))      # other kernels: untouched


@pytest.mark.parametrize("block", [0, 16])
def test_solve_bit_identical_with_and_without_the_tiled_copy(ba, monkeypatch, block):
    n, m, k = 2500, 290000 if block == 0 else 9000, 6      # 290 000 variants: with the warm start
    monkeypatch.setenv("BSN_NO_SMAJ", "1")                  # (the sample-major copy of 16-vector solves: tests/test_gpu_smaj.py)
    res = {}
    for tiled in (False, True):
        if tiled:
            monkeypatch.delenv("BSN_NO_TILED", raising=False)
        else:
            monkeypatch.setenv("BSN_NO_TILED", "1")
        gb = ba.bed.synthetic(n, m, seed=13)
        if tiled and block == 16:
            gb.tile()        # the two-block kernels do not ask for the copy (they gain nothing from it) but use one that exists
        res[tiled] = ba.bed_randomSVD(gb, k=k, block=block)
        assert res[tiled]["tiled"] == int(tiled)
        gb.close()
    a, b = res[False], res[True]
    assert a["niter"] == b["niter"] and a["nops"] == b["nops"] and a["warm_launches"] == b["warm_launches"]
    assert a["warm_launches"] == (4 if block == 0 else 0)
    for key in ("d", "u", "v", "center", "scale"):
        np.testing.assert_array_equal(a[key], b[key])


def test_release_workspace_returns_the_memory(ba):
    """bsn_bed_release_workspace gives back the solve's workspace, the tiled copy and the cached work buffers:
    what stays allocated is the image itself; the next solve rebuilds everything and finds the same answer"""
    import ctypes as C
    from bigsnpr_amd import _lib
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    n, m = 20000, 40000
    gb = ba.bed.synthetic(n, m, seed=2)
    _lib.check(_lib.load().bsn_bed_release_workspace(gb.handle))
    base = free_bytes()
    r1 = ba.bed_randomSVD(gb, k=4)
    # (round 5: the early steps of the default solve run on 24-bit panels, 8 x 3 digit columns: the second copy of the image
    # is the sample-major one; round 6: made BEHIND the first such solve — sample_major() waits for it)
    assert gb.sample_major() and ba.bed_randomSVD(gb, k=4)["tiled"] == 2
    used = base - free_bytes()
    assert used >= gb.hbm_bytes()                      # at least the second copy of the image
    _lib.check(_lib.load().bsn_bed_release_workspace(gb.handle))
    assert base - free_bytes() <= 64 << 20             # allocator granularity only
    r2 = ba.bed_randomSVD(gb, k=4)
    np.testing.assert_array_equal(r1["d"], r2["d"])
