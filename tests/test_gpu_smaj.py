"""The sample-major second copy of the image (bsn_bed_sample_major, round 4) and k_prodT, the product kernel that
reads it: same exact integer sums as k_prod on the variant-major image -> bit-identical results, on ragged sizes, with
missing values, on a chunk-aligned sub-range of the variants, in the warm-started solve, and against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


@pytest.mark.parametrize("n,m", [(3001, 5003), (517, 9000), (70000, 4500)])
def test_product_on_the_sample_major_copy_is_bit_identical(ba, orc, n, m, monkeypatch):
    from bigsnpr_amd import _lib
    gb = ba.bed.synthetic(n, m, seed=11)
    ob = orc.fake_bed(n, m, seed=11) if n * m < 5e7 else None
    sc = ba.bed_scaleBinom(gb)
    keep = sc["scale"] > 0
    ce, sa = np.where(keep, sc["center"], 0.0), np.where(keep, sc["scale"], 1.0)
    rng = np.random.default_rng(0)
    for ic in (None, np.arange(512, m - 7), np.arange(100, m - 200)):   # whole / chunk-aligned / unaligned range
        mm = m if ic is None else ic.size
        cc, ss = (ce, sa) if ic is None else (ce[ic], sa[ic])
        op = ba.ScaledOp(gb, None, ic, cc, ss, slices=2)
        X = ba.DeviceArray.from_numpy(rng.normal(size=(mm, 16)))
        Y = ba.DeviceArray(n, 16)
        sync = _lib.load().bsn_device_sync          # (the products are queued on the handle's stream)
        monkeypatch.setenv("BSN_NO_SMAJ", "1")
        op.prod(X, Y)
        sync()
        ref = Y.to_numpy()
        monkeypatch.delenv("BSN_NO_SMAJ")
        assert gb.sample_major()
        op.prod(X, Y)
        sync()
        np.testing.assert_array_equal(Y.to_numpy(), ref)
        assert np.abs(ref).max() > 0
        if ob is not None and ic is None:   # and both are the reference's product (16-bit panels: 1e-4 of the scale)
            want = np.stack([orc.bed_prodVec(ob, X.to_numpy()[:, v], None, None, ce, sa, 4) for v in range(2)], axis=1)
            assert np.abs(ref[:, :2] - want).max() <= 1e-3 * np.abs(want).max()
        # 9 .. 16 vectors take the copy, 8 do not (one column block): still identical to the plain path
        X8, Y8 = ba.DeviceArray.from_numpy(rng.normal(size=(mm, 8))), ba.DeviceArray(n, 8)
        op.prod(X8, Y8)
        sync()
        a = Y8.to_numpy()
        monkeypatch.setenv("BSN_NO_SMAJ", "1")
        op.prod(X8, Y8)
        sync()
        monkeypatch.delenv("BSN_NO_SMAJ")
        np.testing.assert_array_equal(Y8.to_numpy(), a)


def test_complete_data_and_solve(ba, monkeypatch):
    """no missing values (the kernel variant without the missing-value plane) and the whole 16-vector solve, whose
    product passes — warm start included — run on the copy: d, u, v bit-identical to the solve without it"""
    n, m, k = 2500, 300000, 20
    for na16 in (655, 0):                                   # 1 % missing / none
        gb = ba.bed.synthetic(n, m, seed=5, na16=na16)
        monkeypatch.setenv("BSN_NO_SMAJ", "1")
        ref = ba.bed_randomSVD(gb, k=k, block=16)
        assert ref["tiled"] == 0
        monkeypatch.delenv("BSN_NO_SMAJ")
        assert gb.sample_major()            # (round 6: a solve no longer builds the copy inside the call; bench.py does the same)
        res = ba.bed_randomSVD(gb, k=k, block=16)
        assert res["tiled"] == 2 and res["converged"] and res["warm_launches"] == 4
        for f in ("d", "u", "v", "center", "scale"):
            np.testing.assert_array_equal(res[f], ref[f])
        # (without the copy a 48-column pass of the precision schedule is two launches of k_prod<2>: more launches, same sums)
        assert res["niter"] == ref["niter"] and res["nops"] <= ref["nops"]
        # several launches per pass (5 vectors x 7 slices = 28 + 7 digit columns: a two-block and a one-block launch):
        # the copy exists but such a pass stays on k_prod — same numbers as without the copy
        wide = ba.bed_randomSVD(gb, k=5, block=5, slices=7, tol=1e-8, return_uv=False)
        monkeypatch.setenv("BSN_NO_SMAJ", "1")
        wide0 = ba.bed_randomSVD(gb, k=5, block=5, slices=7, tol=1e-8, return_uv=False)
        monkeypatch.delenv("BSN_NO_SMAJ")
        np.testing.assert_array_equal(wide["d"], wide0["d"])
        gb.release_workspace()                               # frees the copy too; the next solve starts another one behind itself
        again = ba.bed_randomSVD(gb, k=k, block=16)
        assert again["tiled"] == 0 and gb.sample_major()
        np.testing.assert_array_equal(again["d"], ref["d"])
        assert ba.bed_randomSVD(gb, k=k, block=16)["tiled"] == 2


def test_random_small_shapes(ba, monkeypatch):
    """k_prodT on shapes around its tile sizes: fewer samples than a tile (16) or a workgroup (512), fewer variants than
    a chunk (512), one / two / many slabs, ragged everything — bit-identical to k_prod."""
    from bigsnpr_amd import _lib
    sync = _lib.load().bsn_device_sync
    rng = np.random.default_rng(77)
    shapes = [(5, 100), (16, 512), (17, 513), (511, 1025), (513, 4097), (1000, 70000), (33, 20000), (2049, 3000)]
    shapes += [(int(rng.integers(1, 3000)), int(rng.integers(1, 9000))) for _ in range(8)]
    for n, m in shapes:
        gb = ba.bed.synthetic(n, m, seed=n + m, na16=int(rng.choice([0, 655, 6000])))
        nv = int(rng.integers(9, 17))
        ce, sa = rng.uniform(0.1, 1.9, m), rng.uniform(0.3, 1.0, m)
        op = ba.ScaledOp(gb, None, None, ce, sa, slices=2)
        X, Y = ba.DeviceArray.from_numpy(rng.normal(size=(m, nv))), ba.DeviceArray(n, nv)
        monkeypatch.setenv("BSN_NO_SMAJ", "1")
        op.prod(X, Y)
        sync()
        ref = Y.to_numpy()
        monkeypatch.delenv("BSN_NO_SMAJ")
        assert gb.sample_major()
        op.prod(X, Y)
        sync()
        np.testing.assert_array_equal(Y.to_numpy(), ref, err_msg="n=%d m=%d nv=%d" % (n, m, nv))
        gb.close()


def test_the_copy_made_behind_the_first_solve(ba, monkeypatch):
    """the first 16-vector solve of a handle no longer waits for the sample-major copy — it runs on the
    variant-major image alone (k_prod) and the copy is made BEHIND it: a helper thread allocates it and queues the
    transposition on a stream of its own when the call returns; a solve that starts while the copy is still on its way
    takes k_prod for its product passes until it has arrived.  Whichever pass switches over, the sums are the same
    integers: the first solve of a fresh handle, later ones (copy on its way / in place), a solve with the copy built up
    front (BSN_SMAJ_SYNC=1, what a sharded solve does) and a solve without any copy are bit-identical;
    release_workspace while a build may still be in flight, and closing such a handle, are safe."""
    n, m, k = 2500, 300000, 20
    res = {}
    for tag, env in (("behind", {}), ("up_front", {"BSN_SMAJ_SYNC": "1"}), ("none", {"BSN_NO_SMAJ": "1"})):
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        gb = ba.bed.synthetic(n, m, seed=5)
        first = ba.bed_randomSVD(gb, k=k)
        later = ba.bed_randomSVD(gb, k=k)          # (may start while the copy is still being made)
        assert first["converged"] and first["block"] == 16
        for f in ("d", "u", "v", "center", "scale"):
            np.testing.assert_array_equal(first[f], later[f], err_msg="%s: %s of the first and of a later solve" % (tag, f))
        if tag == "behind":
            assert first["tiled"] != 2             # the first solve did not wait for a copy
            assert gb.sample_major()               # (waits for the build in flight)
            third = ba.bed_randomSVD(gb, k=k)
            assert third["tiled"] == 2
            np.testing.assert_array_equal(third["u"], first["u"])
            gb.release_workspace()                 # frees the copy
            again = ba.bed_randomSVD(gb, k=k)      # ... and starts another build behind itself
            np.testing.assert_array_equal(again["u"], first["u"])
            gb.release_workspace()                 # (adopts and frees a copy that may still be on its way)
            fresh = ba.bed.synthetic(n, m, seed=5)
            ba.bed_randomSVD(fresh, k=k)
            fresh.close()                          # ... and so does closing the handle right behind its first solve
        if tag == "up_front":
            assert first["tiled"] == 2 and later["tiled"] == 2
        res[tag] = first
        gb.close()
        for kk in env:
            monkeypatch.delenv(kk)
    for f in ("d", "u", "v"):
        np.testing.assert_array_equal(res["behind"][f], res["up_front"][f], err_msg=f)
        np.testing.assert_array_equal(res["behind"][f], res["none"][f], err_msg=f)
