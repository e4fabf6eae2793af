"""Edge cases the reference's tests touch (ragged / tiny shapes, all-missing or constant
columns, duplicated indices, empty selections) through the C ABI on the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


def _bed_from_matrix(ba, orc, g):
    """g: n x m int matrix with values 0,1,2 and 3 = missing"""
    n, m = g.shape
    code = np.array([3, 2, 0, 1], dtype=np.uint8)[g]          # genotype -> PLINK 2-bit code
    n_byte = (n + 3) // 4
    payload = np.zeros((m, n_byte), dtype=np.uint8)
    for i in range(n):
        payload[:, i // 4] |= (code[i] << (2 * (i % 4))).astype(np.uint8)
    ob = orc.BedFile.from_payload(payload.ravel(), n, m)
    return ob, ba.bed.from_payload(payload.ravel(), n, m)


@pytest.mark.parametrize("n,m", [(1, 1), (3, 5), (4, 64), (5, 63), (17, 65), (255, 129), (1025, 3)])
def test_tiny_and_ragged_shapes(ba, orc, n, m):
    rng = np.random.default_rng(n * 1000 + m)
    g = rng.integers(0, 4, size=(n, m))
    ob, gb = _bed_from_matrix(ba, orc, g)
    np.testing.assert_array_equal(gb.download(), ob.payload)
    np.testing.assert_array_equal(ba.bed_counts(gb), orc.bed_col_counts(ob))
    np.testing.assert_array_equal(ba.read_bed(gb, None, None), orc.read_bed(ob))
    x, y = rng.normal(size=m), rng.normal(size=n)
    c, s = rng.normal(size=m), rng.uniform(0.5, 2, size=m)
    ref = orc.bed_prodVec(ob, x, None, None, c, s)
    np.testing.assert_allclose(ba.bed_prodVec(gb, x, center=c, scale=s), ref, rtol=0,
                               atol=1e-9 * max(np.abs(ref).max(), 1e-300))
    ref = orc.bed_cprodVec(ob, y, None, None, c, s)
    np.testing.assert_allclose(ba.bed_cprodVec(gb, y, center=c, scale=s), ref, rtol=0,
                               atol=1e-9 * max(np.abs(ref).max(), 1e-300))
    # duplicated + unsorted rows and columns (test-7-OpenMP.R:31-32)
    ir, ic = rng.integers(0, n, size=2 * n + 1), rng.integers(0, m, size=2 * m + 3)
    np.testing.assert_array_equal(ba.bed_counts(gb, ir, ic), orc.bed_col_counts(ob, ir, ic))
    xx, yy = rng.normal(size=ic.size), rng.normal(size=ir.size)
    ref = orc.bed_prodVec(ob, xx, ir, ic, c[ic], s[ic])
    np.testing.assert_allclose(ba.bed_prodVec(gb, xx, ir, ic, c[ic], s[ic]), ref, rtol=0,
                               atol=1e-9 * max(np.abs(ref).max(), 1e-300))
    ref = orc.bed_cprodVec(ob, yy, ir, ic, c[ic], s[ic])
    np.testing.assert_allclose(ba.bed_cprodVec(gb, yy, ir, ic, c[ic], s[ic]), ref, rtol=0,
                               atol=1e-9 * max(np.abs(ref).max(), 1e-300))


def test_all_missing_and_constant_columns(ba, orc):
    rng = np.random.default_rng(5)
    g = rng.integers(0, 3, size=(120, 10))
    g[:, 2] = 3      # all missing
    g[:, 4] = 1      # constant
    g[:70, 6] = 3    # > 50 % missing -> warning of bed_colstats (src/bed-fun.cpp:40-41)
    ob, gb = _bed_from_matrix(ba, orc, g)
    with pytest.warns(UserWarning, match="2 variants have >50% missing values"):
        st = ba.bed_colstats(gb)
    ref = orc.bed_colstats(ob)
    for k in st:
        np.testing.assert_array_equal(st[k], ref[k])          # incl. NaN in denoX of column 2
    with pytest.warns(UserWarning):
        sc = ba.bed_scaleBinom(gb)
    scr = orc.bed_scaleBinom(ob)
    np.testing.assert_array_equal(sc["center"], scr["center"])
    np.testing.assert_array_equal(sc["scale"], scr["scale"])
    # the all-missing column has NaN centre / scale but only ever uses the NA entry (0) of
    # bedAccScaled's table (src/bed-acc.h:104): cprodVec is finite and 0 there
    y = rng.normal(size=120)
    z = ba.bed_cprodVec(gb, y, center=sc["center"], scale=sc["scale"])
    zr = orc.bed_cprodVec(ob, y, None, None, scr["center"], scr["scale"])
    assert np.isfinite(zr).all() and zr[2] == 0 and z[2] == 0
    np.testing.assert_allclose(z, zr, rtol=0, atol=1e-9 * np.abs(zr).max())
    # a zero scale (monomorphic column under a user-supplied scaling) is non-finite in that column only
    s0 = np.where(np.arange(10) == 4, 0.0, 1.0)
    z0 = ba.bed_cprodVec(gb, y, center=np.ones(10), scale=s0)
    z0r = orc.bed_cprodVec(ob, y, None, None, np.ones(10), s0)
    assert np.array_equal(np.isfinite(z0), np.isfinite(z0r)) and not np.isfinite(z0[4])


def test_ld_small_and_edge_windows(ba, orc):
    rng = np.random.default_rng(6)
    for n, m in ((30, 3), (64, 64), (65, 130), (200, 1)):
        g = rng.integers(0, 4, size=(n, m))
        ob, gb = _bed_from_matrix(ba, orc, g)
        for size in (0.0005, 0.002, 1e3):
            ref = orc.snp_cor(ob, size=size)
            with np.errstate(all="ignore"):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    res = ba.bed_cor(gb, size=size)
            np.testing.assert_array_equal(res.p, ref[1])
            np.testing.assert_array_equal(res.i, ref[0])
            ok = ~np.isnan(ref[2])
            assert np.array_equal(np.isnan(res.x), ~ok)
            np.testing.assert_allclose(res.x[ok], ref[2][ok], rtol=0, atol=1e-12)
            np.testing.assert_allclose(ba.bed_ld_scores(gb, size=size), orc.ld_scores(ob, size=size), rtol=1e-12)


def test_svd_tiny_matrices(ba, orc):
    rng = np.random.default_rng(7)
    for n, m, k in ((40, 25, 5), (12, 300, 10), (300, 11, 10), (9, 9, 3)):
        g = rng.integers(0, 3, size=(n, m))
        g[rng.uniform(size=(n, m)) < 0.03] = 3
        ob, gb = _bed_from_matrix(ba, orc, g)
        sc = orc.bed_scaleBinom(ob)
        ic = np.nonzero(np.nan_to_num(sc["scale"]) > 0)[0]
        if min(n, ic.size) < k:
            continue
        ref = orc.dense_svd(ob, None, ic, k=k)
        res = ba.bed_randomSVD(gb, ind_col=ic, k=k, tol=1e-12, slices=7)
        np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-8, atol=1e-8 * ref["d"][0])


def test_svd_of_fewer_samples_than_the_basis_holds(ba, orc):
    """a 34-sample selection of 135 x 1702 (random shapes, seed offset 3000): the basis reaches the whole of R^34; on
    16- and 24-bit products the solve ends there with exact Ritz pairs instead of restarting on rounding noise"""
    gb, ob = ba.bed.synthetic(135, 1702, seed=196, na16=6000), orc.fake_bed(135, 1702, seed=196, na16=6000)
    ir = np.sort(np.random.default_rng(3).choice(135, 34, replace=False))
    ic = np.nonzero(orc.bed_scaleBinom(ob, ir, None)["scale"] > 0)[0]
    ref = orc.dense_svd(ob, ir, ic, k=6)
    for block, slices in ((0, 2), (0, 0), (4, 3), (16, 2), (3, 0)):
        res = ba.bed_randomSVD(gb, ind_row=ir, ind_col=ic, k=6, block=block, slices=slices)
        assert res["converged"], (block, slices)
        np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-6)


def test_svd_refuses_scalings_that_are_not_finite_or_out_of_range(ba):
    """a scale of 0 (a monomorphic variant under a caller's scaling) makes the scaled matrix non-finite: RSpectra fails on
    such input, and so does this solve — with a message, and with the handle usable afterwards (before round 5 the
    NaN reached the host eigen-solver, whose deflation search ran past its arrays); a scaling that puts the matrix at
    1e120 is refused for the overflow of its Gram matrices instead of coming back "converged" with zeros"""
    gb = ba.bed.synthetic(400, 900, seed=3)

    def scaled(factor, zero_at=None):
        def fun(obj, ind_row=None, ind_col=None, ncores=1):
            ms = ba.bed_scaleBinom(obj, ind_row, ind_col)
            ms = dict(center=np.array(ms["center"]), scale=np.array(ms["scale"]) * factor)
            if zero_at is not None:
                ms["scale"][zero_at] = 0.0
            return ms
        return fun

    with pytest.raises(ba.BsnError, match="NaN or Inf"):
        ba.bed_randomSVD(gb, fun_scaling=scaled(1.0, zero_at=7), k=5)
    with pytest.raises(ba.BsnError, match="1e-70 .. 1e70"):
        ba.bed_randomSVD(gb, fun_scaling=scaled(1e-120), k=5)
    ref = ba.bed_randomSVD(gb, k=5)
    assert ref["converged"]
    big = ba.bed_randomSVD(gb, fun_scaling=scaled(1e-40), k=5)      # inside the range: d scales with the matrix
    np.testing.assert_allclose(big["d"], ref["d"] * 1e40, rtol=1e-6)


def test_empty_selection_errors(ba, orc):
    g = np.random.default_rng(8).integers(0, 3, size=(20, 8))
    ob, gb = _bed_from_matrix(ba, orc, g)
    with pytest.raises(ba.BsnError, match="can't be empty"):
        ba.bed_counts(gb, ind_col=np.zeros(0, dtype=np.int64))
    with pytest.raises(IndexError):
        ba.bed_counts(gb, ind_col=np.array([8]))
    with pytest.raises(IndexError):
        ba.bed_prodVec(gb, np.zeros(1), ind_col=np.array([-1]))


def test_many_variants_use_several_accumulator_slabs(ba):
    """m > 2.5e6: the variant range of the product is split so that the int32 partial sums cannot
    overflow; checked through adjointness and linearity (the oracle would take minutes here)."""
    n, m = 1000, 2600000
    gb = ba.bed.synthetic(n, m, seed=9)
    sc = ba.bed_scaleBinom(gb)
    rng = np.random.default_rng(9)
    x, y = rng.normal(size=m), rng.normal(size=n)
    kw = dict(center=sc["center"], scale=sc["scale"])
    a, z = ba.bed_prodVec(gb, x, **kw), ba.bed_cprodVec(gb, y, **kw)
    assert abs(a @ y - x @ z) <= 1e-10 * np.linalg.norm(a) * np.linalg.norm(y)
    # all-ones weights on unscaled genotypes: the row sums are integers that must be exact
    rs = ba.bed_prodVec(gb, np.ones(m))
    cnt = ba.bed_counts(gb, byrow=True)
    np.testing.assert_array_equal(rs, cnt[1] + 2.0 * cnt[2])


def test_random_shape_sweep(ba, orc):
    """25 random (n, m, missing rate) with random index selections: counts by variant and by
    sample bit-exact, both products within 1e-9 of the oracle, block operator consistent."""
    rng = np.random.default_rng(2025)
    for it in range(25):
        n, m = int(rng.integers(1, 1500)), int(rng.integers(1, 400))
        p_na = float(rng.choice([0.0, 0.02, 0.3, 0.9]))
        g = rng.integers(0, 3, size=(n, m))
        g[rng.random((n, m)) < p_na] = 3
        ob, gb = _bed_from_matrix(ba, orc, g)
        ir = rng.integers(0, n, size=int(rng.integers(1, 2 * n + 2)))
        ic = rng.integers(0, m, size=int(rng.integers(1, 2 * m + 2)))
        np.testing.assert_array_equal(ba.bed_counts(gb, ir, ic), orc.bed_col_counts(ob, ir, ic))
        gsub = g[np.ix_(ir, ic)]
        want = np.stack([(gsub == c).sum(1) for c in range(4)]).astype(np.int32)
        np.testing.assert_array_equal(ba.bed_counts(gb, ir, ic, byrow=True), want)
        c, s = rng.normal(size=ic.size), rng.uniform(0.5, 2, size=ic.size)
        xx, yy = rng.normal(size=ic.size) * 10.0 ** rng.integers(-8, 8), rng.normal(size=ir.size)
        ref = orc.bed_prodVec(ob, xx, ir, ic, c, s)
        np.testing.assert_allclose(ba.bed_prodVec(gb, xx, ir, ic, c, s), ref, rtol=0,
                                   atol=1e-9 * max(np.abs(ref).max(), 1e-300), err_msg="case %d" % it)
        ref = orc.bed_cprodVec(ob, yy, ir, ic, c, s)
        np.testing.assert_allclose(ba.bed_cprodVec(gb, yy, ir, ic, c, s), ref, rtol=0,
                                   atol=1e-9 * max(np.abs(ref).max(), 1e-300), err_msg="case %d" % it)


def test_more_than_65535_variants_through_every_upload_path(ba, orc, tmp_path):
    """grid.y of the per-variant kernels is capped at 65 535: uploads, repacks and read-backs of wider
    matrices fold the variant index over grid.z — from a host payload, from a .bed file and from FBM
    bytes, with an n that leaves pad bits in the last byte and pad bytes in the 256-B pitch."""
    n, m = 37, 70001
    ob = orc.fake_bed(n, m, seed=8)
    ref_counts = orc.bed_col_counts(ob)
    gb = ba.bed.from_payload(ob.payload, n, m)
    np.testing.assert_array_equal(gb.download(), ob.payload)
    np.testing.assert_array_equal(ba.bed_counts(gb), ref_counts)
    base = str(tmp_path / "wide")
    with open(base + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(ob.payload.tobytes())
    open(base + ".fam", "w").write("".join("f%d i%d 0 0 0 -9\n" % (i, i) for i in range(n)))
    open(base + ".bim", "w").write("".join("1 s%d 0 %d A C\n" % (j, j + 1) for j in range(m)))
    gf = ba.bed(base + ".bed")
    np.testing.assert_array_equal(gf.download(), ob.payload)
    np.testing.assert_array_equal(ba.bed_counts(gf), ref_counts)
    G = orc.fbm_from_bed(ob)
    gm = ba.bed.from_fbm(G.bytes)
    np.testing.assert_array_equal(ba.bed_counts(gm), ref_counts)
    cols = np.array([0, 65534, 65535, 65536, m - 1])
    np.testing.assert_array_equal(ba.read_bed(gb, np.arange(n), cols), orc.read_bed(ob, None, cols))


def test_null_index_lists_mean_the_leading_rows_and_columns(ba, orc):
    """every entry point of the C ABI reads a NULL ind_row / ind_col as 'the first n / m' (never a
    dereference), also when n is smaller than the matrix; too large an n is the subscript error"""
    import ctypes as C
    from bigsnpr_amd import _lib
    from bigsnpr_amd._lib import f64p, i32p, ptr
    L = _lib.load()
    n, m, ns, ms = 203, 150, 77, 60
    ob = orc.fake_bed(n, m, seed=2)
    gb = ba.bed.from_payload(ob.payload, n, m)
    ir, ic = np.arange(ns), np.arange(ms)
    cnt = np.empty((ms, 4), dtype=np.int32)
    _lib.check(L.bsn_bed_col_counts(gb.handle, None, ns, None, ms, ptr(cnt, i32p)))
    np.testing.assert_array_equal(cnt.T, orc.bed_col_counts(ob, ir, ic))
    x, y = np.random.default_rng(0).normal(size=ms), np.empty(ns)
    _lib.check(L.bsn_bed_prodvec(gb.handle, None, ns, None, ms, None, None, ptr(x, f64p), ptr(y, f64p)))
    ref = orc.bed_prodVec(ob, x, ir, ic, np.zeros(ms), np.ones(ms))
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    out = np.empty((ms, ns), dtype=np.int32)
    _lib.check(L.bsn_bed_read(gb.handle, None, ns, None, ms, -1, ptr(out, i32p)))
    np.testing.assert_array_equal(out.T, orc.read_bed(ob, ir, ic, na_val=-1))
    assert L.bsn_bed_col_counts(gb.handle, None, n + 1, None, ms, ptr(cnt, i32p)) != 0
    assert b"Subscript out of bounds" in L.bsn_last_error()


def test_counts_remembered_per_row_selection(ba, orc, monkeypatch):
    """round 6: a handle remembers the code counts of the last row selection (snp_autoSVD asks three times for the column
    statistics of the same rows); every answer from memory equals a fresh count and the oracle's — other rows, column
    subsets in any order, repeated columns, the leading-rows form of a NULL list, and an image that changed underneath
    (a compacted solve reuses its sub-image in place)"""
    from bigsnpr_amd import _lib
    from bigsnpr_amd._lib import i32p, i64p, ptr
    L = _lib.load()
    n, m = 333, 700
    ob = orc.fake_bed(n, m, seed=5)
    gb = ba.bed.from_payload(ob.payload, n, m)
    rng = np.random.default_rng(8)

    def counts(ir, ic):
        res = np.empty((len(ic), 4), dtype=np.int32)
        ir64, ic64 = np.asarray(ir, dtype=np.int64), np.asarray(ic, dtype=np.int64)
        _lib.check(L.bsn_bed_col_counts(gb.handle, ptr(ir64, i64p), ir64.size, ptr(ic64, i64p), ic64.size, ptr(res, i32p)))
        return res.T

    rows_a, rows_b = np.arange(n), np.sort(rng.choice(n, 200, replace=False))
    for rows in (rows_a, rows_a, rows_b, rows_b, rows_a, rng.permutation(rows_b), np.arange(120)):
        for cols in (np.arange(m), rng.permutation(m)[:300], np.arange(50, 250), rng.integers(0, m, size=90), np.arange(m)):
            want = orc.bed_col_counts(ob, rows, cols)
            np.testing.assert_array_equal(counts(rows, cols), want)           # (from memory after the first full request)
            monkeypatch.setenv("BSN_NO_COUNTS_CACHE", "1")
            np.testing.assert_array_equal(counts(rows, cols), want)
            monkeypatch.delenv("BSN_NO_COUNTS_CACHE")
    res = np.empty((m, 4), dtype=np.int32)
    _lib.check(L.bsn_bed_col_counts(gb.handle, None, 120, None, m, ptr(res, i32p)))      # NULL lists: the leading 120 rows
    np.testing.assert_array_equal(res.T, orc.bed_col_counts(ob, np.arange(120), np.arange(m)))
    _lib.check(L.bsn_bed_col_counts(gb.handle, None, 120, None, m, ptr(res, i32p)))
    np.testing.assert_array_equal(res.T, orc.bed_col_counts(ob, np.arange(120), np.arange(m)))
    # the statistics that are derived from the counts, twice, and the MAF / colstats entries of the FBM path
    st1, st2 = ba.bed_colstats(gb, rows_b, np.arange(m)), ba.bed_colstats(gb, rows_b, np.arange(m))
    for key in st1:
        np.testing.assert_array_equal(st1[key], st2[key])
    ref = orc.bed_colstats(ob, rows_b, np.arange(m))
    np.testing.assert_array_equal(st1["sumX"], ref["sumX"])
    assert L.bsn_bed_col_counts(gb.handle, None, n, ptr(np.array([0, m], dtype=np.int64), i64p), 2, ptr(res, i32p)) != 0
    assert b"Subscript out of bounds" in L.bsn_last_error()


def test_order_decreasing_in_groups_is_rs_stable_order(ba):
    """bsn_order_decreasing (R/clumping.R:106: order(S.chr, decreasing = TRUE) for every chromosome in one device call)
    against numpy's stable sort per group: continuous statistics, heavy ties (MAF of few samples), both zeros, one group,
    many groups of uneven sizes; the inverse permutation; NaN statistics take the host path inside _OrdersAhead"""
    import ctypes as C
    from bigsnpr_amd import _lib, ld
    from bigsnpr_amd._lib import f64p, i32p, ptr
    L = _lib.load()
    rng = np.random.default_rng(12)
    for sizes, kind in (([70000], "cont"), ([5000, 1, 17, 40000, 3000], "ties"), ([300] * 40, "ties"), ([9000, 9000], "zeros"),
                        (list(rng.integers(1, 3000, size=300)), "cont")):
        stats = []
        for sz in sizes:
            if kind == "cont":
                v = rng.random(int(sz))
            elif kind == "ties":
                a = rng.integers(0, 41, size=int(sz)) / 40.0
                v = np.minimum(a, 1 - a)
            else:
                v = rng.choice([0.0, -0.0, 0.25, 1e-300, -1e-300], size=int(sz))
            stats.append(v)
        allS = np.concatenate(stats)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        ord_, rank = np.empty(allS.size, dtype=np.int32), np.empty(allS.size, dtype=np.int32)
        _lib.check(L.bsn_order_decreasing(ptr(allS, f64p), allS.size, off.ctypes.data_as(C.POINTER(C.c_int64)), len(sizes),
                                          ptr(ord_, i32p), ptr(rank, i32p)))
        for g, v in enumerate(stats):
            want = np.argsort(-v, kind="stable")
            got = ord_[off[g]:off[g + 1]]
            np.testing.assert_array_equal(got, want)
            inv = np.empty(v.size, dtype=np.int32); inv[want] = np.arange(v.size)
            np.testing.assert_array_equal(rank[off[g]:off[g + 1]], inv)
        ahead = ld._OrdersAhead(stats)
        for g, v in enumerate(stats):
            o, r = ahead.get(g)
            np.testing.assert_array_equal(o, np.argsort(-v, kind="stable"))
    bad = np.array([0.5, np.nan, 0.25] * 2000)
    assert L.bsn_order_decreasing(ptr(bad, f64p), bad.size, None, 0, ptr(np.empty(bad.size, dtype=np.int32), i32p),
                                  ptr(np.empty(bad.size, dtype=np.int32), i32p)) != 0
    ahead = ld._OrdersAhead([bad])
    np.testing.assert_array_equal(ahead.get(0)[0], np.argsort(-bad, kind="stable"))
