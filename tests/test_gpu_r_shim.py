"""The R shim RUNNING (bindings/R/bigsnprhip/src/bigsnpr_hip_shim.c against the stand-in R runtime of tests/rstub): every
`.Call` target is called by its registered name with R-like arguments — 1-based integer indices, RC objects as
environments whose `address` / `backingfile` / `code256` fields the shim reads like the reference's native code
does (src/bed-prod-vec.cpp:23, src/colstats.cpp:13-14) — and compared with the oracle."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "rstub"))

CODE_012 = np.array([0, 1, 2] + [np.nan] * 253)
CODE_IMPUTE_PRED = np.array([0, 1, 2, np.nan, 0, 1, 2] + [np.nan] * 249)          # R/bigSNP-class.R:10
CODE_DOSAGE = np.r_[[0, 1, 2, np.nan, 0, 1, 2], np.round(np.arange(201) * 0.01, 2), [np.nan] * 48]  # :13


@pytest.fixture(scope="module")
def R():
    import rshim
    r = rshim.R()
    yield r
    r.call("_bigsnpr_fbm_cache_clear_hip")
    r.reset()


def _bed_obj(R, path, n, m):
    xp = R.call("_bigsnpr_bedXPtr", path, int(n), int(m))
    return R.env(address=xp)


def test_bed_entry_points(R, orc, golden_dir, missing_bed):
    ob = missing_bed
    obj = _bed_obj(R, os.path.join(golden_dir, "example-missing.bed"), ob.n, ob.m)
    rng = np.random.default_rng(0)
    ir = np.sort(rng.choice(ob.n, 150, replace=False))
    ic = rng.choice(ob.m, 300, replace=True)                       # arbitrary order, duplicates allowed
    st = R.call("_bigsnpr_bed_colstats", obj, ir + 1, ic + 1, 1)
    ref = orc.bed_colstats(ob, ir, ic)
    for k in ("sumX", "denoX", "nb_nona_col"):
        np.testing.assert_array_equal(st[k], ref[k])
    np.testing.assert_array_equal(R.call("_bigsnpr_bed_col_counts_cpp", obj, ir + 1, ic + 1, 1), orc.bed_col_counts(ob, ir, ic))
    sc = orc.bed_scaleBinom(ob, ir, ic)
    ok = sc["scale"] > 0
    ic, ce, sa = ic[ok], sc["center"][ok], sc["scale"][ok]
    x, y = rng.normal(size=ic.size), rng.normal(size=ir.size)
    got = R.call("_bigsnpr_bed_pMatVec4", obj, ir + 1, ic + 1, ce, sa, x, 2)
    np.testing.assert_allclose(got, orc.bed_prodVec(ob, x, ir, ic, ce, sa, 2), rtol=0, atol=1e-9 * np.abs(got).max())
    got = R.call("_bigsnpr_bed_cpMatVec4", obj, ir + 1, ic + 1, ce, sa, y, 2)
    np.testing.assert_allclose(got, orc.bed_cprodVec(ob, y, ir, ic, ce, sa, 2), rtol=0, atol=1e-9 * np.abs(got).max())
    g = R.call("_bigsnpr_read_bed", obj, ir[:20] + 1, ic[:30] + 1)
    np.testing.assert_array_equal(np.where(g == -2147483648, -1, g), orc.read_bed(ob, ir[:20], ic[:30]))   # NA_integer_
    gs = R.call("_bigsnpr_read_bed_scaled", obj, ir[:20] + 1, ic[:30] + 1, ce[:30], sa[:30])
    np.testing.assert_array_equal(gs, orc.read_bed_scaled(ob, ir[:20], ic[:30], ce[:30], sa[:30]))


def test_svd_ld_and_clumping_through_the_shim(R, orc, golden_dir, example_bed, tmp_path):
    ob = example_bed
    obj = _bed_obj(R, os.path.join(golden_dir, "example.bed"), ob.n, ob.m)
    ir, ic = np.arange(ob.n), np.arange(0, ob.m, 3)
    # whole-solve SVD, scaling evaluated inside (center = scale = NULL)
    res = R.call("_bigsnpr_bed_randomSVD_hip", obj, ir + 1, ic + 1, None, None, 10, 1e-4, False)
    ref = orc.dense_svd(ob, ir, ic, k=10)
    np.testing.assert_allclose(res["d"], ref["d"], rtol=1e-6)
    np.testing.assert_array_equal(res["center"], ref["center"])
    assert res["u"].shape == (ob.n, 10) and res["v"].shape == (ic.size, 10)
    np.testing.assert_allclose(np.abs(np.sum(res["u"] * ref["u"], axis=0)), 1.0, atol=1e-4)
    # tol = c(tol, slices, block, vec.floor): the accuracy / cost frontier is reachable from the .Call —
    # 56-bit panels to tol 1e-10 give the oracle's vectors to 1e-6
    tight = R.call("_bigsnpr_bed_randomSVD_hip", obj, ir + 1, ic + 1, None, None, 10, np.array([1e-10, 7.0, 4.0, 0.0]), False)
    np.testing.assert_allclose(tight["d"], ref["d"], rtol=1e-10)
    sgn = np.sign(np.sum(tight["u"] * ref["u"], axis=0))
    gap_ok = np.r_[True, np.diff(-ref["d"]) / ref["d"][0] > 1e-4] & np.r_[np.diff(-ref["d"]) / ref["d"][0] > 1e-4, True]
    assert np.abs(tight["u"] * sgn - ref["u"])[:, gap_ok].max() < 1e-6
    # corMat -> list(i, p, x) as R/corr.R:43-47 consumes it; ld_scores
    ic2 = np.arange(400)
    pos = 1000.0 * np.arange(1, ic2.size + 1)
    thr = orc.cor_thresholds(ob.n, alpha=0.05)
    got = R.call("_bigsnpr_corMat", obj, ir + 1, ic2 + 1, 30e3, thr, pos, True, 1)
    i, p, x = orc.corMat(ob, ir, ic2, 30e3, thr, pos, True)
    np.testing.assert_array_equal(got["i"], i)
    np.testing.assert_array_equal(got["p"], p)
    np.testing.assert_allclose(got["x"], x, rtol=0, atol=1e-12)
    np.testing.assert_allclose(R.call("_bigsnpr_ld_scores", obj, ir + 1, ic2 + 1, 30e3, pos, 1),
                               orc.ld_scores(ob, ir, ic2, 30, pos), rtol=1e-12)
    # bed_clumping_chr: `keep` is a 1 x m integer FBM initialised to -1 that the native code writes in place
    infos_chr = np.ones(ob.m, dtype=int)
    bim_pos = 1000.0 * np.arange(1, ob.m + 1)
    st = orc.bed_colstats(ob, ir, np.arange(ob.m))
    center, scale = st["sumX"] / st["nb_nona_col"], np.sqrt(st["denoX"])
    S = np.minimum(st["sumX"], 2.0 * st["nb_nona_col"] - st["sumX"])
    ord_ = orc.r_order_decreasing(S).astype(np.int32)
    rank = orc._rank_from_order(ord_)
    bk = tmp_path / "keep.bk"
    np.full(ob.m, -1, dtype=np.int32).tofile(bk)
    BM2 = R.env(backingfile=str(bk))
    R.call("_bigsnpr_bed_clumping_chr", obj, BM2, ir + 1, np.arange(ob.m) + 1, center, scale, ord_ + 1, rank + 1, bim_pos,
           500e3, 0.2, 1)
    keep = np.fromfile(bk, dtype=np.int32)
    ref_keep = orc.bed_clumping(ob, infos_chr, bim_pos, thr_r2=0.2, size=500)
    np.testing.assert_array_equal(np.nonzero(keep == 1)[0], ref_keep)
    assert set(np.unique(keep)) <= {0, 1}


def test_one_backing_file_several_decode_tables_and_writebed(R, orc, golden_dir, missing_bed, tmp_path):
    """the reference attaches the same .bk with different code256 (G$copy(code = ...), R/impute.R:149-201,
    R/write-plink.R:35): every table must get its own device image"""
    ob = missing_bed
    g = orc.read_bed(ob, na_val=3).astype(np.uint8)               # what snp_readBed stores: 0 / 1 / 2 / 3 = missing
    rng = np.random.default_rng(1)
    imp = g.copy()
    miss = imp == 3
    imp[miss] = 4 + rng.integers(0, 3, size=int(miss.sum()))       # imputed calls are stored as codes 4, 5, 6
    bk = tmp_path / "geno.bk"
    np.asfortranarray(imp).ravel(order="F").tofile(bk)
    n, m = imp.shape
    ir, ic = np.arange(n), np.arange(m)

    def fbm(code):
        return R.env(code256=code, backingfile=str(bk), nrow=float(n), ncol=float(m))

    for code in (CODE_012, CODE_IMPUTE_PRED, CODE_DOSAGE, CODE_012):    # CODE_012 again: still its own image
        got = R.call("_bigsnpr_snp_colstats", fbm(code), ir + 1, ic + 1, 1)
        ref = orc.snp_colstats(orc.FBM256(imp, code), ir, ic)
        np.testing.assert_allclose(got["sumX"], ref["sumX"], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(got["denoX"], ref["denoX"], rtol=1e-9, atol=1e-9, equal_nan=True)
    # with CODE_012 the imputed codes are missing (NaN column sums), with CODE_IMPUTE_PRED they are calls
    a = R.call("_bigsnpr_snp_colstats", fbm(CODE_012), ir + 1, ic + 1, 1)["sumX"]
    b = R.call("_bigsnpr_snp_colstats", fbm(CODE_IMPUTE_PRED), ir + 1, ic + 1, 1)["sumX"]
    has_na = miss.any(axis=0)
    assert np.all(np.isnan(a[has_na])) and np.all(np.isfinite(b))
    # snp_writeBed: BM = G.round, code256 = replace(round(code), NA, 3): 3 MEANS missing
    g_round = np.where(np.isnan(CODE_IMPUTE_PRED), 3.0, np.round(CODE_IMPUTE_PRED))
    out = tmp_path / "out.bed"
    sel_r, sel_c = np.arange(0, n, 2), np.arange(5, m, 3)
    R.call("_bigsnpr_writebina", str(out), fbm(g_round), np.zeros(256, dtype=np.int32), sel_r + 1, sel_c + 1)
    raw = np.fromfile(out, dtype=np.uint8)
    assert list(raw[:3]) == [108, 27, 1]
    wb = orc.BedFile.from_payload(raw[3:], sel_r.size, sel_c.size)
    back = orc.read_bed(wb, na_val=3)
    expect = np.where(imp >= 4, imp - 4, imp)[np.ix_(sel_r, sel_c)]       # imputed calls written as calls
    np.testing.assert_array_equal(back, expect)
    # readbina2: the .bed sub-matrix into a (new) FBM backing file, one decoded byte per genotype
    obj = _bed_obj(R, os.path.join(golden_dir, "example-missing.bed"), ob.n, ob.m)
    bk2 = tmp_path / "read.bk"
    np.zeros(sel_r.size * sel_c.size, dtype=np.uint8).tofile(bk2)
    R.call("_bigsnpr_readbina2", R.env(backingfile=str(bk2)), obj, sel_r + 1, sel_c + 1, 1)
    np.testing.assert_array_equal(np.fromfile(bk2, dtype=np.uint8).reshape((sel_r.size, sel_c.size), order="F"),
                                  g[np.ix_(sel_r, sel_c)])
    # readbina (snp_readBed, R/read-plink.R:54): the WHOLE file through the caller's 4 x 256 raw table; TRUE at EOF
    src = os.path.join(golden_dir, "example-missing.bed")
    for tab in (orc.get_code(), np.random.default_rng(5).integers(0, 256, size=(4, 256)).astype(np.uint8)):
        bk3 = tmp_path / "readbina.bk"
        np.zeros(ob.n * ob.m, dtype=np.uint8).tofile(bk3)
        eof = R.call("_bigsnpr_readbina", src, R.env(backingfile=str(bk3), nrow=float(ob.n), ncol=float(ob.m)), tab)
        ref, ref_eof = orc.readbina(src, ob.n, ob.m, tab)
        np.testing.assert_array_equal(np.fromfile(bk3, dtype=np.uint8).reshape((ob.n, ob.m), order="F"), ref)
        assert bool(eof[0]) is True and ref_eof is True
    longer = tmp_path / "longer.bed"
    longer.write_bytes(open(src, "rb").read() + b"\x00")            # a trailing byte: EOF not reached (the warning case)
    eof = R.call("_bigsnpr_readbina", str(longer), R.env(backingfile=str(bk3), nrow=float(ob.n), ncol=float(ob.m)),
                 orc.get_code())
    assert bool(eof[0]) is False and orc.readbina(str(longer), ob.n, ob.m, orc.get_code())[1] is False
    with pytest.raises(Exception, match="4 x 256"):
        R.call("_bigsnpr_readbina", src, R.env(backingfile=str(bk3), nrow=float(ob.n), ncol=float(ob.m)),
               np.zeros((4, 3), dtype=np.uint8))
