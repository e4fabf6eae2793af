#!/usr/bin/env python3
"""Random shapes through the OUT-OF-CORE handle (round 6, second session): a synthetic matrix is written as a .bed file,
opened once resident and once with BSN_IMAGE_BUDGET forcing slabs of 64 s variants, and every entry point that walks the
file is compared with the resident handle — counts, column statistics, both products, LD scores, bed_cor, clumping
(1 - 3 chromosomes, statistics, exclusions, row subsets), the partial SVD on all variants and on a list in file order,
bed_tcrossprodSelf and the byte conversions.  Ragged sizes on purpose: n % 4 != 0, m % 64 != 0, a last slab of one
variant, windows that reach across one or several slab borders.

usage (GPU box): python tools/fuzz_out_of_core.py <first seed> <draws>
Prints one line per draw and the parameters of every mismatch; exit code = number of draws with a mismatch."""
import os
import sys
import tempfile
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bigsnpr_amd as ba                      # noqa: E402
from bigsnpr_amd import plink_io              # noqa: E402


def write_bed(path, gb):
    with open(path, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(gb.download().tobytes())
    with open(path[:-4] + ".fam", "w") as f:
        f.write("".join("f%d i%d 0 0 0 -9\n" % (i, i) for i in range(gb.nrow)))
    with open(path[:-4] + ".bim", "w") as f:
        f.write("".join("1\trs%d\t0\t%d\tA\tC\n" % (j, 1000 * (j + 1)) for j in range(gb.ncol)))


def one_draw(seed, tmp):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([5, 17, 63, 130, 259, 517, 1030, 2051])) + int(rng.integers(0, 4))
    m = int(rng.choice([65, 70, 127, 129, 191, 193, 320, 500, 1000, 1500])) + int(rng.integers(0, 3))
    s = int(rng.choice([1, 1, 2, 3, 4, 7]))
    while s > 1 and 64 * s + 66 > m + 63:      # (a budget that holds the whole file gives a resident handle)
        s -= 1
    na16 = int(rng.choice([0, 655, 6000]))
    desc = "seed %d: n=%d m=%d slabs of %d, na16=%d" % (seed, n, m, 64 * s, na16)
    gb = ba.bed.synthetic(n, m, seed=1000 + seed, na16=na16)
    path = os.path.join(tmp, "f%d.bed" % seed)
    write_bed(path, gb)
    gb.close()
    res = ba.bed(path)
    pitch = (n + 3) // 4 + 255 & ~255
    os.environ["BSN_IMAGE_BUDGET"] = str(min(64 * s + 66, m + 63) * pitch)
    try:
        ooc = ba.bed(path)
    finally:
        del os.environ["BSN_IMAGE_BUDGET"]
    assert ooc.streamed and not res.streamed, "handle kinds"
    bad = []

    def check(name, fn):
        try:
            fn()
        except ba.BsnError as e:
            msg = str(e)
            if "slab image" in msg or "more than the" in msg:
                return                      # a window that does not fit the slab: named refusal
            bad.append("%s: BsnError %s" % (name, msg[:200]))
        except AssertionError as e:
            bad.append("%s: %s" % (name, str(e)[:300].replace("\n", " ")))
        except Exception:
            bad.append("%s: %s" % (name, traceback.format_exc()[-400:].replace("\n", " | ")))

    def pick_rows():
        t = int(rng.integers(0, 3))
        if t == 0 or n < 8:
            return None
        if t == 1:
            return np.sort(rng.choice(n, max(4, int(n * rng.uniform(0.3, 0.95))), replace=False))
        return rng.integers(0, n, size=max(4, int(n * rng.uniform(0.5, 1.2))))

    def pick_cols(sorted_only=False):
        t = int(rng.integers(0, 2 if sorted_only else 3))
        if t == 0:
            return None
        if t == 1:
            return np.sort(rng.choice(m, max(2, int(m * rng.uniform(0.2, 0.9))), replace=False))
        return rng.permutation(m)[: max(2, m // 2)]

    eq = np.testing.assert_array_equal
    for rep in range(2):
        ir, ic = pick_rows(), pick_cols()
        kw = dict(ind_row=ir, ind_col=ic)
        check("counts", lambda: eq(ba.bed_counts(ooc, **kw), ba.bed_counts(res, **kw)))

        def stats():
            a, b = ba.bed_colstats(ooc, **kw), ba.bed_colstats(res, **kw)
            for f in ("sumX", "denoX", "nb_nona_col"):
                eq(a[f], b[f])
        check("colstats", stats)
        nr = n if ir is None else len(ir)
        nc = m if ic is None else len(ic)
        ce, sa = rng.normal(size=nc), rng.uniform(0.5, 2, size=nc)
        y, x = rng.normal(size=nr), rng.normal(size=nc)
        repeats = ir is not None and np.unique(ir).size < ir.size

        def cprod():
            z1, z0 = ba.bed_cprodVec(ooc, y, ir, ic, ce, sa), ba.bed_cprodVec(res, y, ir, ic, ce, sa)
            if repeats:    # (the values of a repeated sample are added up in an order of the device's choosing before they are rounded to digits)
                assert np.abs(z1 - z0).max() <= 1e-12 * max(np.abs(z0).max(), 1e-300), "cprodVec %g" % np.abs(z1 - z0).max()
            else:
                eq(z1, z0)
        check("cprodVec (rows %s, columns %s)" % ("all" if ir is None else "with repeats" if repeats else "sorted subset",
                                                  "all" if ic is None else "sorted" if (np.diff(ic) > 0).all() else "permuted"), cprod)

        def prod():
            p1, p0 = ba.bed_prodVec(ooc, x, ir, ic, ce, sa), ba.bed_prodVec(res, x, ir, ic, ce, sa)
            assert np.abs(p1 - p0).max() <= 1e-12 * max(np.abs(p0).max(), 1e-300), "prodVec %g" % np.abs(p1 - p0).max()
        check("prodVec", prod)
        rr = np.arange(n) if ir is None else ir
        cc = np.arange(m) if ic is None else ic
        check("accessor", lambda: eq(ooc[rr[:7], cc[:9]], res[rr[:7], cc[:9]]))
        check("bed_to_bytes", lambda: eq(plink_io.bed_to_bytes(ooc, ir, ic), plink_io.bed_to_bytes(res, ir, ic)))
    # LD: positions with random spacing, windows of a few to ~ half a slab of variants (and sometimes more: refusal or wide path)
    posv = np.cumsum(rng.integers(1, 2000, size=m)).astype(np.float64)
    for rep in range(2):
        ic = pick_cols(sorted_only=True)
        pv = posv if ic is None else posv[ic]
        ir = pick_rows() if n >= 30 else None
        size_kb = float(rng.choice([3, 10, 25, 40])) * (1 if rep == 0 else float(rng.choice([1, 2])))
        check("ld_scores size=%g" % size_kb, lambda: eq(ba.bed_ld_scores(ooc, ind_row=ir, ind_col=ic, size=size_kb, infos_pos=pv),
                                                        ba.bed_ld_scores(res, ind_row=ir, ind_col=ic, size=size_kb, infos_pos=pv)))
        ckw = [dict(size=size_kb), dict(size=size_kb, alpha=0.3), dict(size=size_kb, thr_r2=0.05, fill_diag=False)][int(rng.integers(0, 3))]

        def cor():
            c1, c0 = ba.bed_cor(ooc, ind_row=ir, ind_col=ic, infos_pos=pv, **ckw), ba.bed_cor(res, ind_row=ir, ind_col=ic, infos_pos=pv, **ckw)
            eq(c1.p, c0.p), eq(c1.i, c0.i), eq(c1.x, c0.x)
        check("bed_cor %s" % ckw, cor)
    nchr = int(rng.integers(1, 4))
    chrs = np.sort(rng.integers(1, nchr + 1, size=m))
    for rep in range(3):
        kw = dict(thr_r2=float(rng.choice([0.05, 0.2, 0.5])), size=float(rng.choice([5, 20, 45, 120])))
        if rng.random() < 0.5:
            kw["S"] = rng.random(m) if rng.random() < 0.7 else np.round(rng.random(m) * 4)     # (ties)
        if rng.random() < 0.4:
            kw["exclude"] = np.sort(rng.choice(m, m // 7, replace=False))
        if rng.random() < 0.4 and n >= 30:
            kw["ind_row"] = pick_rows()
        check("clumping %s" % {k: (v if np.isscalar(v) else "...") for k, v in kw.items()},
              lambda: eq(ba.bed_clumping(ooc, infos_pos=posv, infos_chr=chrs, **kw), ba.bed_clumping(res, infos_pos=posv, infos_chr=chrs, **kw)))
    sc = ba.bed_scaleBinom(res)["scale"]
    if n >= 40 and (sc > 0).all():
        for ic in (None, np.sort(rng.choice(m, max(8, m // 2), replace=False))):
            def svd():
                a, b = ba.bed_randomSVD(ooc, ind_col=ic, k=3, tol=1e-10, slices=7), ba.bed_randomSVD(res, ind_col=ic, k=3, tol=1e-10, slices=7)
                assert a["out_of_core"] and a["converged"] and b["converged"], "converged"
                eq(a["center"], b["center"]), eq(a["scale"], b["scale"])
                np.testing.assert_allclose(a["d"], b["d"], rtol=1e-11)
                sg = np.sign(np.sum(a["v"] * b["v"], axis=0))
                assert np.abs(a["v"] * sg - b["v"]).max() < 1e-8, "v %g" % np.abs(a["v"] * sg - b["v"]).max()
            check("randomSVD list=%s" % (ic is not None), svd)

        def svd_default():
            a, b = ba.bed_randomSVD(ooc, k=3), ba.bed_randomSVD(res, k=3)
            assert a["converged"] and b["converged"], "converged (default)"
            np.testing.assert_allclose(a["d"], b["d"], rtol=1e-6)
        check("randomSVD default", svd_default)
        if n <= 600:
            def tc():
                (K1, a1), (K0, a0) = ba.bed_tcrossprodSelf(ooc), ba.bed_tcrossprodSelf(res)
                assert np.abs(K1 - K0).max() <= 1e-11 * np.abs(K0).max(), "K"
            check("tcrossprodSelf", tc)
    ooc.close(), res.close()
    for ext in (".bed", ".bim", ".fam"):
        os.unlink(path[:-4] + ext)
    print(desc, "OK" if not bad else "MISMATCH", flush=True)
    for b in bad:
        print("    " + b, flush=True)
    return bad


def main():
    s0, cnt = int(sys.argv[1]), int(sys.argv[2])
    nbad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(s0, s0 + cnt):
            try:
                nbad += 1 if one_draw(seed, tmp) else 0
            except Exception:
                nbad += 1
                print("seed %d: the draw itself failed: %s" % (seed, traceback.format_exc()[-600:].replace("\n", " | ")), flush=True)
    print("draws with a mismatch: %d of %d" % (nbad, cnt))
    return min(nbad, 100)


if __name__ == "__main__":
    sys.exit(main())
