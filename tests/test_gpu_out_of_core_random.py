"""Random shapes through the out-of-core handle (tools/fuzz_out_of_core.py: one synthetic matrix per draw written as a
.bed file, opened resident and — BSN_IMAGE_BUDGET — with slabs of 64 s variants): every entry point that walks the file
gives what the resident handle gives, on ragged sizes (n % 4 != 0, m % 64 != 0, a last slab of a few variants, windows
across slab borders, 1 - 3 chromosomes).  The reference maps a file of any size and every function works on it
(src/bed-acc.h:46, src/clumping-bed.cpp:11-91, src/read-plink.cpp:13-80).  BSN_TEST_SEED_OFFSET moves the draws."""
import os
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_OFF = 100 * int(os.environ.get("BSN_TEST_SEED_OFFSET", "0"))


@pytest.mark.parametrize("draw", range(6))
def test_streamed_handle_equals_resident_on_random_shapes(draw):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_out_of_core
    with tempfile.TemporaryDirectory() as tmp:
        bad = fuzz_out_of_core.one_draw(_OFF + draw, tmp)
    assert bad == [], "\n".join(bad)


def test_streamed_ld_and_clumping_on_bands_wide_enough_for_the_fp4_kernels():
    """the runs of a streamed handle reach the kernels of large bands too (k_pair_stats_f4<., RAW>: >= 1 024 blocks of
    128 x 32 variant pairs per run, no keep-mask when every sample is selected — the slab image's pad samples are code 0
    like the resident image's): LD scores, bed_cor and clumping with slabs of 8 192 variants and windows of ~ 600 equal
    the resident handle's, all samples and a row subset."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_out_of_core
    import bigsnpr_amd as ba
    from bigsnpr_amd import ld as ldm
    n, m = 203, 22003
    rng = np.random.default_rng(77)
    with tempfile.TemporaryDirectory() as tmp:
        gb = ba.bed.synthetic(n, m, seed=4242, na16=1500)
        path = os.path.join(tmp, "wide.bed")
        fuzz_out_of_core.write_bed(path, gb)
        gb.close()
        res = ba.bed(path)
        pitch = (n + 3) // 4 + 255 & ~255
        os.environ["BSN_IMAGE_BUDGET"] = str((8192 + 66) * pitch)
        try:
            ooc = ba.bed(path)
        finally:
            del os.environ["BSN_IMAGE_BUDGET"]
        assert ooc.streamed and not res.streamed
        pos = np.cumsum(rng.integers(1, 2000, size=m)).astype(np.float64)
        chrom = np.repeat([1, 2], [m // 2, m - m // 2])
        ir = np.sort(rng.choice(n, n - 30, replace=False))
        for kw in (dict(), dict(ind_row=ir)):
            a = ba.bed_ld_scores(ooc, size=600, infos_pos=pos, **kw)
            assert "RAW" in ldm.last_stats()["kernel"], ldm.last_stats()["kernel"]
            np.testing.assert_array_equal(a, ba.bed_ld_scores(res, size=600, infos_pos=pos, **kw))
            c1, c0 = ba.bed_cor(ooc, size=600, infos_pos=pos, alpha=0.3, **kw), ba.bed_cor(res, size=600, infos_pos=pos, alpha=0.3, **kw)
            np.testing.assert_array_equal(c1.p, c0.p), np.testing.assert_array_equal(c1.i, c0.i), np.testing.assert_array_equal(c1.x, c0.x)
            np.testing.assert_array_equal(ba.bed_clumping(ooc, thr_r2=0.05, size=600, infos_chr=chrom, infos_pos=pos, **kw),
                                          ba.bed_clumping(res, thr_r2=0.05, size=600, infos_chr=chrom, infos_pos=pos, **kw))
        ooc.close(), res.close()
