"""snp_readBed / snp_writeBed round trips (tests/testthat/test-1-readBed.R:91-115,
test-1-writeBed.R:37-76): decode == oracle, write -> read identity incl. random codes 0..3
and row / column subsets."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import bigsnpr_amd
    return bigsnpr_amd


@pytest.mark.parametrize("name", ["example.bed", "example-missing.bed"])
def test_read_bed_equals_oracle_decode(ba, orc, golden_dir, name):
    path = os.path.join(golden_dir, name)
    x = ba.snp_readBed(path)
    ob = orc.BedFile(path)
    np.testing.assert_array_equal(x["bytes"], orc.read_bed(ob, na_val=3).astype(np.uint8))
    assert len(x["fam"]["sample.ID"]) == ob.n and len(x["map"]["marker.ID"]) == ob.m
    # `[` accessor of the bed object agrees (test-1-readBed.R:91-115)
    gb = ba.bed(path)
    sub = gb[np.arange(5, 50), 10:30]
    np.testing.assert_array_equal(np.where(sub < 0, 3, sub), x["bytes"][5:50, 10:30])


def test_write_read_roundtrip(ba, orc, golden_dir, tmp_path):
    path = os.path.join(golden_dir, "example-missing.bed")
    x = ba.snp_readBed(path)
    out = str(tmp_path / "copy.bed")
    ba.snp_writeBed(x, out)
    assert open(out, "rb").read() == open(path, "rb").read()          # byte-identical .bed
    with pytest.raises(FileExistsError):
        ba.snp_writeBed(x, out)
    # random codes 0..3 and a subset (n not a multiple of 4)
    rng = np.random.default_rng(0)
    n, m = 203, 57
    g = rng.integers(0, 4, size=(n, m)).astype(np.uint8)
    fake = dict(genotypes=ba.FBM_code256(g),
                fam={k: [str(i) for i in range(n)] for k in ("family.ID", "sample.ID", "paternal.ID",
                                                              "maternal.ID", "sex", "affection")},
                map={"chromosome": ["1"] * m, "marker.ID": ["snp%d" % j for j in range(m)],
                     "genetic.dist": ["0"] * m, "physical.pos": [str(1000 * j) for j in range(m)],
                     "allele1": ["A"] * m, "allele2": ["T"] * m})
    ir = np.sort(rng.choice(n, 101, replace=False)); ic = rng.choice(m, 20, replace=False)
    out2 = str(tmp_path / "sub.bed")
    ba.snp_writeBed(fake, out2, ind_row=ir, ind_col=ic)
    y = ba.snp_readBed(out2)
    np.testing.assert_array_equal(y["bytes"], g[np.ix_(ir, ic)])
    assert y["map"]["marker.ID"] == ["snp%d" % j for j in ic]
    # the written payload equals the oracle's packing of the same sub-matrix
    ob = orc.BedFile(out2)
    np.testing.assert_array_equal(orc.read_bed(ob, na_val=3), g[np.ix_(ir, ic)])


def test_readbina_with_the_reference_table_and_a_random_one(ba, orc, golden_dir):
    """readbina (src/read-plink.cpp:13-56) on the device: any 4 x 256 raw table, whole file, EOF flag."""
    from bigsnpr_amd import plink_io
    np.testing.assert_array_equal(plink_io.get_code(), orc.get_code())
    for name in ("example.bed", "example-missing.bed"):
        path = os.path.join(golden_dir, name)
        ob = orc.BedFile(path)
        for tab in (None, np.random.default_rng(1).integers(0, 256, size=(4, 256)).astype(np.uint8)):
            got, eof = plink_io.readbina(path, tab)
            ref, ref_eof = orc.readbina(path, ob.n, ob.m, orc.get_code() if tab is None else tab)
            np.testing.assert_array_equal(got, ref)
            assert eof == ref_eof
