"""bindings/R/bigsnpr_hip_shim.c through a compiler (no R in this image): strict warnings against declarations of
the R API it uses (tests/rstub/include, test infrastructure), its registration table against the reference's
(tests/golden/reference_call_entries.json, made by tools/make_call_entries_fixture.py from
src/RcppExports.cpp:597-640), and the entry points that need no GPU run against the stand-in runtime."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "rstub"))

HOT_PATH = ["_bigsnpr_bedXPtr", "_bigsnpr_bed_pMatVec4", "_bigsnpr_bed_cpMatVec4", "_bigsnpr_bed_colstats",
            "_bigsnpr_bed_col_counts_cpp", "_bigsnpr_bed_row_counts_cpp", "_bigsnpr_read_bed", "_bigsnpr_read_bed_scaled",
            "_bigsnpr_snp_colstats", "_bigsnpr_corMat", "_bigsnpr_ld_scores", "_bigsnpr_clumping_chr",
            "_bigsnpr_bed_clumping_chr", "_bigsnpr_clumping_chr_cached", "_bigsnpr_prod_and_rowSumsSq",
            "_bigsnpr_prod_and_rowSumsSq2", "_bigsnpr_multLinReg", "_bigsnpr_readbina", "_bigsnpr_readbina2",
            "_bigsnpr_writebina"]


def test_shim_compiles_without_warnings():
    import build_rstub
    r = build_rstub.syntax_check()
    assert r.returncode == 0, r.stderr[-4000:]
    assert r.stderr.strip() == ""


@pytest.fixture(scope="module")
def R():
    import rshim
    return rshim.R()


def test_registration_table_matches_the_reference(R):
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_call_entries.json")))["entries"]
    got = R.routines()
    for name in HOT_PATH:   # SURVEY.md 8(b) + 8(f): same symbol, same arity
        assert name in ref, name
        assert got.get(name) == ref[name], (name, got.get(name), ref[name])
    for name, nargs in got.items():   # everything else the shim registers is marked as an addition
        assert name in ref or name.endswith("_hip"), name


def test_argument_checks_run_without_a_gpu(R):
    import rshim
    obj = R.env(address=None)
    one = np.arange(1, 6, dtype=np.int32)
    # myassert_size of the reference (asserted by tests/testthat/test-5-bed-prod-vec.R through the R error text)
    with pytest.raises(rshim.RError, match="Incompatibility between dimensions"):
        R.call("_bigsnpr_bed_pMatVec4", obj, one, one, np.zeros(5), np.ones(5), np.zeros(4), 1)
    with pytest.raises(rshim.RError, match="Incompatibility between dimensions"):
        R.call("_bigsnpr_bed_cpMatVec4", obj, one, one, np.zeros(5), np.ones(4), np.zeros(5), 1)
    # wrong arity and unknown routine are caught by the registered table, as .Call does
    with pytest.raises(rshim.RError, match="takes 7 arguments"):
        R.call("_bigsnpr_bed_pMatVec4", obj, one, one)
    with pytest.raises(rshim.RError, match="no routine"):
        R.call("_bigsnpr_nothing")
    # an FBM object whose code256 is not 256 doubles; a .bed that does not exist (the library's message comes through)
    fbm = R.env(code256=np.zeros(3), backingfile="/nonexistent.bk", nrow=2.0, ncol=2.0)
    with pytest.raises(rshim.RError, match="code256"):
        R.call("_bigsnpr_snp_colstats", fbm, one, one, 1)
    with pytest.raises(rshim.RError):
        R.call("_bigsnpr_bedXPtr", "/nonexistent/file.bed", 10, 10)
    R.reset()
