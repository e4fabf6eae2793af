"""bindings/R/bigsnprhip/src/bigsnpr_hip_shim.c through a compiler (no R in this image): strict warnings against declarations of
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


def test_r_package_files_agree_with_the_shim_and_the_reference(R):
    """bindings/R/bigsnprhip is an installable package — DESCRIPTION, NAMESPACE (useDynLib with registration, cf.
    the reference's NAMESPACE:124), src/Makevars (link line), R/zzz.R (hip_enable / hip_disable, the six-line bed_randomSVD, the
    operator closures).  R is absent here, so the files are checked as text: the symbol list hip_enable() swaps == the
    reference-named entries of the shim's registration table == the hot-path subset of the reference's own table (same
    arities); every .Call in zzz.R names a registered routine and passes as many arguments as it is registered with."""
    import re
    pkg = os.path.join(ROOT, "bindings", "R", "bigsnprhip")
    for f in ("DESCRIPTION", "NAMESPACE", os.path.join("src", "Makevars"), os.path.join("src", "bigsnpr_hip_shim.c"),
              os.path.join("R", "zzz.R")):
        assert os.path.isfile(os.path.join(pkg, f)), f
    desc = open(os.path.join(pkg, "DESCRIPTION")).read()
    assert re.search(r"^Package: bigsnprhip$", desc, re.M) and "NeedsCompilation: yes" in desc
    ns = open(os.path.join(pkg, "NAMESPACE")).read()
    assert re.search(r"^useDynLib\(bigsnprhip, \.registration = TRUE\)$", ns, re.M)
    shim = open(os.path.join(pkg, "src", "bigsnpr_hip_shim.c")).read()
    assert "void R_init_bigsnprhip(DllInfo *dll)" in shim          # the name useDynLib(bigsnprhip) looks for
    mk = open(os.path.join(pkg, "src", "Makevars")).read()
    assert "-lbigsnpr_hip" in mk and "/include" in mk and "-Wl,-rpath," in mk
    zzz = open(os.path.join(pkg, "R", "zzz.R")).read()
    body = zzz[zzz.index("hip_symbols <- function()"):]
    listed = re.findall(r'"(_bigsnpr_\w+)"', body[:body.index("}")])
    got = R.routines()
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_call_entries.json")))["entries"]
    registered_ref_named = sorted(n for n in got if not n.endswith("_hip"))
    assert sorted(listed) == registered_ref_named == sorted(HOT_PATH)
    for n in listed:
        assert got[n] == ref[n], n
    # the commented symbol list of NAMESPACE (documentation of what .registration creates) is the whole table
    assert sorted(set(re.findall(r"`(_bigsnpr_\w+)`", ns))) == sorted(got)
    # every .Call of the R glue: a registered routine, called with its registered number of arguments
    calls = re.findall(r"\.Call\(`(_bigsnpr_\w+)`((?:[^()]|\([^()]*\))*)\)", zzz)
    assert {c[0] for c in calls} == {n for n in got if n.endswith("_hip")}
    for name, args in calls:
        nargs = 0 if not args.strip(", \n") else len([a for a in args.strip(", \n").split(",")])
        assert nargs == got[name], (name, nargs, got[name])
    # exports of NAMESPACE are defined in zzz.R
    for ex in re.findall(r"^export\((\w+)\)$", ns, re.M):
        assert re.search(r"^%s <- function" % ex, zzz, re.M), ex
