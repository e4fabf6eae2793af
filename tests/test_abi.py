"""CPU-only checks of the drop-in boundary: the shared library builds, loads, and exports
every symbol that include/bigsnpr_hip.h declares; compute calls fail loudly without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from bigsnpr_amd import build
    build.build()
    from bigsnpr_amd import _lib
    return _lib


def test_header_symbols_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "bigsnpr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bsn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 25
    L = lib.load()
    for name in declared:
        assert hasattr(L, name), "missing export " + name
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)


def test_no_torch_or_r_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "bigsnpr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for banned in ("SEXP", "Rcpp", "at::", "torch", "hipStream_t", "std::"):
        assert banned not in hdr


def test_product_never_imports_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "bigsnpr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", src, flags=re.M), f
                assert not re.search(r"#\s*include.*oracle", src), f
                assert "libbsn_oracle" not in src and "orc_" not in src.replace("orc_fake_bed", ""), f


def test_fails_loudly_without_gpu(lib, golden_dir):
    import bigsnpr_amd as ba
    if ba.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ba.BsnError, match="no CPU fallback"):
        ba.bed(os.path.join(golden_dir, "example.bed"))
    with pytest.raises(ba.BsnError, match="no CPU fallback"):
        ba.selftest()


def test_open_error_strings_match_reference(lib, golden_dir, tmp_path):
    """src/bed-acc-xptr.cpp:19-34 messages; these checks run before any GPU work."""
    import ctypes as C
    L = lib.load()
    raw = np.fromfile(os.path.join(golden_dir, "example-missing.bed"), dtype=np.uint8)

    def err(path, n, m):
        h = C.c_void_p()
        rc = L.bsn_bed_open(str(path).encode(), n, m, C.byref(h))
        assert rc != 0
        return L.bsn_last_error().decode()

    assert "Error when mapping file" in err(tmp_path / "nope.bed", 200, 500)
    p = tmp_path / "a.bed"
    bad = raw.copy(); bad[1] = 0; bad.tofile(p)
    assert err(p, 200, 500) == "File is not a binary PED file."
    bad = raw.copy(); bad[2] = 0; bad.tofile(p)
    assert err(p, 200, 500) == "Variant-major is the only mode supported."
    raw.tofile(p)
    assert err(p, 196, 500) == "n or p does not match the dimensions of the file."


def test_host_api_argument_checks(lib):
    import bigsnpr_amd as ba
    with pytest.raises(TypeError, match="is not of class 'bed' or 'bed_light'"):
        ba.bed_prodVec(object(), np.zeros(3))
    with pytest.raises(ValueError, match="must have 'bed' extension"):
        ba.bed("foo.txt")
    # sub_bed (R/bed-class.R:20-32, examples in its documentation)
    assert ba.sub_bed("toto.bed") == "toto" and ba.sub_bed("toto.bed", ".bim") == "toto.bim"
    assert ba.sub_bed("toto.bed", "_QC", stop_if_not_ext=False) == "toto_QC"
    with pytest.raises(ValueError, match="must have 'bed' extension"):
        ba.sub_bed("toto.txt")
    with pytest.raises(ValueError, match="Replacement must be an extension starting with '.'"):
        ba.sub_bed("toto.bed", "_QC")


def test_ctypes_signatures_match_the_header_prototypes(lib):
    """every prototype of include/bigsnpr_hip.h against the argument table the host mirror binds it with
    (bigsnpr_amd/_lib.py SIGNATURES): the same number of arguments and, argument by argument, the same machine class —
    pointer, 32-bit integer, 64-bit integer, double.  A 64-bit count bound as c_int32 works until the first large
    matrix."""
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "bigsnpr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)

    def c_class(decl):
        decl = decl.strip()
        if "*" in decl or re.search(r"\bbsn_[a-z_]+_fn\b", decl) or "[" in decl:
            return "ptr"
        base = re.sub(r"\b(const|unsigned|signed|struct)\b", "", decl).split()
        t = base[0] if base else ""
        if decl.startswith("unsigned") and t in ("", "int"):
            return "i32"
        return {"double": "f64", "float": "f32", "int64_t": "i64", "uint64_t": "i64", "size_t": "i64", "long": "i64",
                "int32_t": "i32", "uint32_t": "i32", "int": "i32", "uint8_t": "i8", "char": "i8"}[t]

    def py_class(t):
        if t is None:
            return "void"
        if isinstance(t, type) and (issubclass(t, (C._Pointer, C.c_void_p, C.c_char_p, C._CFuncPtr)) or hasattr(t, "contents")):
            return "ptr"
        return {C.c_double: "f64", C.c_float: "f32", C.c_int64: "i64", C.c_uint64: "i64", C.c_size_t: "i64",
                C.c_long: "i64", C.c_longlong: "i64", C.c_ulonglong: "i64", C.c_int32: "i32", C.c_uint32: "i32",
                C.c_int: "i32", C.c_uint: "i32"}[t]

    protos = re.findall(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(bsn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr)
    seen = 0
    for ret, name, args in protos:
        if name not in lib.SIGNATURES or "typedef" in ret:
            continue
        res, argtypes = lib.SIGNATURES[name]
        args = args.strip()
        c_args = [] if args in ("", "void") else [c_class(a) for a in args.split(",")]
        p_args = [py_class(t) for t in argtypes]
        assert c_args == p_args, (name, c_args, p_args)
        want = "ptr" if "*" in ret else ("void" if ret.split()[-1] == "void" else c_class(ret.replace("extern", "")))
        assert py_class(res) == want, (name, ret, res)
        seen += 1
    assert seen == len(lib.SIGNATURES), (seen, len(lib.SIGNATURES))
