"""The two structures that cross the C ABI by value of their layout (bsn_svd_options, bsn_svd_info, include/bigsnpr_hip.h)
against their ctypes mirrors (bigsnpr_amd/_lib.py): same fields in the same order, same offsets, same size — as a C
compiler lays out the header.  Fields are appended to these structures round by round; a mirror that lags reads the
wrong words without any error."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fields(header, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):        # "int32_t hook_rank, hook_world", "double *center_out, *scale_out"
            m = re.search(r"([A-Za-z_][A-Za-z_0-9]*)\s*(\[\d+\])?$", part.strip())
            assert m, decl
            out.append(m.group(1))
    return out


def test_ctypes_mirrors_have_the_layout_of_the_header(tmp_path):
    from bigsnpr_amd import _lib
    header = open(os.path.join(ROOT, "include", "bigsnpr_hip.h")).read()
    pairs = (("bsn_svd_options", _lib.SvdOptions), ("bsn_svd_info", _lib.SvdInfo))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "bigsnpr_hip.h"', "int main(void) {"]
    for cname, _ in pairs:
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in _fields(header, cname):
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, f, v = ln.split()
        got.setdefault(cname, []).append((f, int(v)))
    for cname, cls in pairs:
        rows = got[cname]
        assert rows[0] == ("sizeof", C.sizeof(cls)), (cname, rows[0], C.sizeof(cls))
        assert [f for f, _ in rows[1:]] == [f[0] for f in cls._fields_], cname
        for f, off in rows[1:]:
            assert getattr(cls, f).offset == off, (cname, f, getattr(cls, f).offset, off)
