#!/usr/bin/env python3
"""Writes tests/golden/reference_call_entries.json: the (symbol, arity) pairs of the reference's `.Call`
registration table (src/RcppExports.cpp, `static const R_CallMethodDef CallEntries[]`).  Run in the build
container, where /root/reference exists; the fixture is data (names and numbers) and travels with the repo."""
import json, os, re, sys
src = open("/root/reference/src/RcppExports.cpp").read()
tab = src[src.index("static const R_CallMethodDef CallEntries[]"):]
tab = tab[:tab.index("};")]
ent = re.findall(r'\{"(_bigsnpr_\w+)",\s*\(DL_FUNC\)\s*&\w+,\s*(\d+)\}', tab)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_call_entries.json")
json.dump({"source": "bigsnpr src/RcppExports.cpp CallEntries", "entries": {n: int(a) for n, a in ent}}, open(out, "w"), indent=1, sort_keys=True)
print(len(ent), "entries ->", out)
