"""README.md's table of environment switches against the source: every getenv("BSN_*") the PRODUCT library
can reach (outside #ifdef BSN_ABLATION, not through abl_getenv) has a row, every switch in the table is read somewhere,
and the profiling-build switches are named as such."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _source_knobs():
    product, profiling = set(), set()
    for f in sorted(glob.glob(os.path.join(ROOT, "bigsnpr_amd", "csrc", "*.h*"))):
        stack = []
        for line in open(f):
            t = line.strip()
            if t.startswith("#if"):
                stack.append("BSN_ABLATION" in t and not t.startswith("#ifndef"))
            elif t.startswith("#else") and stack:
                stack[-1] = False if stack[-1] else stack[-1]
            elif t.startswith("#endif") and stack:
                stack.pop()
            for m in re.finditer(r'(abl_)?getenv\("(BSN_[A-Z0-9_]+)"\)', line):
                (profiling if (m.group(1) or any(stack)) else product).add(m.group(2))
    return product, profiling


def test_every_switch_of_the_library_is_documented():
    product, profiling = _source_knobs()
    readme = open(os.path.join(ROOT, "README.md")).read()
    sec = readme[readme.index("## Environment switches of the library"):]
    table = sec[:sec.index("Python side:")]
    tail = sec[sec.index("Python side:"):]
    listed = set(re.findall(r"`(BSN_[A-Z0-9_]+)", table))
    assert product <= listed, sorted(product - listed)
    assert listed <= product, sorted(listed - product)          # nothing stale in the table
    prof_listed = set(re.findall(r"`(BSN_[A-Z0-9_]+)`", tail[tail.index("Profiling build only"):]))
    assert profiling - product <= prof_listed, sorted(profiling - product - prof_listed)
    # every test file a row names exists
    for name in set(re.findall(r"`(test_\w+\.py)`", table)):
        assert os.path.isfile(os.path.join(ROOT, "tests", name)), name
    for name in set(re.findall(r"`(tools/\w+\.py)`", table)):
        assert os.path.isfile(os.path.join(ROOT, name)), name
