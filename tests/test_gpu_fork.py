"""SURVEY.md §8b "Threading": the library must be safe to load before fork() — bigsnpr's R layer
runs `foreach` workers that re-open their own handle (R/bed-class.R:187-192).  The check runs in
a fresh interpreter: load the library, fork two workers that each open an image and compute,
then compute in the parent."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_load_then_fork_then_compute():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_fork.py")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "child 0 ok" in out and "child 1 ok" in out and "parent ok True" in out, out
