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
