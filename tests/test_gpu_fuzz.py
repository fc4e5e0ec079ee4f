"""Randomised GPU-vs-oracle parity (tools/fuzz_parity.py): random sizes 1..139 x 1..69, content
classes (photo, noise, two-colour checkers, flat, ramps), alpha patterns, qualities 0..4, both
colour spaces, random colour masks.  Byte equality for every case."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt,cases", [("BC7", 60), ("BC1_RGBA", 20), ("BC3", 20), ("ETC2_R8G8B8A8", 25),
                                       ("ETC2_R8G8B8A1", 20), ("ASTC_6x6", 20), ("ASTC_5x4", 15)])
def test_random_surfaces_match_the_oracle(fmt, cases):
    assert fuzz_parity.run(fmt, cases, seed=len(fmt)*1000 + cases) == 0
