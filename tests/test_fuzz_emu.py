"""A seeded slice of tools/fuzz_emu.py in the regular CPU suite: random signals (spurious pulses, >1200-pulse
packages, FSK ring overflow, saturation, noise of every width) x random flow options (levels, filters, frame
sizes, autolevel, split captures with quiet and blind cuts), kernel sources on the wave emulator vs the oracle.
Seed 1189 is the case that exposed the FSK candidate surviving into the second pulse of a package."""
import os
import sys

import pytest

from tests.emu import build_emu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")


@pytest.mark.parametrize("seed", [3, 11, 19, 27, 42, 58, 77, 93, 104, 1189])
def test_fuzz_case(seed):
    import fuzz_emu
    assert fuzz_emu.one_case(seed) is None
