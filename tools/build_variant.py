"""Development aid: a variant of the library with extra compiler flags, for A/B timing on the GPU box (tools/variant_bench.py).
    python tools/build_variant.py <name> [-DFLAG=VALUE ...]   ->   rtl_433_amd/lib/librtl433hip_<name>.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtl_433_amd import build
print(build._build_variant(f"librtl433hip_{sys.argv[1]}.so", sys.argv[2:], False))
