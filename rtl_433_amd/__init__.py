"""rtl_433_amd -- MI355X (gfx950) implementation of rtl_433's IQ -> decoded-event hot path.

Only the hot path lives here: HIP kernels + the C ABI (csrc/, include/r433_hip.h) and a thin
host-side mirror of the reference's offline flow (engine.py).  Importing the package does not load
the shared library; `rtl_433_amd._lib.lib()` does and fails loudly if it is missing.
"""
__version__ = "0.1.0"
