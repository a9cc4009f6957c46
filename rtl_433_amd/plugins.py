"""The reference's protocol decoders as plugins for hosts that are not the rtl_433 CLI (dropin/_build/libr433plugins.so:
the reference's sources compiled where they lie + dropin/plugins_shim.c).  They are the consumers of the hot path --
unchanged host C behind r_device.decode_fn -- and what they report comes back as JSON lines printed by the reference's
own data_print_jsons.  bench.py's multi-GPU run (configs[3]) gathers exactly those lines on rank 0."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "..", "dropin", "_build", "libr433plugins.so")


def available():
    return os.path.exists(LIB_PATH)


class Plugins:
    def __init__(self, flex=None):
        """flex: `-X` specs (list of str) registered behind the default decoders, as the CLI does (src/rtl_433.c:847-851)"""
        if not available():
            raise RuntimeError(f"{LIB_PATH} is missing: `make -C dropin plugins` (needs the reference tree once)")
        L = C.CDLL(os.path.abspath(LIB_PATH))
        L.r433p_create_with.restype = C.c_void_p
        L.r433p_create_with.argtypes = [C.c_char_p]
        L.r433p_devices.restype = C.c_int
        L.r433p_devices.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.r433p_take.restype = C.c_size_t
        L.r433p_take.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_ulong)]
        L.r433p_destroy.argtypes = [C.c_void_p]
        self.L = L
        self.h = L.r433p_create_with("\n".join(flex).encode() if flex else None)
        n = L.r433p_devices(self.h, None, 0)
        self.devices = (C.c_void_p * n)()
        assert L.r433p_devices(self.h, self.devices, n) == n

    def stateless(self):
        """-> ctypes uint8 array for r433_batch_set_stateless: what the plugin library says about its own decoders (the ONE
        place that knows: dropin/plugins_shim.c r433p_stateless; it answers -1 when a decoder it lists as keeping state is
        not among the registered ones -- a renamed decoder must not silently become "stateless")"""
        flags = (C.c_uint8 * len(self.devices))()
        self.L.r433p_stateless.restype = C.c_int
        self.L.r433p_stateless.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        n = self.L.r433p_stateless(self.h, flags, len(flags))
        if n != len(flags):
            raise RuntimeError(f"r433p_stateless answered {n} for {len(flags)} decoders: its list of stateful decoders no longer matches the registered set")
        return flags

    def helper_probe(self):
        """-> address of r433_host_helper_probe (dropin/helper_wrap.c: the decoders' calls of bitbuffer_invert / _search /
        _find_repeated_* go through wrappers that answer the pre-filter's questions without the payload), for
        BatchEngine.probe_prefilter(..., helper=...); None for a plugin library built without the wrappers"""
        fn = getattr(self.L, "r433_host_helper_probe", None)
        return C.cast(fn, C.c_void_p).value if fn is not None else None

    def hooks(self):
        """-> r433_dispatch_hooks for the ordered replay of these plugins: what a decoder reports is rendered to its JSON line
        on the replay thread that ran it (output_render = r433p_render), the commit only appends the lines in order"""
        from ._lib import DispatchHooks
        h = DispatchHooks()
        h.user = self.h
        h.output_render = C.cast(self.L.r433p_render, C.c_void_p).value
        return h

    def take(self):
        """-> (JSON lines since the last call as bytes, number of messages)"""
        text, n = C.c_char_p(), C.c_ulong()
        ln = self.L.r433p_take(self.h, C.byref(text), C.byref(n))
        return (C.string_at(text, ln) if ln else b""), int(n.value)

    def close(self):
        if self.h:
            self.L.r433p_destroy(self.h)
            self.h = None
