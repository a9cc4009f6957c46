"""Host-side mirror of the reference's offline flow for the accelerated path.

`BatchEngine` is what `rtl_433 -r a.cu8 -r b.cu8 ...` does between reading the files and calling
the decoders (reference src/rtl_433.c:1703-1854 -> src/r_flow.c:104-340 -> src/r_api.c:438-550),
for N captures at once, on one MI355X.  Device memory and streams come from PyTorch; everything
else goes through the C ABI in include/r433_hip.h.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from . import _lib
from ._lib import FlowCfg

DEV_DTYPE = np.dtype([("modulation", "<u4"), ("short_width", "<f4"), ("long_width", "<f4"),
                      ("reset_limit", "<f4"), ("gap_limit", "<f4"), ("sync_width", "<f4"),
                      ("tolerance", "<f4"), ("priority", "<u4")])

DEFAULT_DEVICE_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "r_devices_default.json")


def load_device_table(path=DEFAULT_DEVICE_TABLE):
    """Timing rows of the reference's default-enabled r_devices, in registration order.
    Returns (DEV_DTYPE array, protocol numbers, names)."""
    with open(path) as f:
        doc = json.load(f)
    rows = doc["devices"]
    devs = np.zeros(len(rows), dtype=DEV_DTYPE)
    for i, r in enumerate(rows):
        devs[i] = (r["modulation"], r["short_width"], r["long_width"], r["reset_limit"], r["gap_limit"],
                   r["sync_width"], r["tolerance"], r["priority"])
    return devs, [r["protocol"] for r in rows], [r["name"] for r in rows]


def flow_cfg(sample_size=2, samp_rate=250000, fpdm=0, enable_fm=1, use_mag_est=0, fm_low_pass=0.0,
             level_limit_db=0.0, min_level_db=-12.1442, min_snr_db=9.0, auto_level=0.0, frame_samples=0,
             center_frequency=433920000, input_format=0):
    return FlowCfg(sample_size, samp_rate, frame_samples, fpdm, use_mag_est, enable_fm, fm_low_pass,
                   level_limit_db, min_level_db, min_snr_db, auto_level, center_frequency, input_format)


class BatchEngine:
    def __init__(self, cfg: FlowCfg, devs=None, profiling=False, library=None, device=None):
        # `library` is for the test suite's emulator build of the same sources; the product always
        # goes through _lib.lib(), which raises if librtl433hip.so is missing.
        self.L = library if library is not None else _lib.lib()
        self.emulated = library is not None  # (the emulator's "device" memory is host memory)
        _lib.check(self.L.r433_device_count(), "r433_device_count", self.L)
        self.cfg = cfg
        self.devs = np.zeros(0, dtype=DEV_DTYPE) if devs is None else np.ascontiguousarray(devs, dtype=DEV_DTYPE)
        ptr = self.devs.ctypes.data_as(C.c_void_p) if len(self.devs) else None
        # device: the GPU the engine lives on (r433_batch_create_on); None = the calling thread's current device
        self.h = (self.L.r433_batch_create(C.byref(cfg), ptr, len(self.devs)) if device is None else
                  self.L.r433_batch_create_on(int(device), C.byref(cfg), ptr, len(self.devs)))
        if not self.h:
            raise RuntimeError("r433_batch_create failed: " + _lib.last_error(self.L))
        if profiling:
            _lib.check(self.L.r433_batch_set_profiling(self.h, 1), "r433_batch_set_profiling", self.L)
        self._taps = None

    def close(self):
        if getattr(self, "h", None):
            self.L.r433_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_ptr(self, ptr, stride, n_streams, stream_bytes=None, stream=0):
        """Raw form of run(): `ptr` is a device address, `stride` bytes between captures."""
        sb = None
        if stream_bytes is not None:
            sb_arr = np.ascontiguousarray(stream_bytes, dtype=np.uint32)
            assert len(sb_arr) == n_streams
            sb = sb_arr.ctypes.data_as(C.c_void_p)
        rc = self.L.r433_batch_run(self.h, C.c_void_p(ptr), stride, sb, n_streams, C.c_void_p(stream))
        return _lib.check(rc, "r433_batch_run", self.L)

    def run_host(self, captures):
        """captures: list of numpy arrays in host memory (r433_batch_run_host: the library stages them itself)."""
        arrs = [np.ascontiguousarray(a) for a in captures]
        n = len(arrs)
        ptrs = (C.c_void_p * max(1, n))(*[a.ctypes.data for a in arrs])
        lens = (C.c_uint32 * max(1, n))(*[a.nbytes for a in arrs])
        rc = self.L.r433_batch_run_host(self.h, C.cast(ptrs, C.c_void_p), C.cast(lens, C.c_void_p), n)
        return _lib.check(rc, "r433_batch_run_host", self.L)

    def dispatch_hooks(self, rdevices, hooks):
        """hooks: a _lib.DispatchHooks (or None)."""
        rc = self.L.r433_batch_dispatch_hooks(self.h, C.cast(rdevices, C.c_void_p), len(rdevices),
                                              C.byref(hooks) if hooks is not None else None)
        return _lib.check(rc, "r433_batch_dispatch_hooks", self.L)

    def dispatch_ordered(self, rdevices, hooks=None, n_threads=8):
        """Threads own decoders, outputs committed in reference order (r433_batch_dispatch_ordered)."""
        rc = self.L.r433_batch_dispatch_ordered(self.h, C.cast(rdevices, C.c_void_p), len(rdevices),
                                                C.byref(hooks) if hooks is not None else None, n_threads)
        return _lib.check(rc, "r433_batch_dispatch_ordered", self.L)

    def set_stateless(self, flags):
        """flags: ctypes uint8 array, one per device (plugins.stateless_flags), or None (r433_batch_set_stateless)"""
        rc = self.L.r433_batch_set_stateless(self.h, C.cast(flags, C.c_void_p) if flags is not None else None,
                                             len(flags) if flags is not None else 0)
        return _lib.check(rc, "r433_batch_set_stateless", self.L)

    def probe_prefilter(self, rdevices, helper=None):
        """Learn which bitbuffers each decoder provably refuses on its head alone (r433_batch_probe_prefilter); from the
        next run on the slicer kernel drops those records.  -> number of decoders with a table.
        helper: address of the host's r433_helper_probe accessor (plugins.Plugins.helper_probe(): the decoders' bitbuffer
        helpers are wrapped and answer the probe without the payload, r433_prefilter_set_helper_probe), or None"""
        if helper is not None:
            self.L.r433_prefilter_set_helper_probe(helper)
        rc = self.L.r433_batch_probe_prefilter(self.h, C.cast(rdevices, C.c_void_p), len(rdevices))
        return _lib.check(rc, "r433_batch_probe_prefilter", self.L)

    def set_prefilter(self, on):
        _lib.check(self.L.r433_batch_set_prefilter(self.h, int(on)), "r433_batch_set_prefilter", self.L)

    def prefilter_counts(self):
        """[device][5] records the last run dropped on the device, by failure code (0, 1 = ABORT_LENGTH, 2 = ABORT_EARLY ...)"""
        p, n = C.c_void_p(), C.c_uint32()
        _lib.check(self.L.r433_batch_prefilter_counts(self.h, C.byref(p), C.byref(n)), "r433_batch_prefilter_counts", self.L)
        if not n.value or not p.value:
            return np.zeros((0, 5), dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value, 5)).copy()

    def decoded(self):
        """per package: events reported by its decoders in the last dispatch"""
        p, n = C.c_void_p(), C.c_uint32()
        _lib.check(self.L.r433_batch_decoded(self.h, C.byref(p), C.byref(n)), "r433_batch_decoded", self.L)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int)), shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.int32)

    def run(self, iq, stream_bytes=None, stream=None):
        """iq: CUDA tensor [n_streams, stride] of uint8 (cu8) or int16 (cs16), contiguous."""
        import torch
        assert (iq.is_cuda or self.emulated) and iq.is_contiguous() and iq.dim() == 2 and iq.data_ptr() % 16 == 0
        n_streams = iq.shape[0]
        stride = iq.shape[1] * iq.element_size()
        sb = None
        if stream_bytes is not None:
            sb_arr = np.ascontiguousarray(stream_bytes, dtype=np.uint32)
            assert len(sb_arr) == n_streams
            sb = sb_arr.ctypes.data_as(C.c_void_p)
        st = stream if stream is not None else (torch.cuda.current_stream().cuda_stream if iq.is_cuda else 0)
        rc = self.L.r433_batch_run(self.h, C.c_void_p(iq.data_ptr()), stride, sb, n_streams, C.c_void_p(st))
        return _lib.check(rc, "r433_batch_run", self.L)

    def set_split(self, segment_samples):
        """Cut captures longer than segment_samples into independently processed, verified segments."""
        _lib.check(self.L.r433_batch_set_split(self.h, int(segment_samples)), "r433_batch_set_split", self.L)

    def enable_logic_dump(self, on=True):
        _lib.check(self.L.r433_batch_enable_logic_dump(self.h, int(on)), "r433_batch_enable_logic_dump", self.L)

    def logic_dump(self, lengths):
        """-> list of uint8 arrays, one per capture (lengths in samples): the `-w file.u8` bytes of the last run."""
        p, st = C.c_void_p(), C.c_uint64()
        _lib.check(self.L.r433_batch_logic_dump(self.h, C.byref(p), C.byref(st)), "r433_batch_logic_dump", self.L)
        out = []
        for s, n in enumerate(lengths):
            out.append(np.ctypeslib.as_array(C.cast(p.value + s * st.value, C.POINTER(C.c_uint8)), shape=(int(n),)).copy() if n else np.zeros(0, dtype=np.uint8))
        return out

    def set_exclusive_detect(self, on=True):
        """Engines of a software pipeline take turns on the detection kernel (1 / True), or on all kernels of a pass (2)
        (r433_batch_set_exclusive_detect)."""
        _lib.check(self.L.r433_batch_set_exclusive_detect(self.h, int(on)), "r433_batch_set_exclusive_detect", self.L)

    def set_staging_slot(self, nbytes):
        """bytes per (package, device) staging slot of the slicers, 512 .. 8192 (0: the default 8192); records over it are sliced again by the placing pass"""
        _lib.check(self.L.r433_batch_set_staging_slot(self.h, int(nbytes)), "r433_batch_set_staging_slot", self.L)

    def set_debug(self, flags):
        """Development switches (R433_DEBUG_* of include/r433_hip.h): 1 blind cuts, 2 two-pass slicer, 1024 phase timing."""
        _lib.check(self.L.r433_batch_set_debug(self.h, int(flags)), "r433_batch_set_debug", self.L)

    def split_stats(self):
        a, b = C.c_uint32(), C.c_uint32()
        _lib.check(self.L.r433_batch_split_stats(self.h, C.byref(a), C.byref(b)), "r433_batch_split_stats", self.L)
        return dict(segments=a.value, pieces_rerun=b.value, detect_form=_lib.check(self.L.r433_batch_detect_form(self.h), "r433_batch_detect_form", self.L))

    def enable_taps(self, n_streams, n_samples):
        import torch
        env = torch.zeros((n_streams, n_samples), dtype=torch.int16, device="cuda")  # u16 payload
        am = torch.zeros((n_streams, n_samples), dtype=torch.int16, device="cuda")
        fm = torch.zeros((n_streams, n_samples), dtype=torch.int16, device="cuda")
        _lib.check(self.L.r433_batch_set_taps(self.h, C.c_void_p(env.data_ptr()), C.c_void_p(am.data_ptr()),
                                              C.c_void_p(fm.data_ptr()), n_samples), "r433_batch_set_taps")
        self._taps = (env, am, fm)
        return self._taps

    def taps(self):
        env, am, fm = self._taps
        return env.cpu().numpy().view(np.uint16), am.cpu().numpy(), fm.cpu().numpy()

    def _blob(self, fn):
        p, n, c = C.c_void_p(), C.c_size_t(), C.c_uint32()
        _lib.check(fn(self.h, C.byref(p), C.byref(n), C.byref(c)), fn.__name__, self.L)
        return (C.string_at(p, n.value) if n.value else b""), c.value

    def sizes(self):
        """(package bytes, packages, event-record bytes, event records) of the last run -- no copies"""
        p, n, c = C.c_void_p(), C.c_size_t(), C.c_uint32()
        _lib.check(self.L.r433_batch_packages(self.h, C.byref(p), C.byref(n), C.byref(c)), "r433_batch_packages", self.L)
        pk = (n.value, c.value)
        _lib.check(self.L.r433_batch_events(self.h, C.byref(p), C.byref(n), C.byref(c)), "r433_batch_events", self.L)
        return pk[0], pk[1], n.value, c.value

    def packages(self):
        return self._blob(self.L.r433_batch_packages)

    def events(self):
        return self._blob(self.L.r433_batch_events)

    def frame_sums(self, n_streams):
        p, cap = C.c_void_p(), C.c_uint32()
        _lib.check(self.L.r433_batch_frame_sums(self.h, C.byref(p), C.byref(cap)), "r433_batch_frame_sums", self.L)
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n_streams * cap.value,))
        return a.reshape(n_streams, cap.value).copy()

    def timing(self):
        t = _lib.BatchTiming()
        _lib.check(self.L.r433_batch_get_timing(self.h, C.byref(t)), "r433_batch_get_timing", self.L)
        return {n: getattr(t, n) for n, _ in t._fields_}

    def run_pulses(self, pulses, stream=None):
        """The `.ook` side door: `pulses` is a ctypes array of _lib.PulseData; straight to the decoder fan-out."""
        rc = self.L.r433_batch_run_pulses(self.h, C.cast(pulses, C.c_void_p), len(pulses), stream)
        return _lib.check(rc, "r433_batch_run_pulses", self.L)

    def analyze(self, stream=None):
        """Pulse analyzer (-A) over the packages of the last run -> ctypes array of _lib.Analysis."""
        n = self.packages()[1]
        arr = (_lib.Analysis * max(n, 1))()
        got = _lib.check(self.L.r433_batch_analyze(self.h, C.cast(arr, C.c_void_p), n, stream), "r433_batch_analyze", self.L)
        return (_lib.Analysis * got).from_buffer(arr) if got else (_lib.Analysis * 0)()

    def analysis_text(self, pkg, analysis):
        buf = C.create_string_buffer(64 * 1024)
        n = _lib.check(self.L.r433_analysis_text(self.h, pkg, C.byref(analysis), buf, len(buf)), "r433_analysis_text", self.L)
        return buf.raw[:n].decode()

    def grab_plan(self, mode=1, max_grabs=4096):
        """Sample grabber (-S all / unknown / known = 1 / 2 / 3): byte ranges of the captures the reference would save."""
        arr = (_lib.Grab * max_grabs)()
        n = _lib.check(self.L.r433_batch_grab_plan(self.h, mode, C.cast(arr, C.c_void_p), max_grabs), "r433_batch_grab_plan", self.L)
        return [arr[k] for k in range(min(n, max_grabs))]

    def dispatch(self, rdevices, pkg_cb=None, user=None, n_threads=1):
        """rdevices: ctypes array of POINTER(RDevice) in registration order."""
        cb = C.cast(pkg_cb, C.c_void_p) if pkg_cb is not None else None
        rc = self.L.r433_batch_dispatch_mt(self.h, C.cast(rdevices, C.c_void_p), len(rdevices), cb, user, n_threads)
        return _lib.check(rc, "r433_batch_dispatch_mt", self.L)


def make_rdevices(devs, decode_fn_addr=None, ctx_addr=None, names=None, protocols=None):
    """Builds reference-layout r_device objects (include/r433_abi.h) for timing rows `devs`.
    Returns (array of POINTER(RDevice), list of RDevice) -- keep both alive while dispatching."""
    objs = []
    for i, d in enumerate(devs):
        r = _lib.RDevice()
        r.protocol_num = int(protocols[i]) if protocols is not None else i + 1
        r.name = (names[i] if names is not None else f"dev{i}").encode()
        r.modulation = int(d["modulation"])
        r.short_width = float(d["short_width"])
        r.long_width = float(d["long_width"])
        r.reset_limit = float(d["reset_limit"])
        r.gap_limit = float(d["gap_limit"])
        r.sync_width = float(d["sync_width"])
        r.tolerance = float(d["tolerance"])
        r.priority = int(d["priority"])
        r.decode_fn = decode_fn_addr
        r.decode_ctx = ctx_addr
        objs.append(r)
    arr = (C.POINTER(_lib.RDevice) * len(objs))(*[C.pointer(o) for o in objs])
    return arr, objs


def digest_plugin_addr():
    """Address of the library's checksum decode_fn (r433_plugin_digest_decode)."""
    return C.cast(_lib.lib().r433_plugin_digest_decode, C.c_void_p).value


def load_pulse_text(text, sample_rate, max_packages=4096, library=None):
    """`.ook` text (bytes) -> ctypes array of _lib.PulseData, like the reference's file loop reads it."""
    L = library or _lib.lib()
    arr = (_lib.PulseData * max_packages)()
    n = _lib.check(L.r433_pulse_text_load(text, len(text), sample_rate, C.cast(arr, C.c_void_p), max_packages), "r433_pulse_text_load", L)
    return (_lib.PulseData * n).from_buffer(arr) if n else (_lib.PulseData * 0)()


def dump_pulse_text(pd, received=None, library=None):
    """One package as `.ook` text (bytes), reference pulse_data_dump."""
    L = library or _lib.lib()
    buf = C.create_string_buffer(64 * 1024)
    n = _lib.check(L.r433_pulse_text_dump(C.byref(pd), received, buf, len(buf)), "r433_pulse_text_dump", L)
    return buf.raw[:n]


def dump_convert(fmt, sample_size, d_in, n_out, stream=None, library=None):
    """-w dump format `fmt` ("cs16", "cf32", "am.f32", ... see _lib.DUMP_FORMATS) of a device tensor -> new uint8 device
    tensor with the converted stream (reference src/r_flow.c:385-489)."""
    import torch
    L = library or _lib.lib()
    width = {"cu8": 1, "cs8": 1, "cs16": 2, "am.s16": 2, "fm.s16": 2}.get(fmt, 4)
    out = torch.empty(n_out * width + 16, dtype=torch.uint8, device=d_in.device)
    _lib.check(L.r433_dump_convert(_lib.DUMP_FORMATS[fmt], sample_size, C.c_void_p(d_in.data_ptr()), C.c_void_p(out.data_ptr()),
                                   n_out, stream), "r433_dump_convert", L)
    return out[: n_out * width]
