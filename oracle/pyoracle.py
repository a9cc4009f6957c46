"""ctypes bindings for the oracle restatement and the built reference harness.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package rtl_433_amd never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libr433oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libr433ref.so")
REF_CLI = os.path.join(HERE, "_ref", "rtl_433_ref")

PKG_HDR = 64
EVT_HDR = 16
ROW_HDR = 8


class DevTiming(C.Structure):
    _fields_ = [("modulation", C.c_uint32), ("short_width", C.c_float), ("long_width", C.c_float),
                ("reset_limit", C.c_float), ("gap_limit", C.c_float), ("sync_width", C.c_float),
                ("tolerance", C.c_float), ("priority", C.c_uint32)]


DEV_DTYPE = np.dtype([("modulation", "<u4"), ("short_width", "<f4"), ("long_width", "<f4"),
                      ("reset_limit", "<f4"), ("gap_limit", "<f4"), ("sync_width", "<f4"),
                      ("tolerance", "<f4"), ("priority", "<u4")])


class Blob(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t), ("count", C.c_uint32)]


class FlowCfg(C.Structure):
    _fields_ = [("sample_size", C.c_uint32), ("samp_rate", C.c_uint32), ("frame_samples", C.c_uint32),
                ("fpdm", C.c_uint32), ("use_mag_est", C.c_uint32), ("enable_fm", C.c_uint32),
                ("fm_low_pass", C.c_float), ("level_limit_db", C.c_float), ("min_level_db", C.c_float),
                ("min_snr_db", C.c_float), ("auto_level", C.c_float), ("load_format", C.c_uint32)]


class FlowOut(C.Structure):
    _fields_ = [("packages", Blob), ("events", Blob), ("env", C.c_void_p), ("am", C.c_void_p),
                ("fm", C.c_void_p), ("frame_sums", C.c_void_p)]


class LpfState(C.Structure):
    _fields_ = [("y_prev", C.c_int16), ("x_prev", C.c_int16)]


class FmState(C.Structure):
    _fields_ = [("xr", C.c_int32), ("xi", C.c_int32), ("xf", C.c_int32), ("yf", C.c_int32),
                ("rate", C.c_uint32), ("a16", C.c_int32), ("b16", C.c_int32), ("a32", C.c_int64), ("b32", C.c_int64)]


def build_oracle(force=False):
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "r433_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return ORACLE_SO


def build_ref(ref="/root/reference"):
    """Builds oracle/_ref from the reference tree when it is present; otherwise keeps what is there."""
    if os.path.isdir(os.path.join(ref, "src")):
        src_time = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("ref_harness.c", "Makefile"))
        if not os.path.exists(REF_SO) or os.path.getmtime(REF_SO) < src_time:
            subprocess.check_call(["make", "-s", "-C", HERE, "ref", f"REF={ref}"])
    return REF_SO if os.path.exists(REF_SO) else None


def have_ref():
    return os.path.exists(REF_SO)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        u8p, u16p, i16p = C.c_void_p, C.c_void_p, C.c_void_p
        L.orc_envelope_cu8.restype = C.c_uint32
        L.orc_envelope_cu8.argtypes = [u8p, u16p, C.c_uint32]
        L.orc_magest_cu8.restype = C.c_uint32
        L.orc_magest_cu8.argtypes = [u8p, u16p, C.c_uint32]
        L.orc_magest_cs16.restype = C.c_uint32
        L.orc_magest_cs16.argtypes = [i16p, u16p, C.c_uint32]
        L.orc_level_db.restype = C.c_float
        L.orc_level_db.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.orc_lowpass.restype = None
        L.orc_lowpass.argtypes = [C.POINTER(LpfState), u16p, i16p, C.c_uint32]
        L.orc_fm_coeffs.restype = None
        L.orc_fm_coeffs.argtypes = [C.c_float, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_fm_cu8.restype = None
        L.orc_fm_cu8.argtypes = [C.POINTER(FmState), u8p, i16p, C.c_uint32, C.c_uint32, C.c_float]
        L.orc_fm_cs16.restype = None
        L.orc_fm_cs16.argtypes = [C.POINTER(FmState), i16p, i16p, C.c_uint32, C.c_uint32, C.c_float]
        L.orc_flow_run.restype = C.c_int
        L.orc_flow_run.argtypes = [C.POINTER(FlowCfg), C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint,
                                   C.c_uint32, C.c_uint32, C.POINTER(FlowOut)]
        L.orc_slice_packages.restype = C.c_int
        L.orc_slice_packages.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint32, C.POINTER(Blob)]
        L.orc_blob_free.restype = None
        L.orc_blob_free.argtypes = [C.POINTER(Blob)]
        L.orc_events_normalize.restype = C.c_size_t
        L.orc_events_normalize.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_events_digest.restype = C.c_uint64
        L.orc_events_digest.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.orc_events_digest2.restype = C.c_uint64
        L.orc_events_digest2.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        _oracle = L
    return _oracle


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def default_flow_cfg(sample_size=2, samp_rate=250000, fpdm=0, enable_fm=1, use_mag_est=0, fm_low_pass=0.0,
                     level_limit_db=0.0, min_level_db=-12.1442, min_snr_db=9.0, auto_level=0.0, frame_samples=None, load_format=0):
    """load_format 1 / 2: the capture is an am.s16 / fm.s16 file (sample_size 2)"""
    if frame_samples is None:
        frame_samples = 262144 // sample_size
    return FlowCfg(sample_size, samp_rate, frame_samples, fpdm, use_mag_est, enable_fm, fm_low_pass,
                   level_limit_db, min_level_db, min_snr_db, auto_level, load_format)


def _blob_bytes(b):
    if b.len == 0:
        return b""
    return C.string_at(b.data, b.len)


def oracle_flow(iq, devs, cfg, stream_index=0, pkg_base=0, taps=False):
    """Run one capture through the oracle.  iq: uint8 (cu8) or int16 (cs16) array.
    devs: numpy structured array (DEV_DTYPE) or None.  Returns dict."""
    L = oracle()
    iq = np.ascontiguousarray(iq)
    n_bytes = iq.nbytes
    n = n_bytes // cfg.sample_size
    out = FlowOut()
    keep = {}
    if taps:
        keep["env"] = np.zeros(n, dtype=np.uint16)
        keep["am"] = np.zeros(n, dtype=np.int16)
        keep["fm"] = np.zeros(n, dtype=np.int16)
        out.env, out.am, out.fm = _ptr(keep["env"]), _ptr(keep["am"]), _ptr(keep["fm"])
    n_frames = (n + cfg.frame_samples - 1) // cfg.frame_samples + 1
    keep["frame_sums"] = np.zeros(n_frames, dtype=np.uint32)
    out.frame_sums = _ptr(keep["frame_sums"])
    nd = 0 if devs is None else len(devs)
    dptr = None if devs is None else _ptr(np.ascontiguousarray(devs))
    npk = L.orc_flow_run(C.byref(cfg), _ptr(iq), n_bytes, dptr, nd, stream_index, pkg_base, C.byref(out))
    res = dict(n_packages=npk, packages=_blob_bytes(out.packages), events=_blob_bytes(out.events),
               n_events=out.events.count, **keep)
    L.orc_blob_free(C.byref(out.packages))
    L.orc_blob_free(C.byref(out.events))
    return res


def oracle_slice(pkg_blob, devs, pkg_base=0):
    L = oracle()
    b = Blob()
    buf = np.frombuffer(pkg_blob, dtype=np.uint8)
    devs = np.ascontiguousarray(devs)
    L.orc_slice_packages(_ptr(buf) if len(buf) else None, len(buf), _ptr(devs), len(devs), pkg_base, C.byref(b))
    ev = _blob_bytes(b)
    cnt = b.count
    L.orc_blob_free(C.byref(b))
    return ev, cnt


def events_normalize(blob):
    a = np.frombuffer(bytes(blob), dtype=np.uint8).copy()
    if a.size == 0:
        return b""
    n = oracle().orc_events_normalize(_ptr(a), a.size)
    return a[:n].tobytes()


def events_digest(blob):
    a = np.frombuffer(bytes(blob), dtype=np.uint8)
    cnt = C.c_uint32(0)
    d = oracle().orc_events_digest(_ptr(a) if a.size else None, a.size, C.byref(cnt))
    return int(d), int(cnt.value)


def events_digest2(blob):
    a = np.frombuffer(bytes(blob), dtype=np.uint8)
    cnt = C.c_uint32(0)
    d = oracle().orc_events_digest2(_ptr(a) if a.size else None, a.size, C.byref(cnt))
    return int(d), int(cnt.value)


def parse_packages(blob):
    """-> list of dicts with header fields and pulse/gap arrays."""
    out = []
    at = 0
    mv = memoryview(blob)
    while at + PKG_HDR <= len(blob):
        h = np.frombuffer(mv[at:at + PKG_HDR], dtype=np.uint32)
        total = int(h[0])
        npul = int(h[3])
        offset = int(np.frombuffer(mv[at + 24:at + 32], dtype=np.uint64)[0])
        ints = np.frombuffer(mv[at + 40:at + 56], dtype=np.int32)
        body = np.frombuffer(mv[at + PKG_HDR:at + PKG_HDR + 8 * npul], dtype=np.int32)
        out.append(dict(stream=int(h[1]), type=int(h[2]), num=npul, frame=int(h[4]), ret_pos=int(h[5]),
                        offset=offset, start_ago=int(h[8]), end_ago=int(h[9]), low=int(ints[0]), high=int(ints[1]),
                        f1=int(ints[2]), f2=int(ints[3]), rate=int(h[14]), pulse=body[0::2].copy(), gap=body[1::2].copy()))
        at += total
    return out


def parse_events(blob):
    out = []
    at = 0
    mv = memoryview(blob)
    while at + EVT_HDR <= len(blob):
        total, pkg = np.frombuffer(mv[at:at + 8], dtype=np.uint32)
        dev, ordinal, num_rows, free_row = np.frombuffer(mv[at + 8:at + 16], dtype=np.uint16)
        rows = []
        p = at + EVT_HDR
        for _ in range(int(num_rows)):
            bits, syncs, nbytes, _r = np.frombuffer(mv[p:p + 8], dtype=np.uint16)
            data = bytes(mv[p + 8:p + 8 + int(nbytes)])
            rows.append((int(bits), int(syncs), data))
            p += 8 + ((int(nbytes) + 3) & ~3)
        out.append(dict(pkg=int(pkg), dev=int(dev), ordinal=int(ordinal), num_rows=int(num_rows),
                        free_row=int(free_row), rows=rows))
        at += int(total)
    return out


def strip_ret_pos(pkg_blob):
    """The reference harness cannot observe ret_pos; zero it for comparisons."""
    a = np.frombuffer(bytes(pkg_blob), dtype=np.uint8).copy()
    at = 0
    while at + PKG_HDR <= a.size:
        total = int(a[at:at + 4].view(np.uint32)[0])
        a[at + 20:at + 24] = 0
        at += total
    return a.tobytes()


# ---------------------------------------------------------------- reference harness

class Ref:
    """Wrapper over oracle/_ref/libr433ref.so (the unmodified reference + ref_harness.c)."""

    def __init__(self, protocols=None, flex=None, call_real=False, record=True, json_path=None,
                 report_meta=0, report_protocol=0):
        if not have_ref():
            raise RuntimeError("oracle/_ref/libr433ref.so not built (needs /root/reference)")
        L = C.CDLL(REF_SO)
        self.L = L
        L.refh_create.restype = C.c_void_p
        L.refh_create.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
        L.refh_destroy.argtypes = [C.c_void_p]
        L.refh_num_devices.argtypes = [C.c_void_p]
        L.refh_device_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(DevTiming), C.POINTER(C.c_uint32), C.c_char_p, C.c_int]
        L.refh_set_levels.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 6
        L.refh_set_enable_fm.argtypes = [C.c_void_p, C.c_int]
        L.refh_run_capture.restype = C.c_int
        L.refh_run_capture.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                       C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for f in (L.refh_packages, L.refh_events):
            f.restype = C.POINTER(C.c_uint8)
            f.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        L.refh_returns.restype = C.POINTER(C.c_int32)
        L.refh_returns.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.refh_digest.restype = C.c_uint64
        L.refh_digest.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.refh_clear.argtypes = [C.c_void_p]
        L.refh_set_digest_mode.argtypes = [C.c_void_p, C.c_int]
        L.refh_digest2.restype = C.c_uint64
        L.refh_digest2.argtypes = [C.c_void_p]
        L.refh_abi_sizes.argtypes = [C.c_void_p]
        L.refh_add_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.refh_plain_devices.restype = C.c_int
        L.refh_plain_devices.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.refh_sink_count.restype = C.c_ulong
        if protocols is None:
            arr, n = None, 0
        else:
            arr = (C.c_int * max(1, len(protocols)))(*protocols)
            n = len(protocols) if len(protocols) else -1
        flex_s = None if not flex else "\n".join(flex).encode()
        jp = None if not json_path else json_path.encode()
        self.h = L.refh_create(arr, n, flex_s, int(call_real), int(record), jp, report_meta, report_protocol)

    def plain_devices(self):
        """-> ctypes array of r_device* (as void*) with the reference's real decode_fn, in registration order."""
        n = self.L.refh_num_devices(self.h)
        arr = (C.c_void_p * n)()
        got = self.L.refh_plain_devices(self.h, arr, n)
        assert got == n
        return arr

    def sink_count(self):
        return int(self.L.refh_sink_count())

    def text_mode(self, on=True):
        """Decoder messages (reference flow with call_real, and the plain devices) as JSON lines: take_text()."""
        self.L.refh_text_mode.argtypes = [C.c_void_p, C.c_int]
        self.L.refh_text_mode(self.h, int(on))

    def take_text(self):
        self.L.refh_take_text.restype = C.c_size_t
        self.L.refh_take_text.argtypes = [C.POINTER(C.c_char_p)]
        t = C.c_char_p()
        n = self.L.refh_take_text(C.byref(t))
        return C.string_at(t, n) if n else b""

    def add_rows(self, rows):
        """Register synthetic decoders (DEV_DTYPE timing rows, any modulation) after the devices registered so far."""
        rows = np.ascontiguousarray(rows, dtype=DEV_DTYPE)
        self.L.refh_add_rows(self.h, _ptr(rows), len(rows))

    def close(self):
        if self.h:
            self.L.refh_destroy(self.h)
            self.h = None

    def devices(self):
        """-> (DEV_DTYPE array, protocol numbers, names) in registration order."""
        n = self.L.refh_num_devices(self.h)
        devs = np.zeros(n, dtype=DEV_DTYPE)
        nums, names = [], []
        for i in range(n):
            t = DevTiming()
            pn = C.c_uint32()
            nm = C.create_string_buffer(128)
            self.L.refh_device_info(self.h, i, C.byref(t), C.byref(pn), nm, 128)
            devs[i] = (t.modulation, t.short_width, t.long_width, t.reset_limit, t.gap_limit, t.sync_width,
                       t.tolerance, t.priority)
            nums.append(int(pn.value))
            names.append(nm.value.decode(errors="replace"))
        return devs, nums, names

    def set_levels(self, use_mag_est=0, level_limit=0.0, min_level=-12.1442, min_snr=9.0, auto_level=0.0,
                   squelch_offset=0.0, fm_low_pass=0.0):
        self.L.refh_set_levels(self.h, use_mag_est, level_limit, min_level, min_snr, auto_level, squelch_offset, fm_low_pass)

    def set_enable_fm(self, on):
        self.L.refh_set_enable_fm(self.h, int(on))

    def set_load_format(self, fmt):
        """0: cu8 / cs16 by sample size; 1: am.s16, 2: fm.s16 input files (sample_size 2)"""
        self.L.refh_set_load_format.argtypes = [C.c_void_p, C.c_int]
        self.L.refh_set_load_format(self.h, int(fmt))

    def run(self, iq, sample_size=2, samp_rate=250000, center_freq=433920000, fpdm=2, stream_index=0, taps=False):
        iq = np.ascontiguousarray(iq)
        n = iq.nbytes // sample_size
        am = np.zeros(n, dtype=np.int16) if taps else None
        fm = np.zeros(n, dtype=np.int16) if taps else None
        nfr = (iq.nbytes + 262143) // 262144 + 1
        sums = np.zeros(nfr, dtype=np.uint32)
        dbs = np.zeros(nfr, dtype=np.float32)
        ev = self.L.refh_run_capture(self.h, _ptr(iq), iq.nbytes, sample_size, samp_rate, center_freq, fpdm,
                                     stream_index, _ptr(am) if taps else None, _ptr(fm) if taps else None,
                                     _ptr(sums), _ptr(dbs))
        return dict(events_ok=ev, am=am, fm=fm, frame_sums=sums, frame_db=dbs)

    def packages(self):
        ln, cnt = C.c_size_t(), C.c_uint32()
        p = self.L.refh_packages(self.h, C.byref(ln), C.byref(cnt))
        return (C.string_at(p, ln.value) if ln.value else b""), cnt.value

    def events(self):
        ln, cnt = C.c_size_t(), C.c_uint32()
        p = self.L.refh_events(self.h, C.byref(ln), C.byref(cnt))
        return (C.string_at(p, ln.value) if ln.value else b""), cnt.value

    def returns(self):
        cnt = C.c_size_t()
        p = self.L.refh_returns(self.h, C.byref(cnt))
        return np.ctypeslib.as_array(p, shape=(cnt.value,)).copy() if cnt.value else np.zeros(0, dtype=np.int32)

    def digest(self):
        ne, npk = C.c_uint32(), C.c_uint32()
        d = self.L.refh_digest(self.h, C.byref(ne), C.byref(npk))
        return int(d), int(ne.value), int(npk.value)

    def set_digest_mode(self, mode):
        self.L.refh_set_digest_mode(self.h, mode)

    def digest2(self):
        return int(self.L.refh_digest2(self.h))

    def clear(self):
        self.L.refh_clear(self.h)

    def abi_sizes(self):
        a = np.zeros(8, dtype=np.uint32)
        self.L.refh_abi_sizes(_ptr(a))
        return a


def canonical_events(blob):
    """Sort an event stream into (pkg, dev, ordinal) order (the reference emits by priority level)."""
    recs = []
    at = 0
    while at + EVT_HDR <= len(blob):
        total = int.from_bytes(blob[at:at + 4], "little")
        pkg = int.from_bytes(blob[at + 4:at + 8], "little")
        dev = int.from_bytes(blob[at + 8:at + 10], "little")
        ordinal = int.from_bytes(blob[at + 10:at + 12], "little")
        recs.append(((pkg, dev, ordinal), blob[at:at + total]))
        at += total
    recs.sort(key=lambda r: r[0])
    return b"".join(r[1] for r in recs)


# ---- -w dump formats: numpy restatement of reference src/r_flow.c:385-489 (TEST INFRASTRUCTURE) ----
DUMP_FORMATS = ("cu8", "cs16", "cs8", "cf32", "am.s16", "fm.s16", "am.f32", "fm.f32", "i.f32", "q.f32")


def dump_convert(fmt, sample_size, data):
    """What the reference's dumper writes for `fmt` given the IQ stream (sample_size 2: uint8 components,
    4: int16 components) -- or, for am.* / fm.*, given the am / fm int16 stream.  Returns bytes."""
    if fmt in ("am.s16", "fm.s16"):                       # r_flow.c:436-443: the buffers as they are
        return np.asarray(data, dtype=np.int16).tobytes()
    if fmt in ("am.f32", "fm.f32"):                       # r_flow.c:444-455: v * (1.0f / 0x8000)
        return (np.asarray(data, dtype=np.int16).astype(np.float32) * np.float32(1.0 / 0x8000)).tobytes()
    if sample_size == 2:
        c = np.asarray(data, dtype=np.uint8).astype(np.int32)
        if fmt == "cu8":                                  # r_flow.c:396: iq_buf itself
            return c.astype(np.uint8).tobytes()
        if fmt == "cs16":                                 # :404-408  x * 256 - 32768
            return (c * 256 - 32768).astype(np.int16).tobytes()
        if fmt == "cs8":                                  # :412-415  x - 128
            return (c - 128).astype(np.int8).tobytes()
        if fmt == "cf32":                                 # :424-427  (x - 128) / 128.0f
            return ((c - 128).astype(np.float32) / np.float32(128.0)).tobytes()
        if fmt in ("i.f32", "q.f32"):                     # :456-479  (x - 128) * (1.0f / 0x80)
            return ((c[(fmt == "q.f32")::2] - 128).astype(np.float32) * np.float32(1.0 / 0x80)).tobytes()
    else:
        c = np.asarray(data, dtype=np.int16).astype(np.int32)
        if fmt == "cs16":
            return c.astype(np.int16).tobytes()
        if fmt == "cu8":                                  # :397-400  x / 256 + 128, C division, stored to uint8
            q = np.where(c < 0, -((-c) // 256), c // 256)
            return ((q + 128) & 0xFF).astype(np.uint8).tobytes()
        if fmt == "cs8":                                  # :416-419  x >> 8
            return (c >> 8).astype(np.int8).tobytes()
        if fmt == "cf32":                                 # :428-431  x / 32768.0f
            return (c.astype(np.float32) / np.float32(32768.0)).tobytes()
        if fmt in ("i.f32", "q.f32"):                     # :466-468, :476-478  x * (1.0f / 0x8000)
            return (c[(fmt == "q.f32")::2].astype(np.float32) * np.float32(1.0 / 0x8000)).tobytes()
    raise ValueError(f"unknown dump format {fmt!r} for sample size {sample_size}")
