"""ctypes loader for librtl433hip.so.  Fails loudly: there is no Python/CPU fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "librtl433hip.so")


class FlowCfg(C.Structure):
    """r433_flow_cfg (include/r433_hip.h)."""
    _fields_ = [("sample_size", C.c_uint32), ("samp_rate", C.c_uint32), ("frame_samples", C.c_uint32),
                ("fpdm", C.c_uint32), ("use_mag_est", C.c_uint32), ("enable_fm", C.c_uint32),
                ("fm_low_pass", C.c_float), ("level_limit_db", C.c_float), ("min_level_db", C.c_float),
                ("min_snr_db", C.c_float), ("auto_level", C.c_float), ("center_frequency", C.c_uint32),
                ("input_format", C.c_uint32)]


class BatchTiming(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("detect_ms", "dir_ms", "count_ms", "scan_ms", "write_ms", "d2h_ms", "total_ms")]


class RDevice(C.Structure):
    """r433_r_device == the reference's r_device (include/r433_abi.h)."""


DECODE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(RDevice), C.c_void_p)
RDevice._fields_ = [
    ("protocol_num", C.c_uint), ("name", C.c_char_p), ("modulation", C.c_uint),
    ("short_width", C.c_float), ("long_width", C.c_float), ("reset_limit", C.c_float),
    ("gap_limit", C.c_float), ("sync_width", C.c_float), ("tolerance", C.c_float),
    ("decode_fn", C.c_void_p), ("create_fn", C.c_void_p), ("priority", C.c_uint), ("disabled", C.c_uint),
    ("fields", C.c_void_p), ("verbose", C.c_int), ("verbose_bits", C.c_int), ("log_fn", C.c_void_p),
    ("output_fn", C.c_void_p), ("decode_events", C.c_uint), ("decode_ok", C.c_uint),
    ("decode_messages", C.c_uint), ("decode_fails", C.c_uint * 5), ("decode_ctx", C.c_void_p),
    ("output_ctx", C.c_void_p)]

class PulseData(C.Structure):
    """pulse_data_t (reference include/pulse_data.h:30-50) == r433_pulse_data (include/r433_abi.h)."""
    _fields_ = [("offset", C.c_uint64), ("sample_rate", C.c_uint32), ("depth_bits", C.c_uint), ("start_ago", C.c_uint),
                ("end_ago", C.c_uint), ("num_pulses", C.c_uint), ("pulse", C.c_int * 1200), ("gap", C.c_int * 1200),
                ("ook_low_estimate", C.c_int), ("ook_high_estimate", C.c_int), ("fsk_f1_est", C.c_int),
                ("fsk_f2_est", C.c_int), ("freq1_hz", C.c_float), ("freq2_hz", C.c_float), ("centerfreq_hz", C.c_float),
                ("range_db", C.c_float), ("rssi_db", C.c_float), ("snr_db", C.c_float), ("noise_db", C.c_float)]


class HistBin(C.Structure):
    _fields_ = [("count", C.c_uint32), ("sum", C.c_int32), ("mean", C.c_int32), ("min", C.c_int32), ("max", C.c_int32)]


class Histogram(C.Structure):
    _fields_ = [("bins_count", C.c_uint32), ("bins", HistBin * 16)]


class DevTimingRow(C.Structure):
    _fields_ = [("modulation", C.c_uint32), ("short_width", C.c_float), ("long_width", C.c_float), ("reset_limit", C.c_float),
                ("gap_limit", C.c_float), ("sync_width", C.c_float), ("tolerance", C.c_float), ("priority", C.c_uint32)]


class Analysis(C.Structure):
    """r433_analysis (include/r433_records.h)."""
    _fields_ = [("num_pulses", C.c_uint32), ("total_period", C.c_int32), ("guess", C.c_uint32), ("reserved", C.c_uint32),
                ("pulses", Histogram), ("gaps", Histogram), ("periods_pg", Histogram), ("periods_gp", Histogram),
                ("timings", Histogram), ("device", DevTimingRow)]


class Grab(C.Structure):
    """r433_grab (include/r433_hip.h)."""
    _fields_ = [("stream", C.c_uint32), ("counter", C.c_uint32), ("byte_offset", C.c_uint64), ("byte_len", C.c_uint64),
                ("n_samples", C.c_uint32), ("clipped", C.c_uint32), ("pushed", C.c_uint64)]


class SigmfInfo(C.Structure):
    _fields_ = [("datatype", C.c_char * 32), ("sample_rate", C.c_uint32), ("frequency", C.c_uint32), ("sample_start", C.c_uint32),
                ("reserved", C.c_uint32), ("data_offset", C.c_uint64), ("data_len", C.c_uint64)]


class DigestCtx(C.Structure):
    _fields_ = [("sum", C.c_uint64), ("events", C.c_uint64)]


class DispatchInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("stream", "package", "device", "ordinal", "package_type", "start_ago")]


PACKAGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p)


class PkgRec(C.Structure):
    """r433_pkg_rec (include/r433_records.h): the header of a package record."""
    _fields_ = [("total_bytes", C.c_uint32), ("stream", C.c_uint32), ("type", C.c_uint32), ("num_pulses", C.c_uint32),
                ("frame", C.c_uint32), ("ret_pos", C.c_uint32), ("offset", C.c_uint64), ("start_ago", C.c_uint32),
                ("end_ago", C.c_uint32), ("ook_low", C.c_int32), ("ook_high", C.c_int32), ("fsk_f1", C.c_int32),
                ("fsk_f2", C.c_int32), ("sample_rate", C.c_uint32), ("reserved", C.c_uint32)]


HOOK_BEGIN_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(PkgRec), C.POINTER(PulseData))
HOOK_EVENT_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(RDevice), C.c_int, C.c_void_p)
HOOK_END_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(PkgRec), C.c_int)


HOOK_FILTER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(PkgRec))


class DispatchHooks(C.Structure):
    """r433_dispatch_hooks (include/r433_hip.h)."""
    _fields_ = [("user", C.c_void_p), ("package_begin", HOOK_BEGIN_FN), ("event_done", HOOK_EVENT_FN),
                ("package_end", HOOK_END_FN), ("package_filter", HOOK_FILTER_FN), ("output_render", C.c_void_p)]

_lib = None

# -w dump formats (include/r433_hip.h R433_DUMP_*)
DUMP_FORMATS = {"cu8": 1, "cs16": 2, "cs8": 3, "cf32": 4, "am.s16": 5, "fm.s16": 6, "am.f32": 7, "fm.f32": 8,
                "i.f32": 9, "q.f32": 10}

EXPORTS = [
    "r433_version", "r433_last_error", "r433_device_count", "r433_flow_cfg_default", "r433_level_db",
    "r433_batch_create", "r433_batch_create_on", "r433_batch_device", "r433_batch_destroy", "r433_batch_run", "r433_batch_packages", "r433_batch_events",
    "r433_batch_frame_sums", "r433_batch_device_events", "r433_batch_set_taps", "r433_batch_set_split",
    "r433_batch_split_stats", "r433_batch_detect_form", "r433_warmup", "r433_host_register", "r433_host_unregister", "r433_batch_set_profiling", "r433_batch_set_debug", "r433_batch_set_exclusive_detect", "r433_batch_set_staging_slot", "r433_batch_enable_logic_dump", "r433_batch_logic_dump",
    "r433_batch_get_timing", "r433_batch_debug_state", "r433_batch_dispatch", "r433_batch_dispatch_mt", "r433_dispatch_current",
    "r433_plugin_digest_decode", "r433_envelope_detect", "r433_magnitude_est_cu8",
    "r433_magnitude_est_cs16", "r433_convert_cs8_cu8", "r433_convert_cf32_cs16", "r433_dump_convert",
    "r433_batch_run_pulses", "r433_pulse_text_load", "r433_pulse_text_dump", "r433_batch_analyze", "r433_analysis_text",
    "r433_pulse_vcd_header", "r433_pulse_vcd", "r433_batch_grab_plan",
    "r433_sigmf_prefix", "r433_sigmf_trailer", "r433_sigmf_probe",
    "r433_filter_frame", "r433_envelope_host", "r433_host_alloc", "r433_host_free", "r433_batch_run_host", "r433_batch_dispatch_hooks", "r433_batch_dispatch_ordered", "r433_batch_decoded",
    "r433_dump_convert_host", "r433_batch_set_package_quality",
    "r433_batch_set_stateless", "r433_batch_probe_prefilter", "r433_batch_set_prefilter", "r433_batch_prefilter_counts", "r433_prefilter_forget", "r433_prefilter_set_helper_probe",
    "r433_fsk_step", "r433_detector_create", "r433_detector_destroy", "r433_detector_reset", "r433_detector_set_levels", "r433_detector_package",
]


def lib():
    """The product library (HIP, gfx950).  There is no fallback: missing library == hard error."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch ships its own HIP runtime (same soname as /opt/rocm's).  Whichever is loaded first serves the whole
        # process; if the system one wins, torch finds "no HIP GPUs" later.  Python callers use torch for device memory
        # anyway, so let it load first.  (A C host -- dropin/ -- has no torch and uses the system runtime.)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m rtl_433_amd.build` (hipcc, gfx950). "
                           "rtl_433_amd has no CPU fallback.")
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def bind(L):
    """Attach the C ABI prototypes of include/r433_hip.h to a loaded shared object."""
    vp = C.c_void_p
    L.r433_version.restype = C.c_int
    L.r433_last_error.restype = C.c_char_p
    L.r433_device_count.restype = C.c_int
    L.r433_flow_cfg_default.restype = None
    L.r433_flow_cfg_default.argtypes = [C.POINTER(FlowCfg), C.c_uint32, C.c_uint32]
    L.r433_level_db.restype = C.c_float
    L.r433_level_db.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    L.r433_batch_create.restype = vp
    L.r433_batch_create.argtypes = [C.POINTER(FlowCfg), vp, C.c_uint32]
    if hasattr(L, "r433_batch_create_on"):
        L.r433_batch_create_on.restype = vp
        L.r433_batch_create_on.argtypes = [C.c_int, C.POINTER(FlowCfg), vp, C.c_uint32]
        L.r433_batch_device.restype = C.c_int
        L.r433_batch_device.argtypes = [vp]
    L.r433_batch_destroy.restype = None
    L.r433_batch_destroy.argtypes = [vp]
    L.r433_batch_run.restype = C.c_int
    L.r433_batch_run.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp]
    for f in (L.r433_batch_packages, L.r433_batch_events):
        f.restype = C.c_int
        f.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    L.r433_batch_frame_sums.restype = C.c_int
    L.r433_batch_frame_sums.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint32)]
    L.r433_batch_device_events.restype = C.c_int
    L.r433_batch_device_events.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.r433_batch_set_taps.restype = C.c_int
    L.r433_batch_set_taps.argtypes = [vp, vp, vp, vp, C.c_uint64]
    L.r433_batch_set_split.restype = C.c_int
    L.r433_batch_set_split.argtypes = [vp, C.c_uint32]
    L.r433_batch_enable_logic_dump.restype = C.c_int
    L.r433_batch_enable_logic_dump.argtypes = [vp, C.c_int]
    L.r433_batch_logic_dump.restype = C.c_int
    L.r433_batch_logic_dump.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.r433_batch_set_exclusive_detect.restype = C.c_int
    L.r433_batch_set_exclusive_detect.argtypes = [vp, C.c_int]
    L.r433_batch_set_staging_slot.restype = C.c_int
    L.r433_batch_set_staging_slot.argtypes = [vp, C.c_uint32]
    L.r433_batch_set_debug.restype = C.c_int
    L.r433_batch_set_debug.argtypes = [vp, C.c_uint32]
    L.r433_batch_detect_form.restype = C.c_int
    L.r433_batch_detect_form.argtypes = [vp]
    L.r433_batch_split_stats.restype = C.c_int
    L.r433_batch_split_stats.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.r433_batch_set_profiling.restype = C.c_int
    L.r433_batch_set_profiling.argtypes = [vp, C.c_int]
    L.r433_batch_get_timing.restype = C.c_int
    L.r433_batch_get_timing.argtypes = [vp, C.POINTER(BatchTiming)]
    L.r433_batch_debug_state.restype = C.c_int
    L.r433_batch_debug_state.argtypes = [vp, vp, C.c_size_t]
    L.r433_batch_dispatch.restype = C.c_int
    L.r433_batch_dispatch.argtypes = [vp, vp, C.c_uint32, vp, vp]
    L.r433_batch_dispatch_mt.restype = C.c_int
    L.r433_batch_dispatch_mt.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_uint32]
    L.r433_dispatch_current.restype = C.c_int
    L.r433_dispatch_current.argtypes = [vp]
    for f in (L.r433_convert_cs8_cu8, L.r433_convert_cf32_cs16):
        f.restype = C.c_int
        f.argtypes = [vp, vp, C.c_uint64, vp]
    L.r433_batch_run_pulses.restype = C.c_int
    L.r433_batch_run_pulses.argtypes = [vp, vp, C.c_uint32, vp]
    L.r433_pulse_text_load.restype = C.c_int
    L.r433_pulse_text_load.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, vp, C.c_uint32]
    L.r433_pulse_text_dump.restype = C.c_int
    L.r433_pulse_text_dump.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
    L.r433_batch_analyze.restype = C.c_int
    L.r433_batch_analyze.argtypes = [vp, vp, C.c_uint32, vp]
    L.r433_analysis_text.restype = C.c_int
    L.r433_analysis_text.argtypes = [vp, C.c_uint32, vp, C.c_char_p, C.c_size_t]
    L.r433_pulse_vcd_header.restype = C.c_int
    L.r433_pulse_vcd_header.argtypes = [C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t]
    L.r433_pulse_vcd.restype = C.c_int
    L.r433_pulse_vcd.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    L.r433_batch_grab_plan.restype = C.c_int
    L.r433_batch_grab_plan.argtypes = [vp, C.c_int, vp, C.c_uint32]
    L.r433_sigmf_prefix.restype = C.c_int
    L.r433_sigmf_prefix.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, C.c_size_t]
    L.r433_sigmf_trailer.restype = C.c_int
    L.r433_sigmf_trailer.argtypes = [C.c_uint64, vp, C.c_size_t]
    L.r433_sigmf_probe.restype = C.c_int
    L.r433_sigmf_probe.argtypes = [vp, C.c_size_t, vp]
    L.r433_dump_convert.restype = C.c_int
    L.r433_dump_convert.argtypes = [C.c_int, C.c_uint32, vp, vp, C.c_uint64, vp]
    L.r433_filter_frame.restype = C.c_int
    L.r433_filter_frame.argtypes = [C.c_uint32, vp, C.c_uint32, vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64]
    L.r433_envelope_host.restype = C.c_int
    L.r433_envelope_host.argtypes = [C.c_uint32, vp, vp, C.c_uint32, vp]
    L.r433_batch_set_package_quality.restype = C.c_int
    L.r433_batch_set_package_quality.argtypes = [vp, vp, C.c_uint32]
    L.r433_dump_convert_host.restype = C.c_int
    L.r433_dump_convert_host.argtypes = [C.c_int, C.c_uint32, vp, vp, C.c_uint64]
    L.r433_detector_create.restype = vp
    L.r433_detector_create.argtypes = []
    L.r433_detector_destroy.restype = None
    L.r433_detector_destroy.argtypes = [vp]
    L.r433_detector_reset.restype = None
    L.r433_detector_reset.argtypes = [vp]
    L.r433_detector_set_levels.restype = None
    L.r433_detector_set_levels.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float]
    L.r433_detector_package.restype = C.c_int
    L.r433_detector_package.argtypes = [vp, vp, vp, C.c_int, C.c_uint32, C.c_uint64, vp, vp, C.c_uint]
    L.r433_fsk_step.restype = C.c_int
    L.r433_fsk_step.argtypes = [C.c_int, vp, C.c_int, vp]
    L.r433_host_alloc.restype = vp
    L.r433_host_alloc.argtypes = [C.c_size_t]
    L.r433_host_free.restype = None
    L.r433_host_free.argtypes = [vp]
    L.r433_host_register.restype = C.c_int
    L.r433_host_register.argtypes = [vp, C.c_size_t]
    L.r433_host_unregister.restype = C.c_int
    L.r433_host_unregister.argtypes = [vp]
    L.r433_warmup.restype = C.c_int
    L.r433_warmup.argtypes = []
    L.r433_batch_run_host.restype = C.c_int
    L.r433_batch_run_host.argtypes = [vp, vp, vp, C.c_uint32]
    L.r433_batch_dispatch_hooks.restype = C.c_int
    L.r433_batch_dispatch_hooks.argtypes = [vp, vp, C.c_uint32, vp]
    L.r433_batch_dispatch_ordered.restype = C.c_int
    L.r433_batch_dispatch_ordered.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    L.r433_batch_decoded.restype = C.c_int
    L.r433_batch_decoded.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint32)]
    L.r433_batch_probe_prefilter.restype = C.c_int
    L.r433_batch_probe_prefilter.argtypes = [vp, vp, C.c_uint32]
    if hasattr(L, "r433_prefilter_forget"):  # (development builds of earlier rounds are still bound for A/B timing)
        L.r433_prefilter_forget.restype = None
        L.r433_prefilter_forget.argtypes = []
    if hasattr(L, "r433_prefilter_set_helper_probe"):
        L.r433_prefilter_set_helper_probe.restype = None
        L.r433_prefilter_set_helper_probe.argtypes = [vp]
    if hasattr(L, "r433_batch_set_stateless"):
        L.r433_batch_set_stateless.restype = C.c_int
        L.r433_batch_set_stateless.argtypes = [vp, vp, C.c_uint32]
    L.r433_batch_set_prefilter.restype = C.c_int
    L.r433_batch_set_prefilter.argtypes = [vp, C.c_int]
    L.r433_batch_prefilter_counts.restype = C.c_int
    L.r433_batch_prefilter_counts.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint32)]
    for f in (L.r433_envelope_detect, L.r433_magnitude_est_cu8, L.r433_magnitude_est_cs16):
        f.restype = C.c_int
        f.argtypes = [vp, vp, C.c_uint32, vp, vp]
    return L


def last_error(L=None):
    return (L or lib()).r433_last_error().decode(errors="replace")


def check(rc, what, L=None):
    if rc < 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error(L)}")
    return rc
