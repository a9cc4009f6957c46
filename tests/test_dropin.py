"""The drop-in itself: rtl_433's own CLI, decoders and JSON printer (compiled unmodified from the reference tree by
dropin/Makefile) with src/r_flow.c replaced by dropin/r_flow_hip.c, against the stock reference binary
(oracle/_ref/rtl_433_ref) on the same `-r` file lists.  stdout must be byte-identical: JSON fields, time stamps,
levels, tags, order.

  * CPU suite: dropin/_build/rtl_433_emu (the kernels on the wave emulator) -- checks r_flow_hip.c and the host API.
  * -m gpu   : dropin/_build/rtl_433_hip (librtl433hip.so on the MI355X).

Both binaries hold reference code, are built only where /root/reference exists and travel to the GPU box prebuilt.
"""
from __future__ import annotations

import os
import shutil
import subprocess

import numpy as np
import pytest

from rtl_433_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "rtl_433_ref")
EMU = os.path.join(ROOT, "dropin", "_build", "rtl_433_emu")
HIP = os.path.join(ROOT, "dropin", "_build", "rtl_433_hip")
# (both are linked WITHOUT src/baseband.c, src/pulse_detect*.c and src/pulse_slicer.c: librtl433seam.so stands in for what the
# rest of the reference calls by those names -- test_no_reference_dsp_symbol_is_linked)
PLUGINS = os.path.join(ROOT, "dropin", "_build", "libr433plugins.so")
# the reference with its OWN src/r_flow.c and src/rtl_433.c, only the four DSP units replaced by librtl433seam.so: every
# frame goes through envelope_detect / baseband_low_pass_filter / baseband_demod_FM / pulse_detect_package / pulse_slicer_*
EMU_REFFLOW = os.path.join(ROOT, "dropin", "_build", "rtl_433_refflow_emu")
HIP_REFFLOW = os.path.join(ROOT, "dropin", "_build", "rtl_433_refflow_hip")
GOLD = os.path.join(ROOT, "tests", "golden")

FLEX = ["-X", "n=pwm,m=OOK_PWM,s=300,l=600,r=5000,g=2000,t=150",
        "-X", "n=ppm,m=OOK_PPM,s=300,l=600,r=5000,g=2000,t=150",
        "-X", "n=mc,m=OOK_MC_ZEROBIT,s=300,l=300,r=5000"]
KAT_LINE = ('{"time" : "@0.008000s", "protocol" : 169, "model" : "Nice-FlorS", "button" : 1, "serial" : 87791745, '
            '"code" : 1139, "count" : 6, "mod" : "ASK", "freq" : 433.955, "rssi" : -2.312, "snr" : 39.833, "noise" : -42.144}')


def _ensure_built(binary):
    if os.path.exists(binary) and os.path.exists(REF):
        return
    if not os.path.isdir("/root/reference/src"):
        pytest.skip(f"{os.path.relpath(binary, ROOT)} not built and no reference tree to build it from")
    from oracle import pyoracle as po
    po.build_ref()
    if binary in (EMU, EMU_REFFLOW):
        from tests.emu import build_emu
        build_emu.build()
    if binary in (EMU_REFFLOW, HIP_REFFLOW):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "dropin"), "refflow"], stdout=subprocess.DEVNULL)
        return
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dropin"), "emu" if binary == EMU else "hip"], stdout=subprocess.DEVNULL)


def run_cli(binary, args, cwd, env=None, with_stderr=False):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([binary] + args, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    return (p.stdout.decode(), p.stderr.decode(errors="replace")) if with_stderr else p.stdout.decode()


def write_ook_files(d, seeds):
    names = []
    for s in seeds:
        iq, _ = synth.ook_stream(s)
        name = f"s{s:05d}_433.92M_250k.cu8"
        iq.tofile(os.path.join(d, name))
        names.append(name)
    return names


def config3_recipe(path):
    """1024 kS/s cs16, two Manchester FSK bursts, +-40 kHz, 51 samples per half bit."""
    iq = synth.fsk_stream_cs16(3, 6000 + 2 * (112 * 2 * 51 + 20000), rate=1024000, dev_hz=40e3, halfbit_us=51 / 1.024,
                               coding="mc", n_bursts=2, nbits=96, amp=0.8, sigma=0.01, lead_in=6000, gap=20000)
    iq.tofile(path)


def file_args(names):
    out = []
    for n in names:
        out += ["-r", n]
    return out


def check_kat(binary, tmp_path):
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    args = ["-r", "g001_433.92M_250k.cu8", "-R", "169", "-F", "json", "-M", "level", "-M", "protocol"]
    ref = run_cli(REF, args, tmp_path)
    got = run_cli(binary, args, tmp_path)
    assert ref.strip() == KAT_LINE  # the reference's own end-to-end vector
    assert got == ref


def check_batch(binary, tmp_path, seeds, extra_env=None):
    names = write_ook_files(tmp_path, seeds)
    # all default decoders + three generic ones (random payloads rarely pass a real decoder's checksum), tagged per file
    args = file_args(names) + FLEX + ["-F", "json", "-M", "level", "-M", "protocol", "-M", "bits", "-K", "FILE"]
    ref = run_cli(REF, args, tmp_path)
    got = run_cli(binary, args, tmp_path, extra_env)
    assert ref.count("\n") >= len(seeds)  # the generic decoders see every burst
    assert got == ref
    return ref


def check_two_passes_in_flight(binary, tmp_path):
    """The GPU leg of a pass on a thread of its own while the pass before is replayed (dropin/r_flow_hip.c pass_start /
    pass_replay): forced for every pass of three files -- five passes, the last one and the tail of the list synchronous --
    stdout is the stock binary's, and the flow's own with the overlap off."""
    seeds = list(range(300, 314))
    ref = check_batch(binary, tmp_path, seeds, {"RTL433_HIP_BATCH": "3", "RTL433_HIP_OVERLAP": "1"})
    names = [f"s{s:05d}_433.92M_250k.cu8" for s in seeds]
    args = file_args(names) + FLEX + ["-F", "json", "-M", "level", "-M", "protocol", "-M", "bits", "-K", "FILE"]
    assert run_cli(binary, args, tmp_path, {"RTL433_HIP_BATCH": "3", "RTL433_HIP_OVERLAP": "0"}) == ref
    assert run_cli(binary, args, tmp_path, {"RTL433_HIP_BATCH": "5", "RTL433_HIP_OVERLAP": "1", "RTL433_HIP_THREADS": "1"}) == ref
    # The decoder pre-filter's questions come due in the THIRD pass (after 0.8 MB of samples), while the second pass is owed its
    # replay: they are asked on the file loop's thread between the join and the next pass's start -- on the pass's own thread
    # they called every decode_fn (output_fn / log_fn swapped for swallowers) beside the replay that runs the same decoders.
    # (-M stats: the per-decoder statistics the filter must keep are in the comparison)
    args_stats = args + ["-M", "stats:2:0"]
    ref_stats = run_cli(REF, args_stats, tmp_path)
    late = {"RTL433_HIP_BATCH": "3", "RTL433_HIP_OVERLAP": "1", "RTL433_HIP_PREFILTER_FROM": "800000", "RTL433_HIP_THREADS": "4", "RTL433_HIP_TRACE": "1"}
    got = run_cli(binary, args_stats, tmp_path, late, with_stderr=True)
    assert got[0] == ref_stats
    asked = [ln for ln in got[1].splitlines() if "pre-filter questions" in ln]
    assert len(asked) >= 1 and all("file loop's thread" in ln for ln in asked), got[1][-3000:]


def check_config3(binary, tmp_path):
    config3_recipe(tmp_path / "mc_868M_1024k.cs16")
    args = ["-r", "mc_868M_1024k.cs16", "-X", "n=mc,m=FSK_MC_ZEROBIT,s=50,l=50,r=120", "-F", "json", "-M", "level"]
    ref = run_cli(REF, args, tmp_path)
    got = run_cli(binary, args, tmp_path)
    assert '"mod" : "FSK"' in ref and ref.count("\n") >= 2
    assert got == ref
    # the classic detector (below 800 MHz) and every default decoder on the same samples
    shutil.copy(tmp_path / "mc_868M_1024k.cs16", tmp_path / "mc_433.92M_1024k.cs16")
    args = ["-r", "mc_433.92M_1024k.cs16", "-X", "n=mc,m=FSK_MC_ZEROBIT,s=50,l=50,r=120", "-F", "json", "-M", "level"]
    assert run_cli(binary, args, tmp_path) == run_cli(REF, args, tmp_path)


def check_mixed_list(binary, tmp_path):
    """cu8 and cs16 files, two sample rates, an empty and a short file in one list: groups and order."""
    names = write_ook_files(tmp_path, [11, 12])
    config3_recipe(tmp_path / "mc_868M_1024k.cs16")
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    (tmp_path / "empty_433.92M_250k.cu8").write_bytes(b"")
    np.full(2 * 300, 128, dtype=np.uint8).tofile(tmp_path / "short_433.92M_250k.cu8")
    iq, _ = synth.ook_stream(13, rate=1000000)
    iq.tofile(tmp_path / "s13_433.92M_1000k.cu8")
    order = [names[0], "mc_868M_1024k.cs16", "empty_433.92M_250k.cu8", "g001_433.92M_250k.cu8", "s13_433.92M_1000k.cu8",
             "short_433.92M_250k.cu8", names[1]]
    args = file_args(order) + FLEX + ["-X", "n=fmc,m=FSK_MC_ZEROBIT,s=50,l=50,r=120", "-F", "json", "-M", "level", "-K", "FILE"]
    ref = run_cli(REF, args, tmp_path)
    assert '"mod" : "FSK"' in ref and '"mod" : "ASK"' in ref
    assert run_cli(binary, args, tmp_path) == ref
    assert run_cli(binary, args, tmp_path, {"RTL433_HIP_BATCH": "1"}) == ref  # one GPU pass per file


# ---- CPU suite: the emulator build ----

def test_emu_kat(tmp_path):
    _ensure_built(EMU)
    check_kat(EMU, tmp_path)


def test_emu_batch(tmp_path):
    _ensure_built(EMU)
    check_batch(EMU, tmp_path, range(6))


def test_emu_config3(tmp_path):
    _ensure_built(EMU)
    check_config3(EMU, tmp_path)


def test_emu_mixed_list(tmp_path):
    _ensure_built(EMU)
    check_mixed_list(EMU, tmp_path)


# ---- GPU: the product library ----

@pytest.mark.gpu
def test_hip_kat(tmp_path):
    _ensure_built(HIP)
    check_kat(HIP, tmp_path)


@pytest.mark.gpu
def test_hip_batch64(tmp_path):
    _ensure_built(HIP)
    ref = check_batch(HIP, tmp_path, range(64))
    assert ref.count("\n") > 64


def test_emu_two_passes_in_flight(tmp_path):
    _ensure_built(EMU)
    check_two_passes_in_flight(EMU, tmp_path)


@pytest.mark.gpu
def test_hip_two_passes_in_flight(tmp_path):
    _ensure_built(HIP)
    check_two_passes_in_flight(HIP, tmp_path)


@pytest.mark.gpu
def test_hip_batch_per_file(tmp_path):
    _ensure_built(HIP)
    check_batch(HIP, tmp_path, range(100, 108), {"RTL433_HIP_BATCH": "1"})


@pytest.mark.gpu
def test_hip_config3(tmp_path):
    _ensure_built(HIP)
    check_config3(HIP, tmp_path)


@pytest.mark.gpu
def test_hip_mixed_list(tmp_path):
    _ensure_built(HIP)
    check_mixed_list(HIP, tmp_path)


def check_analyzer(binary, tmp_path):
    """-A through the seam: the reference's pulse_analyzer runs its trial demodulation through pulse_slicer_pwm, which
    here is librtl433seam.so's (GPU slicer + the reference's own decoder_log_bitbuffer for the printout)."""
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    args = ["-r", "g001_433.92M_250k.cu8", "-A", "-R", "0"]

    def run(b):
        p = subprocess.run([b] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0
        return p.stdout.decode(), "\n".join(l for l in p.stderr.decode().splitlines() if "Detected OOK package" in l or "codes" in l
                                            or "distribution" in l or "Guessing" in l or "Attempting" in l or "flex decoder" in l)
    ref, got = run(REF), run(binary)
    assert "{52}e7a760b94372e" in ref[0] + ref[1]
    assert got == ref


def test_emu_analyzer(tmp_path):
    _ensure_built(EMU)
    check_analyzer(EMU, tmp_path)


@pytest.mark.gpu
def test_hip_analyzer(tmp_path):
    _ensure_built(HIP)
    check_analyzer(HIP, tmp_path)


REFERENCE_DSP_SYMBOLS = ("envelope_detect", "envelope_detect_nolut", "magnitude_est_cu8", "magnitude_true_cu8", "magnitude_est_cs16",
                         "magnitude_true_cs16", "baseband_low_pass_filter", "baseband_demod_FM", "baseband_demod_FM_cs16", "baseband_init",
                         "pulse_detect_create", "pulse_detect_package", "pulse_detect_fsk_classic", "pulse_detect_fsk_minmax",
                         "pulse_detect_fsk_wrap_up", "pulse_slicer_pcm", "pulse_slicer_ppm", "pulse_slicer_pwm",
                         "pulse_slicer_manchester_zerobit", "pulse_slicer_dmc", "pulse_slicer_piwm_raw", "pulse_slicer_piwm_dc",
                         "pulse_slicer_nrzs", "pulse_slicer_osv1", "pulse_slicer_rzi", "push_sdr_flow_reference")


@pytest.mark.parametrize("binary", [EMU, HIP, PLUGINS], ids=["rtl_433_emu", "rtl_433_hip", "libr433plugins"])
def test_no_reference_dsp_symbol_is_linked(binary):
    """By linkage: neither the drop-in CLI nor the plugin library bench.py and pipeline_host put behind the path DEFINES any
    function of the reference's src/baseband.c, src/pulse_detect.c, src/pulse_detect_fsk.c or src/pulse_slicer.c -- what the rest
    of the reference calls by those names is undefined in them and resolved by librtl433seam.so (-> the GPU library)."""
    if not os.path.exists(binary):
        pytest.skip(f"{os.path.relpath(binary, ROOT)} not built")
    syms = subprocess.run(["nm", binary], stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout.decode() \
        + subprocess.run(["nm", "-D", binary], stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout.decode()
    defined = {l.split()[-1] for l in syms.splitlines() if len(l.split()) == 3 and l.split()[1] in "TtDdBbWwVv"}
    undefined = {l.split()[-1] for l in syms.splitlines() if len(l.split()) == 2 and l.split()[0] == "U"}
    assert not defined & set(REFERENCE_DSP_SYMBOLS), sorted(defined & set(REFERENCE_DSP_SYMBOLS))
    assert {"pulse_detect_create", "baseband_init", "pulse_slicer_pcm"} <= undefined  # r_api.c's calls leave the binary
    needed = subprocess.run(["readelf", "-d", binary], stdout=subprocess.PIPE).stdout.decode()
    assert "librtl433seam.so" in needed


def check_pulse_dumpers(binary, tmp_path):
    """-w out.ook / out.u8 next to the JSON: package dumps in reference order, the logic bytes of every capture."""
    names = write_ook_files(tmp_path, [21, 22]) + ["long_433.92M_250k.cu8"]
    np.concatenate([synth.ook_stream(100 + k)[0] for k in range(5)]).tofile(tmp_path / "long_433.92M_250k.cu8")
    outs = {}
    for tag, b in (("ref", REF), ("got", binary)):
        args = file_args(names) + ["-F", "json", "-w", f"{tag}.u8", "-w", f"{tag}.ook"]
        outs[tag] = run_cli(b, args, tmp_path)
    assert outs["ref"] == outs["got"]
    assert (tmp_path / "ref.u8").read_bytes() == (tmp_path / "got.u8").read_bytes()
    assert (tmp_path / "ref.u8").stat().st_size == 2 * 65536 + 5 * 65536
    a, b = (tmp_path / "ref.ook").read_text().splitlines(), (tmp_path / "got.ook").read_text().splitlines()
    wall = (";received", ";timestamp", ";created")  # wall-clock lines
    assert [l for l in a if not l.startswith(wall)] == [l for l in b if not l.startswith(wall)]


def test_emu_pulse_dumpers(tmp_path):
    _ensure_built(EMU)
    check_pulse_dumpers(EMU, tmp_path)


@pytest.mark.gpu
def test_hip_pulse_dumpers(tmp_path):
    _ensure_built(HIP)
    check_pulse_dumpers(HIP, tmp_path)


@pytest.mark.parametrize("binary", [pytest.param(EMU_REFFLOW, id="emu"), pytest.param(HIP_REFFLOW, id="hip", marks=pytest.mark.gpu)])
def test_reference_flow_over_the_function_seam(binary, tmp_path):
    """The reference's own r_flow.c (and everything above it) linked with librtl433seam.so instead of src/baseband.c,
    src/pulse_detect.c, src/pulse_detect_fsk.c and src/pulse_slicer.c: envelope, both filters, pulse_detect_package -- called
    again and again per frame, resumable -- and the slicers all come from the GPU library, call by call.  Same output as the
    stock binary: the known answer with its levels, a flex decoder's rows, an FSK capture through both FSK detectors."""
    _ensure_built(binary)
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    args = ["-r", "g001_433.92M_250k.cu8", "-R", "169"] + FLEX[:2] + ["-F", "json", "-M", "level", "-M", "protocol"]
    ref = run_cli(REF, args, tmp_path)
    got = run_cli(binary, args, tmp_path)
    assert ref.splitlines()[0].strip() == KAT_LINE
    assert got == ref
    # (on the emulator every pulse_detect_package call is a "launch" of 64 fibers: a quarter of the samples there, the same five messages)
    n_f, n_o = (16000, 12000) if binary == EMU_REFFLOW else (60000, 40000)
    synth.fsk_stream_cu8(4, n_f, n_bursts=2, nbits=120, gap=5000).tofile(tmp_path / "f_433.92M_250k.cu8")
    synth.ook_stream(12, n_o)[0].tofile(tmp_path / "o_433.92M_250k.cu8")
    # (both FSK detectors on the GPU; the emulator takes the default one here -- tests/test_seam_lib.py drives both call by call)
    for extra in ([],) if binary == EMU_REFFLOW else ([], ["-Y", "minmax"]):
        args = ["-r", "f_433.92M_250k.cu8", "-r", "o_433.92M_250k.cu8", "-X", "n=fpcm,m=FSK_PCM,s=100,l=100,r=2000"] + FLEX + extra \
            + ["-F", "json", "-M", "level", "-K", "FILE"]
        ref = run_cli(REF, args, tmp_path)
        assert ref.count("\n") >= 5
        assert run_cli(binary, args, tmp_path) == ref
    # run_ook_demods / run_fsk_demods (include/r_api.h:50-52) are the seam's too: the reference's r_flow.c hands every package
    # to librtl433seam.so, which runs ALL registered decoders' slicers over it in one launch and replays them in the
    # reference's order -- with every default decoder registered (335 slicers per package; one launch and one round trip
    # EACH through the pulse_slicer_* exports, which is what this binary did before)
    syms = subprocess.run(["nm", binary], stdout=subprocess.PIPE).stdout.decode()
    assert " U run_ook_demods" in syms and " U run_fsk_demods" in syms
    import time
    args = ["-r", "o_433.92M_250k.cu8", "-r", "g001_433.92M_250k.cu8", "-r", "f_433.92M_250k.cu8"] + FLEX + ["-F", "json", "-M", "level", "-K", "FILE"]
    ref = run_cli(REF, args, tmp_path)  # (no -R: every default decoder is registered, the three generic ones behind them)
    t0 = time.perf_counter()
    got = run_cli(binary, args, tmp_path)
    dt = time.perf_counter() - t0
    assert got == ref and ref.count("\n") >= 5
    out = os.path.join(ROOT, "gpurun_out")
    if binary == HIP_REFFLOW and os.path.isdir(out):
        with open(os.path.join(out, "refflow_seam_timing.txt"), "a") as f:
            f.write(f"rtl_433_refflow_hip, three captures, all default decoders, fan-out through the seam's run_ook_demods / run_fsk_demods: {dt * 1e3:.0f} ms\n")


OPTION_SETS = [
    ["-F", "json", "-M", "level", "-M", "stats:2:3600"],  # (a short report interval is a wall-clock event: not comparable)
    ["-F", "json", "-M", "level", "-Y", "autolevel", "-Y", "magest"],
    ["-F", "json", "-M", "level", "-Y", "level=-5", "-Y", "minlevel=-20", "-Y", "minsnr=6"],
    ["-F", "csv", "-M", "level"],
    ["-F", "json", "-M", "level", "-s", "1000k"],
    ["-F", "json", "-M", "level", "-Y", "classic"],
    ["-F", "json", "-M", "level", "-Y", "minmax", "-Y", "ampest"],
    ["-F", "json", "-M", "level", "-Y", "filter=0.3", "-M", "noise:1"],
    ["-F", "log", "-F", "json", "-M", "level", "-vv"],
]


def _strip_banner(text):
    return "\n".join(l for l in text.splitlines() if not l.startswith("rtl_433 version") and "Use \"-F log\"" not in l)


@pytest.mark.parametrize("binary", [pytest.param(EMU, id="emu"), pytest.param(HIP, id="hip", marks=pytest.mark.gpu)])
def test_cli_option_matrix(binary, tmp_path):
    """The flow options a user can reach from the command line -- levels, estimators, FSK detector choice, filters, sample rate
    override, statistics, verbosity, output formats -- on an OOK, an FSK and the known-answer capture: stdout AND stderr equal the
    stock binary's (stderr carries the -vv messages of the flow, e.g. the per-package levels)."""
    _ensure_built(binary)
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    synth.ook_stream(12, 40000)[0].tofile(tmp_path / "o_433.92M_250k.cu8")
    synth.fsk_stream_cu8(4, 60000, n_bursts=2, nbits=120, gap=5000).tofile(tmp_path / "f_433.92M_250k.cu8")
    base = ["-r", "g001_433.92M_250k.cu8", "-r", "o_433.92M_250k.cu8", "-r", "f_433.92M_250k.cu8", "-R", "169", "-X", FLEX[1],
            "-X", "n=fpcm,m=FSK_PCM,s=100,l=100,r=2000"]
    for opts in OPTION_SETS:
        outs = []
        for b in (REF, binary):
            p = subprocess.run([b] + base + opts, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
            assert p.returncode == 0, (opts, p.stderr.decode(errors="replace")[-1000:])
            outs.append((p.stdout.decode(), _strip_banner(p.stderr.decode(errors="replace"))))
        assert outs[0][0] == outs[1][0], opts
        assert outs[0][1] == outs[1][1], opts
        assert outs[0][0].count("\n") >= 4, opts


@pytest.mark.parametrize("binary", [pytest.param(EMU, id="emu"), pytest.param(HIP, id="hip", marks=pytest.mark.gpu)])
def test_log_printing_outputs_keep_the_file_order(binary, tmp_path):
    """-F kv prints the file loop's own messages ("Test mode active. Reading samples from file: ...") in the same stream as
    the events: the flow then runs one pass per file so that both come out in the reference's order.  (-F kv lays its pairs
    out for a terminal width that an ioctl on a non-terminal leaves uninitialised in the reference -- src/output_file.c takes
    whatever the stack held: the separator lines are left out and the pairs compared as one stream of words, whatever line
    they landed on.)"""
    _ensure_built(binary)
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    synth.ook_stream(12, 40000)[0].tofile(tmp_path / "o_433.92M_250k.cu8")
    args = ["-r", "g001_433.92M_250k.cu8", "-r", "o_433.92M_250k.cu8", "-r", "g001_433.92M_250k.cu8", "-R", "169", "-X", FLEX[1], "-F", "kv", "-M", "level"]

    def words(b):
        out = run_cli(b, args, tmp_path)
        return " ".join(w for l in out.splitlines() if l.strip(" _") for w in l.split())  # (separator lines: any width, down to a single "_")
    ref = words(REF)
    assert ref.count("Reading samples from file") == 3
    assert words(binary) == ref


DUMPER_SETS = [["x.cu8"], ["x.cs16"], ["x.cs8"], ["x.cf32"], ["x.i.f32"], ["x.q.f32"], ["x.am.s16", "x.am.f32"], ["x.fm.s16", "x.fm.f32"], ["x.logic.u8"]]


@pytest.mark.parametrize("binary", [pytest.param(EMU, id="emu"), pytest.param(HIP, id="hip", marks=pytest.mark.gpu)])
def test_every_sample_dumper(binary, tmp_path):
    """-W for every sample format of include/fileformat.h, from cu8 and from cs16 input: the files equal the stock binary's
    byte for byte.  (One IQ conversion per run: in the reference an IQ dumper listed before fm.s16 clobbers it)"""
    _ensure_built(binary)
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    synth.fsk_stream_cu8(4, 40000, n_bursts=1, nbits=120).tofile(tmp_path / "f_433.92M_250k.cu8")
    synth.fsk_stream_cs16(3, 30000, n_bursts=1, lead_in=3000).tofile(tmp_path / "c_868M_1024k.cs16")
    for inputs in (["g001_433.92M_250k.cu8", "f_433.92M_250k.cu8"], ["c_868M_1024k.cs16"]):
        for dumpers in DUMPER_SETS:
            got = {}
            for who, b in (("ref", REF), ("new", binary)):
                d = tmp_path / who
                shutil.rmtree(d, ignore_errors=True)
                d.mkdir()
                args = sum((["-r", "../" + f] for f in inputs), []) + ["-R", "169"] + sum((["-W", x] for x in dumpers), []) + ["-F", "json"]
                out = run_cli(b, args, d)
                got[who] = (out, {x: (d / x).read_bytes() for x in dumpers})
            assert got["ref"][0] == got["new"][0], (inputs, dumpers)
            for x in dumpers:
                assert len(got["ref"][1][x]) > 0, (inputs, x)
                assert got["ref"][1][x] == got["new"][1][x], (inputs, x)


@pytest.mark.parametrize("binary", [pytest.param(EMU, id="emu"), pytest.param(HIP, id="hip", marks=pytest.mark.gpu)])
def test_am_fm_sample_files_as_input(binary, tmp_path):
    """-r x.am.s16 / -r x.fm.s16 (file_info S16_AM / S16_FM, src/r_flow.c:212-225): the stock binary writes the dumps, then both
    binaries read them back in a list that also holds the cu8 captures they came from; events, levels and the dumps written
    from such input equal the stock binary's."""
    _ensure_built(binary)
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    names = write_ook_files(tmp_path, [21, 22])
    synth.fsk_stream_cu8(5, 40000, n_bursts=2, nbits=100).tofile(tmp_path / "f_433.92M_250k.cu8")
    made = []
    for src in ["g001_433.92M_250k.cu8", names[0], "f_433.92M_250k.cu8"]:
        stem = src[:-4]
        run_cli(REF, ["-r", src, "-W", stem + ".am.s16", "-W", stem + ".fm.s16"], tmp_path)
        made += [stem + ".am.s16", stem + ".fm.s16"]
        assert (tmp_path / made[-2]).stat().st_size > 0 and (tmp_path / made[-1]).stat().st_size > 0
    order = [made[0], "g001_433.92M_250k.cu8", made[2], made[3], names[1], made[5], made[4], made[1]]
    got = {}
    for who, b in (("ref", REF), ("new", binary)):
        d = tmp_path / who
        d.mkdir()
        args = sum((["-r", "../" + f] for f in order), []) + FLEX + ["-X", "n=fmc,m=FSK_MC_ZEROBIT,s=50,l=50,r=1200", "-F", "json",
                "-M", "level", "-M", "bits", "-K", "FILE", "-W", "again.am.s16", "-W", "again.fm.s16"]
        got[who] = (run_cli(b, args, d), (d / "again.am.s16").read_bytes(), (d / "again.fm.s16").read_bytes())
    assert got["ref"][0].count("\n") >= 4 and "am.s16" in got["ref"][0]
    assert got["new"][0] == got["ref"][0]
    assert got["new"][1] == got["ref"][1] and len(got["ref"][1]) > 0
    assert got["new"][2] == got["ref"][2]


@pytest.mark.parametrize("binary", [pytest.param(EMU, id="emu"), pytest.param(HIP, id="hip", marks=pytest.mark.gpu)])
def test_sample_grabber(binary, tmp_path):
    """-S all | unknown | known | undecoded and the SigMF variant over a list of files: the same g###_<freq>M_<rate>k files with the same
    bytes (the first grab reaches back across a file boundary: the ring's history), the same messages; also with one GPU pass
    per file, where the history comes from the pass before."""
    _ensure_built(binary)
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    synth.ook_stream(12, 40000)[0].tofile(tmp_path / "o_433.92M_250k.cu8")
    synth.ook_stream(21, 400000)[0].tofile(tmp_path / "big_433.92M_250k.cu8")
    files = ["g001_433.92M_250k.cu8", "o_433.92M_250k.cu8", "big_433.92M_250k.cu8", "g001_433.92M_250k.cu8"]

    def run(b, mode, env=None):
        d = tmp_path / "work"
        shutil.rmtree(d, ignore_errors=True)
        d.mkdir()
        e = dict(os.environ)
        e.update(env or {})
        p = subprocess.run([b] + sum((["-r", "../" + f] for f in files), []) + ["-R", "169", "-S", mode, "-F", "json"], cwd=d, env=e,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
        assert p.returncode == 0, p.stderr.decode(errors="replace")[-1000:]
        notes = [l for l in p.stderr.decode(errors="replace").splitlines() if l.startswith("***") or l.startswith("Signal bigger")]
        return p.stdout.decode(), notes, {f: (d / f).read_bytes() for f in sorted(os.listdir(d))}

    for mode in ("all", "unknown", "known", "undecoded", "sigmf:all"):
        ref = run(REF, mode)
        if mode in ("all", "undecoded", "sigmf:all"):
            assert len(ref[2]) >= 2 and len(ref[1]) >= 2
        for env in (None, {"RTL433_HIP_BATCH": "1"}):
            got = run(binary, mode, env)
            assert got[0] == ref[0], (mode, env)
            assert got[1] == ref[1], (mode, env)
            assert sorted(got[2]) == sorted(ref[2]), (mode, env)
            for f in ref[2]:
                assert got[2][f] == ref[2][f], (mode, env, f)


@pytest.mark.parametrize("binary", [pytest.param(EMU, id="emu"), pytest.param(HIP, id="hip", marks=pytest.mark.gpu)])
def test_stop_after_successful_events(binary, tmp_path):
    """-E quit / -E hop: the file loop acts on the event count of every push (src/rtl_433.c:1136-1143), so the flow answers
    every push at once (the capture so far is run again, its newest packages replayed): the process stops where the stock
    binary stops -- after the first frame with an event, then one frame of every further file -- and sample dumps written
    along the way are the same bytes."""
    _ensure_built(binary)
    a, b, c = synth.ook_stream(12, 131072)[0], synth.noise_cu8(5, 131072, 2.0), synth.ook_stream(21, 131072)[0]
    np.concatenate([a, b, c, a]).tofile(tmp_path / "multi_433.92M_250k.cu8")  # events in frames 0, 2 and 3
    synth.ook_stream(12, 40000)[0].tofile(tmp_path / "o_433.92M_250k.cu8")
    shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), tmp_path / "g001_433.92M_250k.cu8")
    lists = [["multi_433.92M_250k.cu8", "o_433.92M_250k.cu8"], ["o_433.92M_250k.cu8", "multi_433.92M_250k.cu8", "g001_433.92M_250k.cu8"]]
    all_events = run_cli(REF, file_args(lists[0]) + ["-R", "169"] + FLEX + ["-F", "json", "-M", "level"], tmp_path)
    for files in lists:
        for opts in (["-E", "quit"], ["-E", "hop"], ["-E", "quit", "-Y", "autolevel", "-M", "stats:2:3600"]):
            args = file_args(files) + ["-R", "169"] + FLEX + ["-F", "json", "-M", "level"] + opts
            ref = run_cli(REF, args, tmp_path)
            assert run_cli(binary, args, tmp_path) == ref, (files, opts)
            if files is lists[0] and opts == ["-E", "quit"]:
                assert 0 < ref.count("\n") < all_events.count("\n")  # it did stop early
    # dumpers fed frame by frame while frames are replayed again
    got = {}
    for who, bin_ in (("ref", REF), ("new", binary)):
        d = tmp_path / who
        d.mkdir()
        dumps = ["x.logic.u8", "x.am.s16", "x.fm.f32", "x.cs16"]
        out = run_cli(bin_, ["-r", "../multi_433.92M_250k.cu8", "-r", "../o_433.92M_250k.cu8", "-R", "169"] + FLEX + ["-F", "json", "-E", "hop"]
                      + sum((["-W", x] for x in dumps), []), d)
        got[who] = (out, {x: (d / x).read_bytes() for x in dumps})
    assert got["ref"] == got["new"]


def test_the_flow_renders_reports_through_the_library():
    """-A and -w x.ook / x.vcd in the drop-in go through the GPU library (r433_batch_analyze + r433_analysis_text,
    r433_pulse_text_dump, r433_pulse_vcd), not through the reference's host pulse_analyzer / pulse_data_dump /
    pulse_data_print_vcd: the CLI tests above (check_analyzer, check_pulse_dumpers) compare the library's text with the stock
    binary's.  (-S undecoded still asks the reference's pulse_analyzer_check, a yes / no heuristic outside SURVEY 8f.)"""
    import re
    src = open(os.path.join(ROOT, "dropin", "r_flow_hip.c")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name in ("pulse_analyzer(", "pulse_data_dump(", "pulse_data_print_vcd("):
        assert name not in code, name
    for name in ("r433_batch_analyze(", "r433_analysis_text(", "r433_pulse_text_dump(", "r433_pulse_vcd("):
        assert name in code, name
