"""bench.py -- IQ Msamples/s end-to-end (cu8 -> decoder callbacks) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` is honoured either way: started by torchrun the process checks WORLD_SIZE == N and stops otherwise; started
plainly with N > 1 it re-executes itself under torch.distributed.run with N ranks on 127.0.0.1.  `n_gpus` in the line is the
number of ranks the process group really has (RCCL, one distinct GPU each: checked).

Workloads (BASELINE.json `configs`, recipes in SURVEY.md 8d):

  --config 2 (default)  configs[1]: batches of 1024 synthetic 250 kS/s cu8 OOK bursts x 65536 samples, all 335 default
                        decoders fanned out.  One step = one pass of the hot path over --batches (8) such batches PER GPU,
                        submitted together: the 8192 captures are resident in HBM when the timed region starts (--from-host:
                        they start in pinned HOST memory and cross PCIe inside every step), run through the detection pass
                        (k_wave, IQ -> packages; one grid, so the scheduler fills the SIMDs a finished capture leaves idle with
                        the next batch's captures), the slicer fan-out (with the device-side pre-filter), the record copy back
                        to pinned host memory and the ordered replay of every bitbuffer into the REAL decode_fn of the
                        reference's 335 default decoders (dropin/_build/libr433plugins.so), whose JSON lines are the
                        output.  Three DISTINCT sets of
                        batches rotate (no step re-reads the input of the step before it); steps are software-pipelined over
                        three engines / HIP streams.  Weak scaling: every rank has its own batches, the only collective is
                        the gather of per-rank records to rank 0.
  --config 4            configs[3]: ONE fixed list of 65536 captures (seeds 0..65535), sharded contiguously over the ranks
                        (65536/N each), launches of --launch captures; a step = one pass over the whole list.  Strong scaling.
  --config 3            configs[2]: one 1024 kS/s cs16 FSK stream of 64 Mi samples, min/max detector, Manchester decoders;
                        the stream is spread over the chip by verified cuts (r433_batch_set_split).
  --config 5            configs[4]: one 2 MS/s cu8 stream, OOK + FSK bursts over a stepping noise floor, -Y autolevel and a
                        -Y filter, all decoders; reports the latency-per-burst histogram.

`value` (config 2) is whole-job samples / wall time from IQ samples resident in HBM to decoded events (JSON lines) on the
host; the rate of the same pipeline fed from pinned host memory, every step's H2D copy inside the timed region (what
`value` was in rounds 1-4; the link's 18.7 ms per GiB is its floor), is measured in the same process and reported next to
it (`pcie_inclusive`), as are the unmodified reference with the same real decoders on one core (`cpu_baseline`; its JSON
lines for the WHOLE last step of the timed region are compared with the GPU path's by sha256: `parity_detail`), the drop-in
CLI and the C pipeline host over the files of that step (`dropin`), the reference with a checksum decode_fn on one core and on all cores as independent processes
(every bitbuffer checked: `cpu_baseline_checksum_decode_fn`, `cpu_baseline_nproc`), one-pass variants of the replay
(`real_decoders`) and the single-stream workloads at their full sizes (`other_configs`: configs[2] and configs[4]).
Configs 3, 4 and 5 keep their inputs resident (their lines say so in `data`).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# Eight hardware queues instead of the runtime's four (read once, at the process's first HIP call -- before torch is imported):
# three engines have six streams of their own, and streams that share a hardware queue run in submission order -- the sizing pass of
# one engine sat for 17 ms behind the 1 GiB input copy of another (rtl_433_amd/csrc/host_api.cpp, profiles/r06_hw_queues.txt)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
def _newest(name):  # the committed counter passes of the latest round that has them
    for rnd in ("r06", "r05", "r04"):
        p = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", f"r06_{name}")


PMC_TRAFFIC = _newest("pmc_traffic.json")  # separate rocprofv3 --pmc passes (tools/pmc_traffic.py)
PMC_ISSUE = _newest("pmc_issue.json")      # ... and the instruction counters of the same build (tools/pmc_issue.py)


def _issue_note():
    """What `bound by instruction issue` means in numbers: the committed SQ counter pass of this build (not this run)."""
    try:
        return json.load(open(PMC_ISSUE))
    except Exception:
        return None


ISSUE_NOTE = _issue_note()
PMC_SLICE = _newest("pmc_slice.json")      # ... and of the decoder fan-out (R433_PMC_WHAT=slice tools/pmc_issue.py)


def slicers_note(live_ms):
    """k_slice beside the detection pass: its time in THIS run (sizing + placing of one pass, the kernel alone on the device) and what
    the committed counter pass of this build says about it"""
    out = {"ms": live_ms, "ms_what": "sizing pass + placing pass of one pass over the step's packages x all decoders, HIP events, the engine alone on the device"}
    try:
        s = json.load(open(PMC_SLICE))["derived"]["summary"]
        out.update({k: s[k] for k in ("wave_instr_per_pulse", "simd_ipc", "waves_per_simd", "packages", "pulses")})
        out["counts_from"] = os.path.basename(PMC_SLICE) + " (a counter pass of this build, not of this run; without the pre-filter)"
    except Exception as e:
        out["counts_from"] = f"no counter pass: {e}"
    return out
SIMDS, SHADER_CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, nominal clock (MI355X_MICROARCH.md)


def issue_floor_ms(valu_scale=1.0):
    """The roofline that binds the detection pass: its VALU wave-instructions (committed SQ counter pass of this build, both
    roles) at 2 clocks per wave64 VALU instruction over 1024 SIMDs -- the time the pass would take if every SIMD issued a
    VALU instruction whenever it could.  -> (ms, where the counts come from) or (None, why not)"""
    try:
        d = ISSUE_NOTE["derived"]
        valu = sum(d[r]["wave_instructions"]["valu"] for r in ("producers", "consumers", "run_again_or_pairs") if r in d)
        return valu * valu_scale * 2.0 / (SIMDS * SHADER_CLOCK_HZ) * 1e3, os.path.basename(PMC_ISSUE)
    except Exception as e:
        return None, f"no counter pass: {e}"


class Backend:
    """Where the engines run.  "hip" (the only form that measures anything): librtl433hip.so on cuda:<local_rank>, ranks over
    RCCL ("nccl").  "emu": TEST ONLY (tests/test_bench_launch.py) -- the same kernel sources on the CPU wave emulator of the
    test suite, ranks over gloo, so that the launcher, the sharding and the gather of THIS file run where there is no GPU; its
    line says so in `backend` and its numbers mean nothing."""

    def __init__(self, name):
        import contextlib
        import torch
        self.name, self.torch, self.null = name, torch, contextlib.nullcontext()
        self.emu = name == "emu"
        if self.emu:
            from tests.emu import host
            self.library = host.emu_lib()
        else:
            self.library = None  # BatchEngine -> _lib.lib(): raises when librtl433hip.so is missing
            self.hip = C.CDLL("libamdhip64.so")

    def device(self, local_rank):
        return self.torch.device("cpu") if self.emu else self.torch.device("cuda", local_rank)

    def set_device(self, local_rank):
        if not self.emu:
            self.torch.cuda.set_device(local_rank)

    def sync(self):
        if not self.emu:
            self.torch.cuda.synchronize()

    def resident(self, host_array):
        t = self.torch.from_numpy(host_array)
        return t if self.emu else t.cuda()

    def pinned(self, host_array):
        t = self.torch.from_numpy(host_array)
        return t if self.emu else t.pin_memory()

    def stream(self):
        """A HIP stream made by hipStreamCreateWithFlags(hipStreamNonBlocking), handed to torch as an ExternalStream.  NOT a
        stream of torch's pool: a host -> device copy on a pool stream (hipStreamCreateWithPriority) serialises with every
        device -> host copy in flight -- 1 GiB in + 200 MiB back take 22.3 ms at once on pool streams, 18.7 ms (= the input
        alone: the link is full duplex) on plain ones (tools/pcie_probe.py, profiles/r05_pcie_duplex.txt)."""
        if self.emu:
            return None
        st = C.c_void_p()
        if self.hip.hipStreamCreateWithFlags(C.byref(st), 1) != 0:
            raise RuntimeError("hipStreamCreateWithFlags failed")
        return self.torch.cuda.ExternalStream(st.value)

    def on(self, st):
        return self.null if st is None else self.torch.cuda.stream(st)

    def handle(self, st):
        return 0 if st is None else st.cuda_stream


BK = None  # set by main()
METRIC = "IQ Msamples/sec end-to-end (cu8 -> decoded events)"


def pmc_traffic(key, alg_bytes, det_s):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (FETCH_SIZE corrected as the
    microarchitecture guide prescribes, + WRITE_SIZE) as GB/s over the kernel time of THIS run; only when the pass was
    taken on this very workload."""
    try:
        allk = json.load(open(PMC_TRAFFIC))
        d = allk.get(key)
        if d is None or d["algorithmic_bytes_per_launch"] != alg_bytes:  # the same kernel on a launch of the same size
            d = next(v for k, v in allk.items() if k in ("config2", "config4") and key in ("config2", "config4")
                     and v["algorithmic_bytes_per_launch"] == alg_bytes)
        return round(d["hbm_bytes_per_launch"] / det_s / 1e9, 2)
    except Exception:
        return None


# ---------------------------------------------------------------- synthetic input

CAPTURE_SAMPLES = 65536  # samples per config-2 / config-4 capture (--capture-samples: the emulator test shortens it)


def _synth_one(seed):
    """Capture `seed` of the config-2 workload: the SURVEY 8d recipe (one OOK burst, family PWM / PPM / Manchester, random
    payload) -- and for every third seed a transmission of a REAL protocol with a frame its decoder accepts
    (rtl_433_amd/protocols.py: twelve protocols fit a 65536-sample capture), so that the decoders behind the path have
    something to say."""
    from rtl_433_amd import synth
    if CAPTURE_SAMPLES != 65536:
        return synth.ook_stream(seed, CAPTURE_SAMPLES)[0]
    if seed % 3 == 2:
        from rtl_433_amd import protocols
        return protocols.bench_capture(seed)[0]
    return synth.ook_stream(seed)[0]


def ook_batches(seed0, n, procs=1):
    """n config-2 captures (seeds seed0 .. seed0+n-1) as one [n, 131072] uint8 array; generated by `procs` processes."""
    if procs <= 1 or n < 256:
        return np.stack([_synth_one(s) for s in range(seed0, seed0 + n)])
    import multiprocessing as mp
    with mp.get_context("fork").Pool(procs) as pool:
        rows = pool.map(_synth_one, range(seed0, seed0 + n), chunksize=64)
    return np.stack(rows)


# ---------------------------------------------------------------- CPU legs (the checker: oracle/_ref, never the product)

_G = {}


def _ref_worker_init():
    from oracle import pyoracle as po
    ref = po.Ref(record=False)
    ref.set_digest_mode(0)
    _G["ref"] = ref


def _ref_worker_run(bounds):
    ref, host = _G["ref"], _G["host"]
    for s in range(bounds[0], bounds[1]):
        ref.run(host[s], 2, 250000, 433920000, fpdm=0, stream_index=s)
    return bounds[1] - bounds[0]


def cpu_baseline_config2(host_iq, devs_expected, gpu_digest, gpu_events, reps=3):
    """The unmodified reference (oracle/_ref) on the same batch, one thread, with a decode_fn that does the same
    checksum work as the GPU leg's plugin; then the same over all cores as independent processes."""
    from oracle import pyoracle as po
    n_streams, n_samples = host_iq.shape[0], host_iq.shape[1] // 2
    if not po.have_ref():  # the restatement, if the prebuilt reference did not travel
        cfg = po.default_flow_cfg(2, 250000)
        t0 = time.perf_counter()
        dg, nev, base = 0, 0, 0
        for s in range(n_streams):
            o = po.oracle_flow(host_iq[s], devs_expected, cfg, stream_index=s, pkg_base=base)
            base += o["n_packages"]
            d, c = po.events_digest2(o["events"])
            dg = (dg + d) & 0xFFFFFFFFFFFFFFFF
            nev += c
        best = time.perf_counter() - t0
        parity = "digest-match" if (dg == gpu_digest and nev == gpu_events) else f"MISMATCH cpu {dg}/{nev} gpu {gpu_digest}/{gpu_events}"
        return dict(value=round(n_streams * n_samples / best / 1e6, 2), unit="Msamples/s", cores=1, kind="port",
                    sample=f"the same {n_streams} x {n_samples}-sample batch, oracle restatement, one thread"), None, parity
    ref = po.Ref(record=False)
    ref.set_digest_mode(2)
    best = None
    for _ in range(reps):
        ref.clear()
        t0 = time.perf_counter()
        for s in range(n_streams):
            ref.run(host_iq[s], 2, 250000, 433920000, fpdm=0, stream_index=s)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    dg, nev = ref.digest2(), ref.digest()[1]
    ref.close()
    parity = "digest-match" if (dg == gpu_digest and nev == gpu_events) else f"MISMATCH cpu {dg}/{nev} gpu {gpu_digest}/{gpu_events}"
    one = dict(value=round(n_streams * n_samples / best / 1e6, 2), unit="Msamples/s", cores=1, kind="reference",
               sample=f"the same {n_streams} x {n_samples}-sample batch, all {len(devs_expected)} default decoders registered with a "
                      f"checksum decode_fn, best of {reps}, single thread")
    # N independent reference processes over the cores of this host (process-level scaling: not a reference feature)
    nproc = os.cpu_count() or 1
    many = None
    try:
        import multiprocessing as mp
        from rtl_433_amd import shard
        _G["host"] = host_iq
        bounds = shard.partition(n_streams, nproc)
        jobs = [(bounds[i], bounds[i + 1]) for i in range(nproc) if bounds[i + 1] > bounds[i]]
        with mp.get_context("fork").Pool(len(jobs), initializer=_ref_worker_init) as pool:
            pool.map(_ref_worker_run, jobs)  # warm: library loaded, decoders registered in every process
            bestn = None
            for _ in range(reps):
                t0 = time.perf_counter()
                pool.map(_ref_worker_run, jobs, chunksize=1)
                dt = time.perf_counter() - t0
                bestn = dt if bestn is None else min(bestn, dt)
        many = dict(value=round(n_streams * n_samples / bestn / 1e6, 2), unit="Msamples/s", cores=min(nproc, cpu_quota()), processes=len(jobs),
                    kind="reference", sample=f"the same batch split contiguously over {len(jobs)} independent reference processes "
                                             f"(nproc = {nproc}, of which the container's CPU quota grants {cpu_quota()}), best of {reps}; "
                                             "process-level scaling, the reference itself is single-threaded")
    except Exception as e:
        many = dict(error=str(e))
    return one, many, parity


def cpu_baseline_real_decoders(host_iq, reps=1, what="the inputs of the LAST step of the timed region"):
    """The unmodified reference (oracle/_ref) with its REAL decoders over the batch, one thread; what the decoders report
    comes back as the JSON lines of the reference's own printer (the text the GPU path's replay is compared with)."""
    from oracle import pyoracle as po
    if not po.have_ref():
        return None, None
    n_streams, n_samples = host_iq.shape[0], host_iq.shape[1] // 2
    ref = po.Ref(call_real=True, record=False)
    ref.set_digest_mode(0)
    ref.text_mode(True)
    best, text = None, b""
    for _ in range(reps):
        ref.clear()
        ref.take_text()
        t0 = time.perf_counter()
        for s in range(n_streams):
            ref.run(host_iq[s], 2, 250000, 433920000, fpdm=0, stream_index=s)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        text = ref.take_text()
    ref.close()
    return dict(value=round(n_streams * n_samples / best / 1e6, 2), unit="Msamples/s", cores=1, kind="reference",
                sample=f"{n_streams} x {n_samples}-sample captures ({what}), all 335 default decoders with their real decode_fn, "
                       f"best of {reps}, single thread (the reference has no other)"), text


def other_configs_summary(args):
    """configs[2] and configs[4] (one long cs16 FSK stream; one 2 MS/s mixed stream with -Y autolevel) at BASELINE's full
    sizes in the same process, the reference's real decoders behind them: their detection time, roofline fraction, the JSON
    lines of the decoders against the unmodified reference's over the whole stream, the checksum of every bitbuffer, and for
    configs[4] the latency-per-burst histogram; then configs[3] (the list of 65536 captures) on this one GPU."""
    out = {}
    for cfg_no, n_samples in ((3, 64 << 20), (5, 256 << 20)):
        a = argparse.Namespace(**vars(args))
        a.config, a.stream_samples, a.steps, a.warmup, a.quick, a.no_cpu_baseline = cfg_no, n_samples, 3, 1, True, True
        r = run_stream(a, dict(rank=0, world=1, local_rank=0, dist=None), parity_prefix=True)
        out[f"config{cfg_no}"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "roofline", "breakdown_ms", "packages_per_step", "bitbuffers_per_step",
                                                       "decoded_messages_per_step", "decoders_behind_the_path", "parity", "parity_detail", "cpu_baseline",
                                                       "latency_per_burst_ms") if k in r}
        out[f"config{cfg_no}"]["workload"] = r["config"]["workload"]
    # configs[3] at N = 1: the one list of 65536 captures in launches of 8192 (the multi-GPU workload on one GPU; the driver's
    # scaling run is `--config 4 --gpus N`)
    try:
        a = argparse.Namespace(**vars(args))
        a.config, a.list_len, a.launch, a.list_distinct, a.steps, a.warmup, a.quick, a.no_cpu_baseline = 4, 65536, 8192, 16384, 2, 1, False, False
        r = run_batched(a, dict(rank=0, world=1, local_rank=0, dist=None))
        out["config4"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "roofline", "breakdown_ms", "packages_per_step", "decoded_messages_per_step",
                                             "bitbuffers_to_host_per_step", "decoders_behind_the_path", "parity", "parity_detail", "cpu_baseline") if k in r}
        out["config4"]["roofline"] = {k: v for k, v in out["config4"].get("roofline", {}).items() if k != "issue"}
        out["config4"]["workload"] = r["config"]["workload"]
    except Exception as e:
        out["config4"] = dict(error=str(e))
    return out


def real_decoders_leg(host_iq, d_iq, devs, threads, local_rank, reps=2, pipe_steps=12):
    """One-pass variants of the replay, with the reference's REAL decoders on both sides: the GPU path replays its bitbuffers
    into the decode_fn of the plugin library's decoders (dropin/_build/libr433plugins.so -- never the checker's), the CPU
    side is the unmodified reference as it is (oracle/_ref).  Some decoders keep state between calls (e.g.
    src/devices/secplus_v1.c:142), so the replay is the ordered one: a decoder stays on one thread and sees its calls in
    reference order, outputs are committed in reference order (r433_batch_dispatch_ordered).  Measured without and with the
    device-side pre-filter (r433_batch_probe_prefilter: records a decoder provably refuses on num_rows / bits_per_row[0]
    never leave the GPU; statistics unchanged)."""
    import torch

    from oracle import pyoracle as po
    from rtl_433_amd.engine import BatchEngine, flow_cfg
    if not po.have_ref():
        return None
    n_streams, n_samples = host_iq.shape[0], host_iq.shape[1] // 2
    plug = real_decoder_plugins()
    plain = plug.devices
    objs = [C.cast(p, C.POINTER(_RDevice())).contents for p in plain]
    ref = po.Ref(call_real=True, record=False)
    ref.set_digest_mode(0)

    def stats():
        return [(o.decode_events, o.decode_ok, o.decode_messages, tuple(o.decode_fails)) for o in objs]

    def zero():
        for o in objs:
            o.decode_events = o.decode_ok = o.decode_messages = 0
            for k in range(5):
                o.decode_fails[k] = 0
    eng = BatchEngine(flow_cfg(2, 250000), devs)
    if DEBUG_FLAGS:
        eng.set_debug(DEBUG_FLAGS)
    stateless = plug.stateless()  # the plugin library's own statement about its decoders (r433p_stateless)
    best, decoded, d2h, seen = {}, {}, {}, {}
    t0 = time.perf_counter()
    tables = eng.probe_prefilter(plain, helper=plug.helper_probe())
    probe_s = time.perf_counter() - t0
    for mode in ("ordered", "single", "ordered_prefilter", "ordered_prefilter_stateless"):
        eng.set_prefilter(1 if mode.startswith("ordered_prefilter") else 0)
        eng.set_stateless(stateless if mode.endswith("stateless") else None)
        for _ in range(reps + 1):
            zero()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(d_iq)
            decoded[mode] = eng.dispatch(plain, n_threads=1) if mode == "single" else eng.dispatch_ordered(plain, None, threads)
            dt = time.perf_counter() - t0
            best[mode] = dt if mode not in best else min(best[mode], dt)
        ev_bytes = len(eng.events()[0])
        d2h[mode] = ev_bytes + len(eng.packages()[0])
        seen[mode] = stats()
    eng.close()
    plug.take()
    plug.close()
    cpu_best, cpu_events = None, 0
    for _ in range(reps):
        ref.clear()
        t0 = time.perf_counter()
        ev = 0
        for s in range(n_streams):
            ev += ref.run(host_iq[s], 2, 250000, 433920000, fpdm=0, stream_index=s)["events_ok"]
        dt = time.perf_counter() - t0
        cpu_best = dt if cpu_best is None else min(cpu_best, dt)
        cpu_events = ev
    ref.close()
    rate = lambda t: round(n_streams * n_samples / t / 1e6, 2)
    return dict(gpu=dict(value=rate(best["ordered_prefilter_stateless"]), unit="Msamples/s", host_threads=threads,
                         note="one pass of ONE batch, not pipelined, inputs resident: GPU leg with the device-side pre-filter + ordered replay into the "
                              "real decode_fn of every decoder, the decoders the host declared stateless spread over the threads "
                              "(the headline is this as a three-engine pipeline over grids of 8 batches)"),
                gpu_decoders_on_one_thread_each=dict(value=rate(best["ordered_prefilter"]), unit="Msamples/s",
                                                     note="the same with every decoder's calls kept on one thread (no host knowledge about the plugins)"),
                gpu_no_prefilter=dict(value=rate(best["ordered"]), unit="Msamples/s", note="... and with every record crossing to the host"),
                gpu_single_thread=dict(value=rate(best["single"]), unit="Msamples/s", note="no pre-filter, the replay on one thread"),
                prefilter=dict(decoders_with_table=int(tables), probe_s=round(probe_s, 2),
                               d2h_bytes_per_step_before=int(d2h["ordered"]), d2h_bytes_per_step_after=int(d2h["ordered_prefilter"]),
                               statistics_equal=bool(seen["ordered"] == seen["ordered_prefilter"] == seen["single"] == seen["ordered_prefilter_stateless"])),
                cpu=dict(value=rate(cpu_best), unit="Msamples/s", cores=1, kind="reference"),
                decoded_events=dict(gpu=int(decoded["ordered_prefilter_stateless"]), gpu_no_prefilter=int(decoded["ordered"]),
                                    gpu_single_thread=int(decoded["single"]), cpu=int(cpu_events),
                                    equal=bool(decoded["ordered"] == cpu_events == decoded["single"] == decoded["ordered_prefilter"]
                                               == decoded["ordered_prefilter_stateless"])),
                sample=f"one batch: the same {n_streams} x {n_samples}-sample captures, 335 default decoders, best of {reps}")


def _RDevice():
    from rtl_433_amd import _lib
    return _lib.RDevice


def dropin_legs(host_iq, expect_json, reps=5):
    """The drop-in itself under this run's clock: the captures of one step written as `.cu8` files (tmpfs), then
      * dropin/_build/rtl_433_hip -- the reference's unmodified CLI, decoders and JSON printer over dropin/r_flow_hip.c and the
        GPU library -- `-r f1 -r f2 ... -F json -M level -K FILE`: wall time of the whole process, cold start included, and
        its output against the stock binary's (oracle/_ref/rtl_433_ref, one run) by SHA-256;
      * dropin/_build/pipeline_host_hip -- a plain C host over the C ABI and the plugin library, three engines --: wall time
        of the process and its own clock over the passes; its JSON lines against what the timed region of this run produced
        for the same captures (`expect_json`)."""
    import hashlib
    import shutil
    import subprocess
    import tempfile
    cli, stock = os.path.join(ROOT, "dropin", "_build", "rtl_433_hip"), os.path.join(ROOT, "oracle", "_ref", "rtl_433_ref")
    ph = os.path.join(ROOT, "dropin", "_build", "pipeline_host_hip")
    if not os.path.exists(cli):
        return dict(error="dropin/_build/rtl_433_hip did not travel")
    n, n_samples = host_iq.shape[0], host_iq.shape[1] // 2
    d = tempfile.mkdtemp(prefix="r433_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        names = []
        for k in range(n):
            names.append(f"s{k:05d}_433.92M_250k.cu8")
            host_iq[k].tofile(os.path.join(d, names[-1]))
        args = [a for f in names for a in ("-r", f)] + ["-F", "json", "-M", "level", "-K", "FILE"]

        def run(binary, argv, settle=0.0, env=None):
            # settle: seconds to wait first.  When a process that held gigabytes of device and pinned memory is gone, the driver
            # goes on freeing them for about a second, and a process that opens the GPU meanwhile waits for it (its hipInit takes
            # 200-300 ms instead of 80, its first allocations likewise): profiles/r06_g_cli_series.txt.  A run is timed by
            # itself; what back-to-back runs cost is reported beside it.
            if settle:
                time.sleep(settle)
            t0 = time.perf_counter()
            p = subprocess.run([binary] + argv, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            return (time.perf_counter() - t0) * 1e3, p
        out = {"files": n, "samples_per_file": n_samples, "where": d.rsplit("/", 1)[0]}
        walls, shas = [], set()
        for _ in range(reps):
            ms, p = run(cli, args, settle=1.5)
            if p.returncode != 0:
                return dict(error="rtl_433_hip: " + p.stderr.decode(errors="replace")[-300:])
            walls.append(round(ms, 1))
            shas.add(hashlib.sha256(p.stdout).hexdigest())
        b2b = []
        for _ in range(4):
            ms, p = run(cli, args)
            b2b.append(round(ms, 1))
            shas.add(hashlib.sha256(p.stdout).hexdigest())
        cli_out = {"wall_ms": walls, "median_ms": float(np.median(walls)), "max_over_min": round(max(walls) / min(walls), 2),
                   "value": round(n * n_samples / (float(np.median(walls)) * 1e-3) / 1e6, 1), "unit": "Msamples/s",
                   "wall_ms_back_to_back": b2b[1:],
                   "json_lines": p.stdout.count(b"\n"), "same_output_every_run": len(shas) == 1,
                   "note": "wall time of the whole process (start, GPU opening, file reads, passes, replay, JSON), every run by itself: 1.5 s after the "
                           "process before it has gone (the driver frees that one's device and pinned memory for about a second, and a process that opens "
                           "the GPU meanwhile waits: `wall_ms_back_to_back` are runs started at once behind another -- the 2x spread of round 5's line)"}
        # one more run with the drop-in's own stage clock on stderr (RTL433_HIP_TRACE=1): where the wall time of a run goes
        ms, p = run(cli, args, settle=1.5, env=dict(os.environ, RTL433_HIP_TRACE="1"))
        cli_out["traced_run"] = {"wall_ms": round(ms, 1), "stages": [l.replace("hip flow: ", "")[:110] for l in p.stderr.decode(errors="replace").splitlines()
                                                                    if "hip flow:" in l][:24]}
        if os.path.exists(stock):
            ms, p = run(stock, args)
            cli_out["stock_binary_ms"] = round(ms, 1)
            cli_out["json_sha256_equals_stock_binary"] = bool(p.returncode == 0 and shas == {hashlib.sha256(p.stdout).hexdigest()})
        out["rtl_433_hip"] = cli_out
        if os.path.exists(ph):
            walls, own, ok, ph_texts = [], [], True, set()
            for _ in range(3):
                ms, p = run(ph, ["-e", "3", "-b", "1024", "-p"] + names, settle=1.5)
                if p.returncode != 0:
                    out["pipeline_host_hip"] = dict(error=p.stderr.decode(errors="replace")[-300:])
                    break
                walls.append(round(ms, 1))
                tail = p.stderr.decode(errors="replace")
                at = tail.rfind("passes over")
                own.append(float(tail[at:].split(": ", 1)[1].split(" ms", 1)[0]) if at >= 0 else None)
                ok = ok and (expect_json is None or p.stdout == expect_json)
                ph_texts.add(p.stdout)
            else:
                out["pipeline_host_hip"] = {"wall_ms": walls, "median_ms": float(np.median(walls)), "passes_ms_by_its_own_clock": own,
                                            "value": round(n * n_samples / (float(np.median(walls)) * 1e-3) / 1e6, 1), "unit": "Msamples/s",
                                            "json_equals_timed_region": (bool(ok) if expect_json is not None else None),
                                            "note": "a C host over include/r433_hip.h + libr433plugins.so, three engines, passes of 1024 files, "
                                                    "pre-filter on; wall time of the whole process"}
                out["_ph_texts"] = ph_texts  # (popped by the caller: compared with the timed region's JSON once that exists)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ---------------------------------------------------------------- the software pipeline (configs 2 and 4)

EXCLUSIVE = 2    # --exclusive: 1 the engines take turns on the detection kernel, 2 on the slicer kernels as well, 3 on the whole GPU leg, 0 no turns
DEBUG_FLAGS = 0  # --debug: R433_DEBUG_* for every engine (development, A/B timing)
NAP_WAIT = 2097152  # R433_DEBUG_NAP_WAIT (include/r433_hip.h)
H2D_NAP = False     # --h2d-wait nap: the GPU leg's thread sleeps through its input copy instead of spinning (measured: no gain, 22.0-22.15 against 21.85 ms per step: the replay threads take what it frees)


class Pipeline:
    """n_eng batch engines on their own HIP streams: while the host threads replay step k's bitbuffers into the
    decoders, the GPU already works on steps k+1, k+2.  Every step is one complete pass of the hot path over its
    batch, and all K of them finish inside the timed region."""

    def __init__(self, cfg_fn, devs, rdev_arr, threads, n_eng, local_rank, on_host_leg=None, ordered=False, hooks=None):
        import torch
        from concurrent.futures import ThreadPoolExecutor

        from rtl_433_amd.engine import BatchEngine
        self.torch = torch
        self.n_eng = n_eng
        self.engines = [BatchEngine(cfg_fn(), devs, profiling=True, library=BK.library) for _ in range(n_eng)]
        for e in self.engines:  # the kernels of a step run back to back (the engines take turns on the GPU work of a pass); record
            # copies and the host replay of step k overlap the kernels of step k+1
            e.set_exclusive_detect(EXCLUSIVE)
            if DEBUG_FLAGS:
                e.set_debug(DEBUG_FLAGS)
        self.streams = [BK.stream() for _ in range(n_eng)]
        self.pool = ThreadPoolExecutor(n_eng - 1)
        self.rdev_arr = rdev_arr
        self.threads = threads
        self.local_rank = local_rank
        self.on_host_leg = on_host_leg
        self.ordered = ordered  # the replay for decoders that keep state between calls (r433_batch_dispatch_ordered)
        self.hooks = hooks      # r433_dispatch_hooks of the ordered replay (the plugins' output_render: JSON lines made on the replay threads)
        self.stagger = 0.0
        self.replay_s = []  # the library's replay call alone, per host leg
        self.leg_tm = {}
        self.h2d_trace = [] if os.environ.get("R433_BENCH_H2D_TRACE") else None  # development: when every step's input copy ran (h2d_timeline)

    def gpu_leg(self, k, src, lens=None, h2d_from=None, d_buf=None):
        BK.set_device(self.local_rank)
        e, st = self.engines[k % self.n_eng], self.streams[k % self.n_eng]
        if h2d_from is not None:  # host -> HBM over PCIe on the engine's own stream, overlapping the other engines' kernels
            with BK.on(st):
                if self.h2d_trace is not None and st is not None:
                    ev0, ev1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
                    ev0.record(st)
                    d_buf.copy_(h2d_from, non_blocking=True)
                    ev1.record(st)
                    self.h2d_trace.append((k, time.perf_counter(), ev0, ev1))
                else:
                    d_buf.copy_(h2d_from, non_blocking=True)
                if H2D_NAP and st is not None:
                    # The copy takes 18.7 ms and the library's first wait of the pass would SPIN through all of it
                    # (hipEventSynchronize spins on this stack, tools/spin_probe.py): two legs in flight = two of the 16 CPUs
                    # the box grants, taken from the decoders.  Sleep through the copy instead; the kernels' waits stay as they are.
                    done = self.torch.cuda.Event()
                    done.record(st)
                    while not done.query():
                        time.sleep(0.0004)
            src = d_buf
        return e.run(src, lens, stream=BK.handle(st)), e.timing()

    def host_leg(self, k, n_pkgs):
        e = self.engines[k % self.n_eng]
        t0 = time.perf_counter()
        if self.ordered:
            e.dispatch_ordered(self.rdev_arr, self.hooks, self.threads)
        else:
            e.dispatch(self.rdev_arr, n_threads=self.threads)
        self.replay_s.append(time.perf_counter() - t0)
        if self.on_host_leg:
            self.on_host_leg(k, e, n_pkgs)

    def run(self, n, leg_args):
        """n complete passes; up to n_eng-1 GPU legs run ahead of the host leg (in-order hand-off).  leg_args(k) gives
        the keyword arguments of step k's GPU leg."""
        from collections import deque
        det, tot, host = [], [], []
        futs, nxt, n_pkgs = deque(), 0, 0
        while nxt < min(n, self.n_eng - 1):
            if nxt and self.stagger:
                # prologue: legs that start together stay in phase (detection kernel against detection kernel) for
                # several steps; half a leg apart they interleave at once
                time.sleep(self.stagger)
            futs.append(self.pool.submit(self.gpu_leg, nxt, **leg_args(nxt)))
            nxt += 1
        for k in range(n):
            n_pkgs, tm = futs.popleft().result()
            if self.h2d_trace is not None:
                self.leg_tm[k] = (time.perf_counter(), dict(tm))
            if nxt < n:
                futs.append(self.pool.submit(self.gpu_leg, nxt, **leg_args(nxt)))
                nxt += 1
            t1 = time.perf_counter()
            self.host_leg(k, n_pkgs)
            host.append(time.perf_counter() - t1)
            det.append(tm["detect_ms"])
            tot.append(tm["total_ms"])
        return det, tot, host, n_pkgs

    def h2d_timeline(self, last=16):
        """development (R433_BENCH_H2D_TRACE=1): start / end of the last input copies on the device's clock, and the link's idle time between them"""
        if self.leg_tm:  # the parts of a GPU leg as they were beside the other engines' legs (mean over the legs of the region)
            parts = list(self.leg_tm.values())[-last:]
            sys.stderr.write("legs overlapped, mean parts: " + " ".join(f"{n[:-3]} {float(np.mean([p[1][n] for p in parts])):.2f}" for n in parts[0][1]) + "\n")
        tr = self.h2d_trace[-last:]
        if len(tr) < 2:
            return
        self.torch.cuda.synchronize()
        base = tr[0][2]
        rows = [(k, t_host - tr[0][1], base.elapsed_time(e0), base.elapsed_time(e1)) for k, t_host, e0, e1 in tr]
        busy_to = rows[0][3]
        for k, th, a, b in rows:
            got = self.leg_tm.get(k)
            parts = "" if got is None else f" | leg taken at {(got[0] - tr[0][1]) * 1e3:8.2f} ms: " + " ".join(f"{n[:-3]} {v:.2f}" for n, v in got[1].items())
            sys.stderr.write(f"h2d step {k:3d}: issued at {th * 1e3:8.2f} ms (host clock), on the device {a:8.2f} .. {b:8.2f} ms ({b - a:6.2f}), link idle before it {max(0.0, a - busy_to):6.2f} ms{parts}\n")
            busy_to = max(busy_to, b)

    def close(self):
        for e in self.engines:
            e.close()
        self.pool.shutdown()


def cpu_quota():
    """CPUs this process may really use: the affinity mask, cut by the container's CFS quota (cgroup v2 cpu.max, v1
    cfs_quota_us).  The GPU boxes of this pool show 256 logical CPUs and a quota of 16: threads beyond it only get throttled
    (nr_throttled in cpu.stat; whole replay passes stalled for 40-60 ms with 64 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cgroup_cpu_stat():
    """usage_usec / nr_throttled / throttled_usec of this container's cgroup (v2), {} where there is none"""
    try:
        return {k: int(v) for k, v in (l.split()[:2] for l in open("/sys/fs/cgroup/cpu.stat")) if k in ("usage_usec", "nr_throttled", "throttled_usec")}
    except Exception:
        return {}


def replay_threads(world):
    """host threads of the ordered replay per rank: one and a half per CPU the quota grants (they wait on memory), at most 64"""
    return max(1, min(64, (cpu_quota() * 3 // 2) // max(1, world)))


def real_decoder_plugins(flex=None):
    """The reference's decoders as plugins (dropin/_build/libr433plugins.so: built by `make -C dropin plugins` without the
    reference's DSP units, linked to librtl433seam.so).  No stand-in: the timed path never loads anything under oracle/."""
    from rtl_433_amd import plugins
    if not plugins.available():
        raise SystemExit(f"bench.py: {plugins.LIB_PATH} is missing (`make -C dropin plugins` where the reference tree is; the file travels "
                         "to the GPU box with the snapshot) -- the decoders behind the path are part of the job, there is no substitute")
    p = plugins.Plugins(flex=flex)
    p.source = "dropin/_build/libr433plugins.so"
    return p


def timed(dist, torch, fn):
    """barrier + synchronize on both sides, MAX over ranks"""
    if dist:
        dist.barrier()
    BK.sync()
    t0 = time.perf_counter()
    out = fn()
    BK.sync()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if BK.emu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def fnv64(data: bytes) -> int:
    import zlib
    return (zlib.crc32(data) << 32) | zlib.adler32(data)  # cheap 64-bit fingerprint of a record blob


# ---------------------------------------------------------------- config 2 / config 4

def run_batched(args, ctxd):
    import torch

    from rtl_433_amd import _lib, shard
    from rtl_433_amd.engine import digest_plugin_addr, flow_cfg, load_device_table, make_rdevices
    rank, world, local_rank, dist = ctxd["rank"], ctxd["world"], ctxd["local_rank"], ctxd["dist"]
    dev = BK.device(local_rank)
    strong = args.config == 4
    n_samples = CAPTURE_SAMPLES
    threads = args.threads or replay_threads(world)
    procs = max(1, min(32, (os.cpu_count() or 1) // max(1, world)))
    devs, protocols, names = load_device_table()
    ctx = _lib.DigestCtx(0, 0)
    rdev_arr, rdev_objs = make_rdevices(devs, digest_plugin_addr(), C.addressof(ctx), names, protocols)
    # The job is decoded events: the reference's REAL decoders behind the replay (dropin/_build/libr433plugins.so: the
    # reference's sources as plugins, unchanged behind r_device.decode_fn), what they report as JSON lines printed by the
    # reference's own data_print_jsons -- that text is what the ranks gather and what is compared with the reference.
    plug = real_decoder_plugins()
    assert len(plug.devices) == len(devs), "the plugin library registers another decoder set than the device table"
    stateless = plug.stateless()  # what the plugin library says about its own decoders (r433p_stateless -> r433_batch_set_stateless)

    dropin_first = None
    if strong:
        total = args.list_len
        bounds = shard.partition(total, world)
        first, n_mine = bounds[rank], bounds[rank + 1] - bounds[rank]
        launch = min(args.launch, n_mine)
        n_launches = (n_mine + launch - 1) // launch
        # this rank's shard of the one list, resident in HBM (8.6 GB in all, 1.07 GB per GPU at N = 8)
        d_all = torch.empty((n_mine, 2 * n_samples), dtype=torch.uint8, device=dev)
        host_first = None
        host_launch0 = []  # the captures of this rank's first launch on the host: their decoded JSON is compared with the reference's
        distinct = args.list_distinct or total  # (--list-distinct D: capture i of the list is seed i mod D -- the default run's short form)
        made = {}
        for c0 in range(0, n_mine, 4096):
            c1 = min(n_mine, c0 + 4096)
            s0 = (first + c0) % distinct
            if (first + c0) // distinct == (first + c1 - 1) // distinct:  # (4096 divides every D used: a piece never wraps)
                part = made.get(s0)
                if part is None:
                    part = ook_batches(s0, c1 - c0, procs)
                    if distinct < total:
                        made[s0] = part
            else:
                part = np.stack([_synth_one((first + c) % distinct) for c in range(c0, c1)])
            if c0 == 0:
                host_first = part[: min(256, c1)].copy()
            if c0 < launch:
                host_launch0.append(part[: min(c1, launch) - c0])
            d_all[c0:c1].copy_(torch.from_numpy(part))
        made.clear()
        host_launch0 = np.concatenate(host_launch0) if host_launch0 else None
        batches = [d_all[l * launch: min(n_mine, (l + 1) * launch)] for l in range(n_launches)]
        per_step = n_launches
        n_streams = launch
    else:
        n_batch = args.streams                 # captures of one config-2 batch
        n_streams = n_batch * args.batches     # captures a step submits together (one grid)
        n_rot = 3
        host_batches = [ook_batches((rank * n_rot + b) * n_streams, n_streams, procs if n_streams >= 2048 else 1) for b in range(n_rot)]
        per_step = 1
        if rank == 0 and world == 1 and not BK.emu and not args.quick and not args.no_cpu_baseline:
            # The drop-in CLI and the C pipeline host over the files of the LAST step of the timed region (what a user of
            # `rtl_433 -r` gets): processes of their own, timed before this one has touched the GPU -- no device memory, no
            # pinned memory, no replay threads beside them.  Their JSON is compared with the timed region's once that has run.
            # (It makes no difference to their wall time -- 570-600 ms at the start as at the end of the run, this box; the
            # 285 ms of profiles/r06_g_cli_series.txt are tools/cli_trace.sh's lighter captures, synth.ook_stream without the
            # protocol transmissions: a third of the replay -- but a quiet machine is the cleaner measurement.)
            try:
                dropin_first = dropin_legs(host_batches[(args.steps * per_step - 1) % len(host_batches)], None)
            except Exception as e:
                dropin_first = dict(error=str(e))
        pinned = [BK.pinned(h) for h in host_batches]
        batches = [BK.resident(h) for h in host_batches]  # the same inputs resident in HBM (`hbm_resident`)

    records = {}

    def on_host_leg(k, e, n_pkgs):
        # what leaves a rank: the JSON lines its decoders produced for this launch's captures, in capture order
        text, n_msg = plug.take()
        st = records.setdefault("acc", [0, 0, 0, 0, [], 0, 0])
        if k % per_step == 0:
            st[0], st[1], st[3], st[4], st[5], st[6] = 0, 0, 0, [], 0, 0  # a new pass: what is gathered is the last pass
        st[0] += n_pkgs
        st[1] += n_msg
        if strong:
            st[3] ^= fnv64(e.packages()[0])
        st[4].append(text)
        if strong and k % per_step == 0:
            records["launch0_text"] = text  # (of the last pass over the list: the same captures every pass)
        pk_b, _, ev_b, ev_n = e.sizes()
        st[5] += ev_n          # bitbuffers that crossed to the host (the pre-filter keeps the provably refused ones on the device)
        st[6] += pk_b + ev_b   # bytes of records copied back

    pipe = Pipeline(lambda: flow_cfg(2, 250000), devs, plug.devices, threads, max(2, args.engines), local_rank, on_host_leg, ordered=True,
                    hooks=plug.hooks() if hasattr(plug, "hooks") else None)
    for e in pipe.engines:
        e.set_stateless(stateless)
        e.probe_prefilter(plug.devices, helper=plug.helper_probe())  # records a decoder provably refuses on their head stay on the device (statistics unchanged)

    d_bufs = None if strong else [torch.empty_like(batches[0]) for _ in range(max(2, args.engines))]

    def leg_args_resident(k):
        return dict(src=batches[k % len(batches)])

    def leg_args_pcie(k):  # every step first copies its input from pinned host memory on its engine's stream
        return dict(src=None, h2d_from=pinned[k % len(pinned)], d_buf=d_bufs[k % pipe.n_eng])

    # `value` is measured with the inputs resident in HBM when the timed region starts; the PCIe-inclusive rate of the same
    # pipeline (pinned host memory -> H2D inside every step) is the secondary line `pcie_inclusive` (--from-host swaps the two)
    leg_args = leg_args_pcie if args.from_host and not strong else leg_args_resident

    gathered = {}

    def gather_records():
        """The one collective of the path: per-rank records to rank 0 (RCCL over xGMI), variable length."""
        npk, nmsg, dsum, pkh, texts, nbits, nbytes = records.get("acc", [0, 0, 0, 0, [], 0, 0])
        records["sent"] = (nbits, nbytes)
        records["last_text"] = b"".join(texts)  # what this rank's decoders said in the LAST step of the timed region
        payload = shard.pack_rank_record(first if strong else rank * n_streams, npk, nmsg, nbits, pkh if strong else nbytes, b"".join(texts))
        got = shard.gather_rank_records(payload, dist is not None, dst=0, device=dev, tail="text")
        if rank == 0:
            gathered.update(got)

    # prime every engine (buffers grow to their steady size, dispatch threads start), independent of --warmup
    for k in range(pipe.n_eng):
        pipe.host_leg(k, pipe.gpu_leg(k, **leg_args(k))[0])
    records.pop("acc", None)
    solo = [pipe.gpu_leg(0, **leg_args_resident(i))[1] for i in range(1 if BK.emu else 5)]  # the kernel alone on the device, HIP events on its stream
    solo_det_ms = float(np.mean([t["detect_ms"] for t in solo]))
    solo_tot_ms = float(np.mean([t["total_ms"] for t in solo]))
    solo_parts = {k: round(float(np.mean([t[k] for t in solo])), 3) for k in solo[0] if k != "total_ms"}
    pipe.stagger = solo_tot_ms / 1e3 / (pipe.n_eng - 1)
    pipe.run(args.warmup * per_step, leg_args)
    records.pop("acc", None)

    def timed_region():
        out = pipe.run(args.steps * per_step, leg_args)
        gather_records()  # inside the timed region: the event gather is part of the job
        return out

    pipe.replay_s.clear()
    cpu0, cg0 = os.times(), cgroup_cpu_stat()
    elapsed, (det_ms, tot_ms, disp_s, n_pkgs) = timed(dist, torch, timed_region)
    cpu1, cg1 = os.times(), cgroup_cpu_stat()
    if pipe.h2d_trace is not None:
        pipe.h2d_timeline()
    replay_ms = float(np.mean(pipe.replay_s)) * 1e3 if pipe.replay_s else 0.0

    result = None
    if rank == 0:
        per = gathered["per_rank"]
        import hashlib
        decoded_json = gathered["merged"]  # rank order = capture order: the single-process output
        gathered_json = {"bytes": len(decoded_json), "lines": decoded_json.count(b"\n"), "sha256": hashlib.sha256(decoded_json).hexdigest(),
                         "note": ("JSON lines of the last pass over the list, every rank's decoded events in capture order; the same for every N" if strong else
                                  "JSON lines of the last step, every rank's decoded events in capture order (the ranks have their own batches)")}
        tot_pk = sum(p["packages"] for p in per)
        tot_msg = sum(p["events"] for p in per)
        tot_bits = sum(p["digest"] for p in per)
        samples_per_step = args.list_len * n_samples if strong else world * n_streams * n_samples
        value = samples_per_step * args.steps / elapsed / 1e6
        live_det_ms = float(np.mean(det_ms))  # HIP events on the launching stream, over the launches of the timed region
        det_s = live_det_ms / 1e3
        alg_bytes = 2 * n_streams * n_samples  # 2 B per cu8 IQ sample, read once x the captures of one launch
        achieved = alg_bytes / det_s / 1e9
        workload = (f"configs[3]: one list of {args.list_len} independent cu8 captures x {n_samples} samples (seeds 0..{args.list_len - 1}"
                    + (f" mod {args.list_distinct}: {args.list_distinct} distinct captures, the short form of the default run; `--config 4` makes all {args.list_len} distinct" if args.list_distinct and args.list_distinct < args.list_len else "") + ") "
                    f"sharded contiguously over {world} GPU(s), launches of {n_streams} captures, all {len(devs)} default decoders, "
                    "records gathered on rank 0") if strong else \
                   (f"configs[1]: batches of {n_batch} synthetic 250 kS/s cu8 OOK bursts x {n_samples} samples (every third capture a "
                    f"protocol-valid transmission of one of 12 real protocols, the others random payloads), all {len(devs)} default -R "
                    f"decoders fanned out and their real decode_fn called; a step = {args.batches} such batches per GPU submitted together "
                    f"(one detection grid of {n_streams} captures), from IQ samples in HBM to decoded events (JSON lines) on the host")
        result = {
            "metric": METRIC, "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8",
            "data": ("synthetic (resident in HBM; one fixed list)" if strong else
                     "synthetic (in pinned host memory when the timed region starts: --from-host, every step's H2D copy is timed; three distinct inputs rotate)" if args.from_host else
                     "synthetic (resident in HBM when the timed region starts; three distinct inputs rotate; `pcie_inclusive` is the same pipeline fed from pinned host memory)"),
            "config": {"workload": workload,
                       "value_definition": ("whole-job IQ samples / wall time of the timed region, inputs RESIDENT IN HBM when the region starts, decoded events "
                                            "(JSON lines of the reference's real decoders) on the host and gathered on rank 0 when it ends.  Fixed since round 5 "
                                            "(the task's definition) and not to move again; `value_r04_definition` is the SAME pipeline of the SAME run fed from "
                                            "pinned host memory, every step's H2D copy inside the timed region: what `value` meant in rounds 1-4 (BENCH_r01..r04), "
                                            "for like-for-like reads across rounds (= `pcie_inclusive.value`)") if not strong else
                                           "whole-job IQ samples / wall time of the timed region over the one list, inputs resident in HBM, records gathered on rank 0",
                       "streams_per_launch": n_streams, "samples_per_stream": n_samples,
                       "sample_rate": 250000, "decoders": len(devs), "host_dispatch_threads": threads, "host_cpus": {"logical": os.cpu_count(), "cfs_quota": cpu_quota()},
                       "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "launches_per_step_per_gpu": per_step if strong else args.batches,
                       **({} if strong else {"captures_per_batch": n_batch, "batches_per_step_per_gpu": args.batches,
                                              "note": f"the {args.batches} launches of {n_batch} captures of a step are issued as ONE grid of {n_streams} workgroups"}),
                       "parallelism": f"captures sharded over {world} GPU(s), no data-path collective; one variable-length gather of records to rank 0"},
            "roofline": {"bound": "hbm", "kernel": "k_wave<2> (IQ -> packages): the detection pass -- for grids of 6144 captures and more a launch of producers (filters), one of consumers (detector) and a run-again launch, timed together", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": pmc_traffic("config2" if not strong else "config4", alg_bytes, det_s),
                         "issue_frac": (round(issue_floor_ms()[0] / live_det_ms, 4) if issue_floor_ms()[0] else None),
                         "issue_floor": {"ms": (round(issue_floor_ms()[0], 3) if issue_floor_ms()[0] else None), "counts_from": issue_floor_ms()[1],
                                         "formula": "VALU wave-instructions of one pass (producers + consumers + run-again) x 2 clocks per wave64 VALU "
                                                    "instruction / (1024 SIMDs x 2.4 GHz); issue_frac = that floor / the pass's measured time in THIS run "
                                                    "(k_wave_timed_region): the fraction of the roofline that binds this kernel, beside `frac` (HBM), which does not"},
                         "issue": ISSUE_NOTE,
                         "slicers": slicers_note(round(solo_parts.get("count_ms", 0.0) + solo_parts.get("write_ms", 0.0), 3)),
                         "note": "achieved = 2 B/sample x samples of one launch / the kernel's mean launch duration over the timed region (HIP events on "
                                 "the launching stream; the engines of the pipeline take turns on the kernels of a pass).  `traffic` is NOT measured in "
                                 "this run: it is the HBM byte count of the committed counter pass (profiles/, tools/pmc_run.sh) over this run's kernel "
                                 "time.  The kernel is bound by wavefront instruction issue, not by HBM: see `issue` (DESIGN.md 3.1)"},
            "host_cpu": {"cpus_granted": cpu_quota(), "replay_threads": threads,
                         "cpu_ms_per_step_this_rank": round(((cpu1.user + cpu1.system) - (cpu0.user + cpu0.system)) * 1e3 / args.steps, 1),
                         "cgroup_cpu_ms_per_step": round((cg1["usage_usec"] - cg0["usage_usec"]) / 1e3 / args.steps, 1) if "usage_usec" in cg0 and "usage_usec" in cg1 else None,
                         "throttled_periods": (cg1["nr_throttled"] - cg0["nr_throttled"]) if "nr_throttled" in cg0 and "nr_throttled" in cg1 else None,
                         "throttled_ms_per_step": round((cg1["throttled_usec"] - cg0["throttled_usec"]) / 1e3 / args.steps, 2) if "throttled_usec" in cg0 and "throttled_usec" in cg1 else None,
                         "note": "CPU time of this process (replay threads into the decoders, the GPU legs' threads -- their waits spin --, Python) over the timed "
                                 "region, per step; cgroup = everything in the container (all ranks at N > 1); throttled = time the container's threads "
                                 "were held back by its CFS quota.  cpu_ms_per_step / cpus_granted is the floor the host sets under ms_per_step: what decides "
                                 "the N > 1 curve on a host whose ranks share one quota (DESIGN 4)"},
            "breakdown_ms": {"k_wave_timed_region": round(live_det_ms, 3), "k_wave_alone": round(solo_det_ms, 3), "gpu_leg_alone_incl_d2h": round(solo_tot_ms, 3),
                             "gpu_leg_alone_parts": {**solo_parts, "what": "detect = the detection pass (k_wave: producers, consumers, run-again), dir = package directory, "
                                                                            "count = the slicers' sizing pass (k_slice into staging slots, the pre-filter's verdicts applied), scan = offsets, "
                                                                            "write = placing pass + slice index, d2h = records to pinned host memory"},
                             "gpu_leg_overlapped": round(float(np.mean(tot_ms)), 3),
                             "host_dispatch": round(float(np.mean(disp_s)) * 1e3, 3), "host_replay_call": round(replay_ms, 3), "engines": pipe.n_eng},
            "kernel_only": {"value": round(n_streams * n_samples / (live_det_ms * 1e-3) / 1e6, 1), "unit": "Msamples/s"},
            "packages_per_step": int(tot_pk), "decoded_messages_per_step": int(tot_msg), "bitbuffers_to_host_per_step": int(tot_bits),
            "d2h_bytes_per_step_per_gpu": int(records["sent"][1]),
            "decoders_behind_the_path": f"the reference's real decode_fn ({plug.source}), ordered multi-threaded replay, "
                                        f"device-side pre-filter on, {sum(1 for x in stateless if x == 1)} of {len(stateless)} decoders declared stateless by the host, "
                                        f"{sum(1 for x in stateless if x == 2)} with their state behind decode_ctx (asked with it out of reach), the others never asked",
            "decoded_events_gathered": gathered_json,
            "gathered": [{k: v for k, v in p.items() if k != "pk"} for p in per],
        }
    pipe_for_extra = pipe
    want_dropin = False

    # ---- secondary measurements of the same run (rank 0, N = 1, the default workload only) ----
    if not strong and not BK.emu and not (args.quick and args.resident):  # (--quick --resident: the profile run -- under rocprofv3 the H2D copies are blit kernels beside k_wave)
        # the same pipeline fed the other way (every rank runs it: the ranks share the host): from pinned host memory with the
        # H2D copy inside every step when `value` is the resident rate, resident when --from-host made the copy part of `value`
        k_res = max(3, min(args.steps, 20))
        other = leg_args_resident if args.from_host else leg_args_pcie
        pipe.run(2, other)
        el, (det_res, _, _, _) = timed(dist, torch, lambda: pipe.run(k_res, other))
        if rank == 0:
            if not args.from_host:
                result["value_r04_definition"] = round(world * n_streams * n_samples * k_res / el / 1e6, 2)
            result["hbm_resident" if args.from_host else "pcie_inclusive"] = {
                "value": round(world * n_streams * n_samples * k_res / el / 1e6, 2), "unit": "Msamples/s", "steps": k_res,
                "ms_per_step": round(el / k_res * 1e3, 3), "k_wave_ms": round(float(np.mean(det_res)), 3),
                "note": ("the same pipeline without the H2D copy (inputs resident in HBM when the timed region starts)" if args.from_host else
                         "the same pipeline with every step's input copied from pinned host memory first (1 GiB per step over PCIe: the link's "
                         "18.7 ms per GiB is the floor of this line, profiles/r04_probes.txt)")}
    if not strong and not args.quick and not args.no_cpu_baseline:
        # Parity of the timed region's own output, on every rank: the JSON lines this rank's decoders produced in the LAST step
        # of the timed region (what it sent into the gather) against the unmodified reference (oracle/_ref, its real decoders,
        # one thread) over the very captures that step read.  The same run is the CPU baseline (rank 0's is reported).
        import hashlib
        last = host_batches[(args.steps * per_step - 1) % len(host_batches)]
        real, cpu_text = cpu_baseline_real_decoders(last)
        mine = records["last_text"]
        ok = bool(cpu_text is not None and cpu_text == mine)
        ok_all = ok
        if dist:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok_all = bool(int(t.item()))
        if rank == 0:
            js = {"captures": int(last.shape[0]), "gpu_sha256": hashlib.sha256(mine).hexdigest(), "gpu_lines": mine.count(b"\n"),
                  "cpu_sha256": hashlib.sha256(cpu_text).hexdigest() if cpu_text is not None else None,
                  "cpu_lines": cpu_text.count(b"\n") if cpu_text is not None else None, "equal": ok, "equal_on_every_rank": ok_all,
                  "note": "gpu = the JSON lines rank 0 put into the gather at the end of the timed region (its last step); cpu = the unmodified "
                          "reference over the same captures; every rank makes the same comparison for its own batches"}
            result["cpu_baseline"] = real
            result["parity"] = ("decoded-json-sha256-match (the timed region's last step, whole)" if ok_all else
                                "decoded json not compared (oracle/_ref did not travel)" if cpu_text is None else "DECODED JSON MISMATCH")
            result["parity_detail"] = {"timed_region_last_step": js}
    if rank == 0 and world == 1 and not strong and not args.quick:
        # ... and every bitbuffer of one batch: a checksum decode_fn on both sides (pre-filter off: the checksum wants every record)
        e0 = pipe.engines[0]
        pipe.on_host_leg = None
        plug.take()
        e0.set_prefilter(0)
        ctx.sum = 0
        ctx.events = 0
        pipe.gpu_leg(0, src=batches[0][:n_batch])
        e0.dispatch(rdev_arr, n_threads=threads)
        e0.set_prefilter(1)
        gpu_digest, gpu_events = int(ctx.sum), int(ctx.events)
        if not args.no_cpu_baseline:
            try:
                one, many, parity = cpu_baseline_config2(host_batches[0][:n_batch], devs, gpu_digest, gpu_events)
                result["cpu_baseline_checksum_decode_fn"] = one
                result["cpu_baseline_nproc"] = many
                if result.get("cpu_baseline") is None:
                    result["cpu_baseline"] = one
                result["parity"] = result.get("parity", "") + "; bitbuffers of batch 0: " + parity
                result.setdefault("parity_detail", {})["bitbuffers"] = {"verdict": parity, "sample": f"batch 0: {n_batch} captures, checksum decode_fn on both sides"}
            except Exception as e:  # the checker must not take the measurement down with it
                result["parity"] = result.get("parity", "") + f"; bitbuffer check failed: {e}"
            try:
                result["real_decoders"] = real_decoders_leg(host_batches[0][:n_batch], batches[0][:n_batch], devs, threads, local_rank)
            except Exception as e:
                result["real_decoders"] = dict(error=str(e))
            want_dropin = True  # (the legs ran first -- dropin_first --; their JSON is compared at the end)
            try:  # the other single-stream workloads of BASELINE.json under the same roof (short passes, inputs resident)
                result["other_configs"] = other_configs_summary(args)
            except Exception as e:
                result["other_configs"] = dict(error=str(e))
    if rank == 0 and strong and world == 1 and not args.quick and host_first is not None:
        # parity of the first captures of the list, record for record, against the unmodified reference
        try:
            from oracle import pyoracle as po
            from rtl_433_amd.engine import BatchEngine
            if po.have_ref():
                e = BatchEngine(flow_cfg(2, 250000), devs)
                e.run(torch.from_numpy(host_first).cuda())
                pk, _ = e.packages()
                ev, _ = e.events()
                e.close()
                ref = po.Ref(record=True)
                t0 = time.perf_counter()
                for s in range(host_first.shape[0]):
                    ref.run(host_first[s], 2, 250000, 433920000, fpdm=0, stream_index=s)
                dt = time.perf_counter() - t0
                ok = po.strip_ret_pos(pk) == ref.packages()[0] and po.events_normalize(ev) == po.events_normalize(po.canonical_events(ref.events()[0]))
                ref.close()
                result["parity"] = ("records-match" if ok else "MISMATCH") + f" (first {host_first.shape[0]} captures of the list, byte for byte vs the reference)"
                ref = po.Ref(record=False)
                ref.set_digest_mode(2)
                t0 = time.perf_counter()
                for s in range(host_first.shape[0]):
                    ref.run(host_first[s], 2, 250000, 433920000, fpdm=0, stream_index=s)
                dt = time.perf_counter() - t0
                ref.close()
                result["cpu_baseline"] = dict(value=round(host_first.shape[0] * n_samples / dt / 1e6, 2), unit="Msamples/s", cores=1, kind="reference",
                                              sample=f"the first {host_first.shape[0]} captures of the list, checksum decode_fn, one pass, single thread")
                if host_launch0 is not None and records.get("launch0_text") is not None:
                    # the decoded events of the whole first launch: the JSON lines of the reference's real decoders behind the GPU path against the
                    # unmodified reference over the same captures (one core: this is also the CPU baseline with real decoders)
                    import hashlib
                    real, cpu_text = cpu_baseline_real_decoders(host_launch0, what="the first launch of the list")
                    mine = records["launch0_text"]
                    result["cpu_baseline"] = real
                    result["parity"] += ("; decoded-json-sha256-match" if cpu_text == mine else "; DECODED JSON MISMATCH") + f" (the {host_launch0.shape[0]} captures of the first launch)"
                    result["parity_detail"] = {"first_launch": {"captures": int(host_launch0.shape[0]), "gpu_sha256": hashlib.sha256(mine).hexdigest(),
                                                                "cpu_sha256": hashlib.sha256(cpu_text).hexdigest(), "json_lines": mine.count(b"\n")}}
        except Exception as e:
            result["parity"] = f"check failed: {e}"
    pipe_for_extra.close()
    if want_dropin and dropin_first is not None:
        # (the legs themselves ran first, before this process opened the GPU: dropin_first)
        ph_texts = dropin_first.pop("_ph_texts", None)
        last_text = records.get("last_text")
        if ph_texts is not None and isinstance(dropin_first.get("pipeline_host_hip"), dict) and last_text is not None:
            dropin_first["pipeline_host_hip"]["json_equals_timed_region"] = bool(ph_texts == {last_text})
        dropin_first["when"] = "before this process opened the GPU (a process of its own beside an idle device, as a user runs it)"
        result["dropin"] = dropin_first
    return result


# ---------------------------------------------------------------- config 3 / config 5: one long stream

def fsk_stream_config3(n_samples, seed=3):
    """constant-envelope 2-FSK +-40 kHz Manchester bursts, half bit 50 us, 64-256 bits, every 20 ms,
    AWGN sigma 0.01 FS, 1024 kS/s cs16.  Built piecewise (the phase track of 64 Mi samples in one go needs GBs)."""
    rng = np.random.default_rng(seed)
    rate, dev_hz, hb = 1024000, 40e3, 51
    out = np.empty(2 * n_samples, dtype=np.int16)
    gap = 20480  # 20 ms of silence between two bursts
    piece = 1 << 22
    for p0 in range(0, n_samples, piece):
        n = min(piece, n_samples - p0)
        level = np.zeros(n)
        keyed = np.zeros(n, dtype=bool)
        b0 = 6000
        while True:
            nbits = int(rng.integers(64, 257))
            bits = np.concatenate([np.zeros(16, dtype=np.uint8), rng.integers(0, 2, nbits).astype(np.uint8)])
            halves = np.empty(2 * bits.size)
            halves[0::2] = np.where(bits == 1, 1.0, -1.0)
            halves[1::2] = -halves[0::2]
            track = np.repeat(halves, hb)
            if b0 + track.size + gap // 2 >= n:
                break
            level[b0:b0 + track.size] = track
            keyed[b0:b0 + track.size] = True
            b0 += track.size + gap
        ph = np.cumsum(2.0 * np.pi * dev_hz * level / rate)
        a = 0.8 * 32767.0 * keyed
        i = a * np.cos(ph) + rng.normal(0.0, 0.01 * 32767.0, n)
        q = a * np.sin(ph) + rng.normal(0.0, 0.01 * 32767.0, n)
        out[2 * p0: 2 * (p0 + n): 2] = np.clip(np.rint(i), -32768, 32767).astype(np.int16)
        out[2 * p0 + 1: 2 * (p0 + n): 2] = np.clip(np.rint(q), -32768, 32767).astype(np.int16)
    return out


def mixed_stream_config5(n_samples, seed=5):
    """2 MS/s cu8, OOK bursts (config-2 recipe, rates scaled) and FSK PCM bursts over a noise floor
    that steps between sigma 1 and sigma 3 LSB."""
    from rtl_433_amd import synth
    rng = np.random.default_rng(seed)
    rate = 2000000
    out = np.empty(2 * n_samples, dtype=np.uint8)
    piece = 1 << 22
    for k, p0 in enumerate(range(0, n_samples, piece)):
        n = min(piece, n_samples - p0)
        sigma = 1.0 if (k // 4) % 2 == 0 else 3.0  # the floor steps every 4 pieces (8.4 s of signal)
        i = 128.0 + rng.normal(0.0, sigma, n)
        q = 128.0 + rng.normal(0.0, sigma, n)
        pos = 20000
        while pos < n - 400000:
            if rng.random() < 0.6:  # OOK burst
                nbits = int(rng.integers(24, 65))
                bits = rng.integers(0, 2, nbits).astype(np.uint8)
                short = int(rng.integers(200, 501) * rate / 1e6)
                fam = ("pwm", "ppm", "mc")[int(rng.integers(0, 3))]
                segs = synth.ook_segments(bits, fam, short, 2 * short, repeats=int(rng.integers(1, 3)), repeat_gap=6 * short)
                length = sum(s[0] for s in segs)
                if pos + length >= n:
                    break
                mask = synth._segments_to_mask(segs, length)
                amp = float(rng.uniform(40.0, 110.0))
                tone = float(rng.uniform(-300e3, 300e3))
                t = np.arange(length)
                ph = 2.0 * np.pi * tone / rate * t
                i[pos:pos + length] += amp * mask * np.cos(ph)
                q[pos:pos + length] += amp * mask * np.sin(ph)
                pos += length + int(rng.integers(60000, 200000))
            else:  # FSK PCM burst, 100 us per bit
                nbits = int(rng.integers(64, 200))
                hb = int(100e-6 * rate)
                halves = np.concatenate([np.tile(np.array([1.0, -1.0]), 12), np.where(rng.integers(0, 2, nbits) == 1, 1.0, -1.0)])
                track = np.repeat(halves, hb)
                length = track.size
                if pos + length >= n:
                    break
                ph = np.cumsum(2.0 * np.pi * 60e3 * track / rate)
                amp = float(rng.uniform(50.0, 110.0))
                i[pos:pos + length] += amp * np.cos(ph)
                q[pos:pos + length] += amp * np.sin(ph)
                pos += length + int(rng.integers(60000, 200000))
        out[2 * p0: 2 * (p0 + n): 2] = np.clip(np.rint(i), 0, 255).astype(np.uint8)
        out[2 * p0 + 1: 2 * (p0 + n): 2] = np.clip(np.rint(q), 0, 255).astype(np.uint8)
    return out


def run_stream(args, ctxd, parity_prefix=False):
    """configs 3 and 5: one long stream per GPU (a single stream does not shard across GPUs: replicas only).
    parity_prefix: the short form the default run appends (other_configs): the reference runs over the whole (shortened)
    stream, whatever --quick / --no-cpu-baseline say."""
    import torch

    from oracle import pyoracle as po
    from rtl_433_amd import _lib
    from rtl_433_amd.engine import BatchEngine, digest_plugin_addr, flow_cfg, load_device_table, make_rdevices
    rank, world, local_rank, dist = ctxd["rank"], ctxd["world"], ctxd["local_rank"], ctxd["dist"]
    devs, protocols, names = load_device_table()
    threads = args.threads or replay_threads(world)
    if args.config == 3:
        n_samples = args.stream_samples or (64 << 20)
        ss, rate, freq = 4, 1024000, 868000000
        host = fsk_stream_config3(n_samples)
        flex = np.zeros(1, dtype=devs.dtype)
        flex[0] = (18, 50.0, 50.0, 120.0, 0.0, 0.0, 0.0, 0)  # -X n=mc,m=FSK_MC_ZEROBIT,s=50,l=50,r=120
        devs = np.concatenate([devs, flex])
        names, protocols = names + ["mc"], protocols + [0]
        cfg = flow_cfg(4, rate, fpdm=1, center_frequency=freq)
        ref_kw = dict(flex=["n=mc,m=FSK_MC_ZEROBIT,s=50,l=50,r=120"])
        ref_levels = None
        workload = (f"configs[2]: one 1024 kS/s cs16 FSK stream of {n_samples} samples ({n_samples * 4 >> 20} MiB), 2-FSK +-40 kHz Manchester bursts every "
                    "20 ms, min/max FSK detector (868 MHz), all default decoders + flex FSK_MC_ZEROBIT s=50 l=50 r=120")
    else:
        n_samples = args.stream_samples or (256 << 20)
        ss, rate, freq = 2, 2000000, 433920000
        host = mixed_stream_config5(n_samples)
        cfg = flow_cfg(2, rate, fpdm=0, center_frequency=freq, auto_level=1.0, fm_low_pass=0.15)
        ref_kw = {}
        ref_levels = dict(auto_level=1.0, fm_low_pass=0.15)
        workload = (f"configs[4]: one 2 MS/s cu8 stream of {n_samples} samples, OOK + FSK bursts over a stepping noise floor, -Y autolevel, "
                    "-Y filter=0.15, all default decoders")
    d = torch.from_numpy(host.view(np.uint8)).cuda().reshape(1, -1)
    ctx = _lib.DigestCtx(0, 0)
    rdev_arr, rdev_objs = make_rdevices(devs, digest_plugin_addr(), C.addressof(ctx), names, protocols)
    # The decoders behind the path are the reference's REAL ones (the plugin library: its default set + this config's flex
    # decoder), replayed in reference order on the host's threads with the device-side pre-filter on -- as in the headline.
    # What they say (JSON lines of the reference's own printer) is what is compared with the unmodified reference over the
    # whole stream; every bitbuffer of the stream is checked by a checksum decode_fn on both sides in an untimed pass after it.
    plug = real_decoder_plugins(flex=ref_kw.get("flex"))
    assert len(plug.devices) == len(devs), "the plugin library registers another decoder set than the device table"
    eng = BatchEngine(cfg, devs, profiling=True)
    if DEBUG_FLAGS:
        eng.set_debug(DEBUG_FLAGS)
    eng.set_stateless(plug.stateless())
    eng.probe_prefilter(plug.devices, helper=plug.helper_probe())
    hooks = plug.hooks()
    lat = []
    said = {}

    def one_pass(record_latency=False, checksum=False):
        ctx.sum = 0
        ctx.events = 0
        t0 = time.perf_counter()
        npk = eng.run(d)
        t_gpu = time.perf_counter() - t0
        if checksum:  # every bitbuffer through a checksum decode_fn (the pre-filter is off for this pass: the checksum wants every record)
            eng.dispatch(rdev_arr, n_threads=threads)
        elif record_latency:
            # the whole stream is on the host at t0; a burst's events leave when its package has been through the decoders
            stamps = []

            @_lib.PACKAGE_FN
            def on_pkg(user, stream, typ, pd):
                stamps.append(time.perf_counter() - t0)
            eng.dispatch(plug.devices, pkg_cb=on_pkg, n_threads=1)
            stamps.append(time.perf_counter() - t0)
            lat.extend(stamps[1:])  # package k is done when package k+1 begins
            plug.take()
        else:
            eng.dispatch_ordered(plug.devices, hooks, threads)
            said["text"], said["messages"] = plug.take()
        return npk, t_gpu

    for _ in range(max(1, args.warmup)):
        one_pass()
    tms = []

    def region():
        out = None
        for _ in range(args.steps):
            out = one_pass()
            tms.append(eng.timing())
        return out
    elapsed, (npk, _) = timed(dist, torch, region)
    gpu_text, gpu_messages = said.get("text", b""), said.get("messages", 0)
    eng.set_prefilter(0)
    one_pass(checksum=True)
    eng.set_prefilter(1)
    gpu_digest, gpu_events = int(ctx.sum), int(ctx.events)
    det_ms = float(np.mean([t["detect_ms"] for t in tms]))
    result = None
    if rank == 0:
        alg_bytes = ss * n_samples
        achieved = alg_bytes / (det_ms / 1e3) / 1e9
        result = {
            "metric": METRIC, "value": round(world * n_samples * args.steps / elapsed / 1e6, 2), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "s16" if ss == 4 else "u8", "data": "synthetic (resident in HBM)",
            "config": {"workload": workload, "samples": n_samples, "sample_rate": rate, "decoders": len(devs),
                       "split": eng.split_stats(), "parallelism": "one stream per GPU (replicas only); inside the GPU the stream is cut "
                                                                  "at verified idle points into one wavefront per segment"},
            "roofline": {"bound": "hbm", "kernel": "detection pass (k_tile_max + k_wave over the segments + stitch)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": pmc_traffic("config3" if args.config == 3 else "config5", alg_bytes, det_ms / 1e3),
                         "note": f"achieved = {ss} B/sample x samples / detection time of one pass (HIP events, mean of the timed steps)"},
            "breakdown_ms": {k: round(float(np.mean([t[k] for t in tms])), 3) for k in tms[0]},
            "packages_per_step": int(npk), "bitbuffers_per_step": gpu_events, "decoded_messages_per_step": int(gpu_messages),
            "decoders_behind_the_path": f"the reference's real decode_fn ({plug.source}{' + flex ' + ref_kw['flex'][0] if ref_kw.get('flex') else ''}), "
                                        f"ordered replay on {threads} threads, device-side pre-filter on",
        }
        if args.config == 5 and (parity_prefix or not args.quick):
            one_pass(record_latency=True)
            a = np.array(lat) * 1e3
            edges = [0, 5, 10, 20, 50, 100, 200, 500, 1000, 2000, 5000, 1e9]
            hist, _ = np.histogram(a, bins=edges)
            result["latency_per_burst_ms"] = {"count": int(a.size), "p50": round(float(np.percentile(a, 50)), 2), "p90": round(float(np.percentile(a, 90)), 2),
                                              "p99": round(float(np.percentile(a, 99)), 2), "max": round(float(a.max()), 2),
                                              "histogram": {f"<{int(e)}" if e < 1e9 else "more": int(h) for e, h in zip(edges[1:], hist)},
                                              "note": "time from the stream being on the host (start of the pass) to the moment a burst's package has been through "
                                                      "every decoder; single-threaded replay in package order"}
        if (parity_prefix or (not args.no_cpu_baseline and not args.quick)) and po.have_ref():
            try:
                import hashlib
                sample = min(n_samples, 1 << 28)
                part = host[: sample * (2 if ss == 2 else 2)].view(np.uint8)[: sample * ss]
                # the unmodified reference with its real decoders over the stream: what they say, as JSON lines
                ref = po.Ref(call_real=True, record=False, **ref_kw)
                if ref_levels:
                    ref.set_levels(**ref_levels)
                ref.set_digest_mode(0)
                ref.text_mode(True)
                ref.take_text()
                t0 = time.perf_counter()
                ref.run(part, ss, rate, freq, fpdm=2, stream_index=0)
                dt = time.perf_counter() - t0
                cpu_text = ref.take_text()
                ref.close()
                result["cpu_baseline"] = dict(value=round(sample / dt / 1e6, 2), unit="Msamples/s", cores=1, kind="reference",
                                              sample=f"the first {sample} samples of the stream, the reference's real decoders, one pass, single thread")
                # ... and with a checksum decode_fn: every bitbuffer of the stream
                ref = po.Ref(record=False, **ref_kw)
                if ref_levels:
                    ref.set_levels(**ref_levels)
                ref.set_digest_mode(2)
                ref.run(part, ss, rate, freq, fpdm=2, stream_index=0)
                dg, nev = ref.digest2(), ref.digest()[1]
                ref.close()
                if sample == n_samples:
                    json_ok = cpu_text == gpu_text
                    dig_ok = dg == gpu_digest and nev == gpu_events
                    result["parity"] = (("decoded-json-sha256-match" if json_ok else "DECODED JSON MISMATCH") + "; bitbuffers: "
                                        + ("digest-match" if dig_ok else f"MISMATCH cpu {dg}/{nev} gpu {gpu_digest}/{gpu_events}"))
                    result["parity_detail"] = {"gpu_sha256": hashlib.sha256(gpu_text).hexdigest(), "cpu_sha256": hashlib.sha256(cpu_text).hexdigest(),
                                               "json_lines": gpu_text.count(b"\n"), "bitbuffers": int(nev),
                                               "what": "JSON lines of the reference's real decoders behind the GPU path against the unmodified reference over the "
                                                       "whole stream; every bitbuffer of the stream by a checksum decode_fn on both sides"}
                else:
                    result["parity"] = "not compared (cpu sample is a prefix of the stream)"
            except Exception as e:
                result["parity"] = f"cpu baseline failed: {e}"
    eng.close()
    plug.close()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (one process each).  Under torchrun it must equal WORLD_SIZE; "
                    "started plainly with N > 1 the script re-executes itself under torch.distributed.run with N ranks")
    ap.add_argument("--backend", default="hip", choices=["hip", "emu"], help="emu: TEST ONLY -- the CPU wave emulator of the test suite + gloo "
                    "(tests/test_bench_launch.py drives the launcher and the gather with it); measures nothing")
    ap.add_argument("--capture-samples", type=int, default=65536, help="configs 2 / 4: samples per capture (the emulator test shortens it; "
                    "any other value than 65536 is not BASELINE's workload and the line says so)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--streams", type=int, default=1024, help="config 2: captures per batch")
    ap.add_argument("--batches", type=int, default=8, help="config 2: batches a step submits together per GPU")
    ap.add_argument("--list-len", type=int, default=65536, help="config 4: captures in the one list")
    ap.add_argument("--launch", type=int, default=8192, help="config 4: captures per launch")
    ap.add_argument("--list-distinct", type=int, default=0, help="config 4: capture i of the list is seed i mod D (0: every capture distinct); the default run's "
                    "other_configs uses 16384 (making 65536 captures on the host takes over a minute)")
    ap.add_argument("--stream-samples", type=int, default=0, help="configs 3 / 5: samples of the stream (0 = the recipe's)")
    ap.add_argument("--threads", type=int, default=0, help="host dispatch threads per rank (0 = auto)")
    ap.add_argument("--engines", type=int, default=3, help="batch engines in the software pipeline (>= 2)")
    ap.add_argument("--exclusive", type=int, default=2, choices=[0, 1, 2, 3], help="pipeline: engines take turns on the detection kernel (1), on detection + slicers (2, the default: "
                    "record copies run beside the next engine's kernels), on the whole GPU leg incl. the record copies (3: the profile run; within 5 %% of 2 either way, "
                    "profiles/r05_*), not at all (0)")
    ap.add_argument("--debug", type=lambda x: int(x, 0), default=0, help="development: R433_DEBUG_* flags for the engines (A/B timing)")
    ap.add_argument("--from-host", action="store_true", help="config 2: the timed region starts from pinned host memory (every step's H2D copy inside it: what "
                    "`pcie_inclusive` reports otherwise)")
    ap.add_argument("--resident", action="store_true", help="(the default since round 5) config 2: inputs resident in HBM for the timed region; "
                    "the profile run uses it: under rocprofv3 the H2D copies of the default run become blit kernels that share the CUs with k_wave")
    ap.add_argument("--h2d-wait", default="spin", choices=["nap", "spin"], help="config 2: how a GPU leg's thread waits for its input copy (A/B)")
    ap.add_argument("--quick", action="store_true", help="the headline measurement only (no PCIe / CPU / real-decoder legs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    global DEBUG_FLAGS, EXCLUSIVE, H2D_NAP
    H2D_NAP = args.h2d_wait == "nap"
    DEBUG_FLAGS = args.debug
    EXCLUSIVE = args.exclusive
    if args.steps is None:
        args.steps = {2: 40, 4: 5, 3: 10, 5: 5}[args.config]
    if args.warmup is None:
        args.warmup = {2: 3, 4: 1, 3: 2, 5: 1}[args.config]

    global BK, CAPTURE_SAMPLES
    CAPTURE_SAMPLES = args.capture_samples
    # ---- how many ranks: --gpus N is a promise about the process group, kept or the run stops
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)  # N ranks of this very command line; rank 0 prints the line
    # stdout carries ONE line, the result: whatever libraries print on the way (RCCL's version banner goes to stdout) is sent
    # to stderr, the line is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to print a line that "
                         "names another number of GPUs than the run used")

    import torch

    emu = args.backend == "emu"
    if not emu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (rtl_433_amd has no CPU path)")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank}, the node shows {torch.cuda.device_count()} GPU(s)")
        torch.cuda.set_device(local_rank)
    BK = Backend(args.backend)
    dist = None
    seen_gpus = None
    if env_world is not None:  # started by a launcher: the process group exists for every N, one rank included
        import torch.distributed as dist
        if emu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
        world = dist.get_world_size()  # what the process group really has
        # one distinct GPU per rank: every rank names its device, all ranks compare
        mine = f"emu:{rank}" if emu else f"{os.uname().nodename}/{getattr(torch.cuda.get_device_properties(local_rank), 'uuid', local_rank)}/{local_rank}"
        names = [None] * world
        dist.all_gather_object(names, mine)
        if len(set(names)) != world:
            raise SystemExit(f"bench.py: {world} ranks on {len(set(names))} distinct GPU(s): {names}")
        seen_gpus = names
    ctxd = dict(rank=rank, world=world, local_rank=local_rank, dist=dist)
    # Waiting for the GPU spins on this stack (tools/spin_probe.py), and a rank keeps two GPU legs in flight: ranks that share a
    # small CPU quota -- eight on the 16 CPUs these boxes grant -- would burn it on waiting while their decoders starve.  With
    # fewer than six CPUs to a rank the engines poll with naps instead (R433_DEBUG_NAP_WAIT; on one GPU with CPUs to spare the
    # spinning wait is the faster one: profiles/r04_wait_asleep_ab.txt).
    if world > 1 and cpu_quota() / world < 6:
        DEBUG_FLAGS |= NAP_WAIT
    result = run_batched(args, ctxd) if args.config in (2, 4) else run_stream(args, ctxd)
    if rank == 0 and isinstance(result, dict) and isinstance(result.get("config"), dict):
        result["config"]["gpu_waits"] = "polled with naps (ranks share a small CPU quota)" if DEBUG_FLAGS & NAP_WAIT else "hipEventSynchronize (spins)"
        result["config"]["ranks"] = {"world_size": world, "launcher": "torch.distributed.run" if env_world is not None else "none (one process)",
                                     "collective_backend": (None if dist is None else "gloo" if emu else "nccl (RCCL)"), "devices": seen_gpus}
        if emu:
            result["backend"] = "CPU wave emulator + gloo: a TEST of the launcher and the gather, not a measurement"
        if CAPTURE_SAMPLES != 65536:
            result["config"]["not_baseline_workload"] = f"captures of {CAPTURE_SAMPLES} samples (BASELINE: 65536)"
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
