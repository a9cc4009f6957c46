"""What `bound by wavefront instruction issue` means in numbers (GPU box): SQ counter passes over the detection kernel on
the bench's grid (8192 captures x 65536 cu8 samples, tools/kbench.py --nodevs), whole kernel and producers alone
(since round 5 the two roles are two launches and counted apart), counters only (--kernel-trace --pmc, separate passes).

    python tools/pmc_issue.py  -> gpurun_out/r04_pmc/issue.json   (copy to profiles/r04_pmc_issue.json; bench.py quotes it)

Derived figures and their formulas are in the file.  Clock: the kernel's duration comes from the same pass's kernel trace;
SIMD issue rates are per shader clock at the nominal 2.4 GHz (MI355X_MICROARCH.md), stated as such."""
import json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", os.environ.get("R433_PMC_TAG", "r05_pmc"))
PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"],
          ["SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"],
          ["SQ_WAIT_ANY", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]]
N_CAP, N_SAMP, SIMDS, CLK = 8192, 65536, 1024, 2.4e9
BENCH_BATCH = os.environ.get("R433_PMC_BENCH_BATCH", "1") == "1"  # the captures of bench.py's own batch (every third one a protocol transmission)
WHAT = os.environ.get("R433_PMC_WHAT", "wave")                     # wave: the detection pass (k_wave); slice: the decoder fan-out (k_slice, all decoders)
KERNEL = "k_wave" if WHAT == "wave" else "k_slice"


def form_of(name):
    """k_wave<2, true, true, false, FORM> -> 'form4' (producers), 'form5' (consumers), 'form2' (pairs: the whole capture, or the run-again launch);
    k_slice<MODE, CAP> -> 'slice<MODE,CAP>' (2 = sizing pass into staging slots, 3 = placing pass; CAP 260 = the small packages' launch)"""
    try:
        if "k_slice<" in name:
            return "slice<" + name[name.index("k_slice<") + 8:].split(">")[0].replace(" ", "") + ">"
        return "form" + name[name.index("k_wave<"):].split(">")[0].split(",")[-1].strip()
    except Exception:
        return "other"


KBENCH_LINE = [""]


def one(tag, counters, debug):
    """one --pmc pass -> {form: {counter: mean over that form's dispatches of the sum over instances, '_duration_ns': mean, '_dispatches': n}}"""
    d = os.path.join(OUT, f"issue_{tag}")
    os.makedirs(d, exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--streams", str(N_CAP), "--reps", "2"] + (["--debug", str(debug)] if debug else []) \
        + (["--nodevs"] if WHAT == "wave" else []) + (["--bench-batch"] if BENCH_BATCH else [])
    if not os.environ.get("R433_PMC_OFFLINE"):  # (offline: only read the databases of an earlier visit again)
      r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "r", "--"] + cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=False, timeout=400)
      for ln in r.stdout.decode(errors="replace").splitlines():
          if ln.startswith("flags="):
              KBENCH_LINE[0] = ln
    dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs:
        return {}
    c = sqlite3.connect(dbs[0])
    out = {}
    try:
        rows = c.execute("select name, counter_name, dispatch_id, sum(counter_value) from pmc_events where name like '%" + KERNEL + "%' group by name, counter_name, dispatch_id").fetchall()
    except Exception as e:
        return {"error": str(e)}
    per = {}
    for name, counter, disp, total in rows:
        per.setdefault((form_of(name), counter), []).append(total)
    for (form, counter), v in per.items():
        out.setdefault(form, {})[counter] = sum(v) / len(v)
        out[form]["_dispatches"] = len(v)
    try:  # kernel durations from the trace of the same pass
        t = [r[0] for r in c.execute("select name from sqlite_master where type='table'").fetchall()]
        kd = [n for n in t if "kernel_dispatch" in n]
        sym = [n for n in t if "kernel_symbol" in n]
        if kd and sym:
            for name, dur in c.execute(f"select s.display_name, avg(d.end - d.start) from {kd[0]} d join {sym[0]} s on d.kernel_id = s.id where s.display_name like '%{KERNEL}%' group by s.display_name").fetchall():
                out.setdefault(form_of(name), {})["_duration_ns"] = dur
    except Exception as e:
        out["_duration_error"] = str(e)
    for f in dbs:
        if os.path.getsize(f) > (8 << 20):
            os.remove(f)
    return out


def main():
    if BENCH_BATCH and not os.environ.get("R433_PMC_OFFLINE"):  # the batch is made once, outside the profiler (tools/kbench.py caches it in /tmp)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--bench-batch", "--make-batch-only", "--streams", str(N_CAP)],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=300)
    forms = {}
    for i, counters in enumerate(PASSES):
        for form, got in one(f"{WHAT}_pass{i}", counters, int(os.environ.get("R433_PMC_DEBUG", "0"), 0)).items():
            if isinstance(got, dict):
                forms.setdefault(form, {}).update(got)
    samples = N_CAP * N_SAMP
    batch = "bench.py's own batch (every third capture a protocol transmission; 1024 distinct captures tiled)" if BENCH_BATCH else "the synthetic grid (synth.ook_batch)"
    def g(form, k):
        return forms.get(form, {}).get(k) or 0.0

    def derive(form):
        valu, salu, lds, dur = g(form, "SQ_INSTS_VALU"), g(form, "SQ_INSTS_SALU"), g(form, "SQ_INSTS_LDS"), g(form, "_duration_ns")
        ins = valu + salu + lds
        wc, wa = g(form, "SQ_WAVE_CYCLES"), g(form, "SQ_WAIT_ANY")
        return {"wave_instructions": {"valu": valu, "salu": salu, "lds": lds, "all": ins}, "duration_ms": round(dur / 1e6, 3),
                "simd_ipc": round(ins / (dur * 1e-9 * CLK * SIMDS), 3) if dur else None,
                "wait_share": round(wa / wc, 3) if wc else None,
                "waves_per_simd_mean": round(wc * 4 / (dur * 1e-9 * CLK * SIMDS), 2) if dur and wc else None,
                "valu_only_time_ms_at_2_clocks_per_wave64_instruction": round(valu * 2 / SIMDS / CLK * 1e3, 3)}
    formulas = {"simd_ipc": f"wave-instructions / (duration x {CLK / 1e9} GHz x {SIMDS} SIMDs)", "wait_share": "SQ_WAIT_ANY / SQ_WAVE_CYCLES",
                "waves_per_simd_mean": "SQ_WAVE_CYCLES x 4 / (duration x clock x SIMDs) (the counter ticks once per 4 clocks of a resident wavefront: MI355X_MICROARCH.md)"}
    how = ("every counter summed over its instances (XCD x SE ...), mean over the dispatches of a kernel in a pass (2 repetitions); separate --pmc passes; "
           "durations from the kernel trace of the same passes")
    if WHAT == "slice":
        out = {"workload": f"the decoder fan-out of one pass over {N_CAP} captures x {N_SAMP} cu8 samples, {batch}, all default decoders, no pre-filter (tools/kbench.py)",
               "kbench": KBENCH_LINE[0], "raw_per_kernel": forms, "how": how}
        try:
            kv = dict(x.split("=", 1) for x in KBENCH_LINE[0].split() if "=" in x)
            pulses, pkgs = int(kv.get("pulses", 0)), int(kv.get("pkgs", 0))
            derived = {k: derive(k) for k in forms if k.startswith("slice<")}
            sizing = [k for k in derived if k.startswith("slice<2,")]
            ins = sum(derived[k]["wave_instructions"]["all"] for k in sizing)
            ms = sum(derived[k]["duration_ms"] for k in derived)
            # the sizing launches run side by side on two streams: their IPC / residency are quoted for the launch that lasts longest
            main_k = max(sizing, key=lambda k: derived[k]["duration_ms"]) if sizing else None
            derived["summary"] = {"kernels_ms_under_counters": round(ms, 3), "packages": pkgs, "pulses": pulses,
                                  "sizing_pass_wave_instructions": ins,
                                  "wave_instr_per_pulse": round(ins / pulses, 1) if pulses else None,
                                  "wave_instr_per_pulse_note": "wave-instructions of the sizing pass (every decoder's slicer over every package, 64 decoders to a wavefront) / pulses of the "
                                                               "packages: what ONE pulse costs across all 335 decoders (the reference spends ~20 instructions per pulse and decoder)",
                                  "simd_ipc": derived[main_k]["simd_ipc"] if main_k else None, "waves_per_simd": derived[main_k]["waves_per_simd_mean"] if main_k else None,
                                  "of": main_k}
            derived["formulas"] = formulas
            out["derived"] = derived
        except Exception as e:
            out["derived_error"] = str(e)
        os.makedirs(OUT, exist_ok=True)
        json.dump(out, open(os.path.join(OUT, "slice.json"), "w"), indent=1)
        print(json.dumps(out.get("derived", out), indent=1))
        return
    out = {"workload": f"one detection pass over a grid of {N_CAP} captures x {N_SAMP} cu8 samples, {batch} (tools/kbench.py --nodevs), lazy tiles on: the "
                       "producers (FORM 4: filters, tile records to HBM) and the consumers (FORM 5: the detector) are two launches, a third (FORM 2) runs "
                       "again what could not be carried",
           "raw_per_form": forms, "how": how}
    try:
        roles = {"producers": "form4", "consumers": "form5", "run_again_or_pairs": "form2"}
        derived = {}
        tot_i = tot_d = 0.0
        for role, form in roles.items():
            if form not in forms:
                continue
            derived[role] = derive(form)
            tot_i += derived[role]["wave_instructions"]["all"]
            tot_d += g(form, "_duration_ns")
        derived["per_pass"] = {"wave_instructions": tot_i, "wave_instructions_per_iq_sample": round(tot_i / samples, 3), "kernels_ms_under_counters": round(tot_d / 1e6, 3)}
        derived["formulas"] = formulas
        out["derived"] = derived
    except Exception as e:
        out["derived_error"] = str(e)
    os.makedirs(OUT, exist_ok=True)
    json.dump(out, open(os.path.join(OUT, "issue.json"), "w"), indent=1)
    print(json.dumps(out.get("derived", out), indent=1))


if __name__ == "__main__":
    main()
