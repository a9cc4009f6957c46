"""What `bound by wavefront instruction issue` means in numbers (GPU box): SQ counter passes over the detection kernel on
the bench's grid (8192 captures x 65536 cu8 samples, tools/kbench.py --nodevs), whole kernel and producers alone
(R433_DEBUG_SKIP_DETECT: the consumers return at once), counters only (--kernel-trace --pmc, separate passes).

    python tools/pmc_issue.py  -> gpurun_out/r04_pmc/issue.json   (copy to profiles/r04_pmc_issue.json; bench.py quotes it)

Derived figures and their formulas are in the file.  Clock: the kernel's duration comes from the same pass's kernel trace;
SIMD issue rates are per shader clock at the nominal 2.4 GHz (MI355X_MICROARCH.md), stated as such."""
import json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", os.environ.get("R433_PMC_TAG", "r04_pmc"))
PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"],
          ["SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"],
          ["SQ_WAIT_ANY", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]]
N_CAP, N_SAMP, SIMDS, CLK = 8192, 65536, 1024, 2.4e9


def one(tag, counters, debug):
    d = os.path.join(OUT, f"issue_{tag}")
    os.makedirs(d, exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--nodevs", "--streams", str(N_CAP), "--reps", "2"] + (["--debug", str(debug)] if debug else [])
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "r", "--"] + cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=900)
    dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs:
        return {}
    c = sqlite3.connect(dbs[0])
    out = {}
    try:
        rows = c.execute("select counter_name, dispatch_id, sum(counter_value), count(*) from pmc_events where name like '%k_wave%' group by counter_name, dispatch_id").fetchall()
    except Exception as e:
        return {"error": str(e)}
    for name, disp, total, inst in rows:
        out.setdefault(name, []).append((total, inst))
    res = {k: {"sum_over_instances_per_dispatch": sum(t for t, _ in v) / len(v), "instances": v[0][1], "dispatches": len(v)} for k, v in out.items()}
    try:  # kernel duration from the trace of the same pass
        t = [r for r in c.execute("select name from sqlite_master where type='table'").fetchall()]
        kd = [n[0] for n in t if "kernel_dispatch" in n[0]]
        sym = [n[0] for n in t if "kernel_symbol" in n[0]]
        if kd and sym:
            q = f"select avg(d.end - d.start) from {kd[0]} d join {sym[0]} s on d.kernel_id = s.id where s.kernel_name like '%k_wave%'"
            res["_duration_ns"] = c.execute(q).fetchone()[0]
    except Exception as e:
        res["_duration_error"] = str(e)
    for f in dbs:
        if os.path.getsize(f) > (8 << 20):
            os.remove(f)
    return res


def main():
    allr = {}
    for role, debug in (("whole_kernel", 0), ("producers_alone", 256)):
        got = {}
        for i, counters in enumerate(PASSES):
            r = one(f"{role}_{i}", counters, debug)
            got.update(r)
        allr[role] = got
    w = allr["whole_kernel"]

    def s(d, k):
        return d.get(k, {}).get("sum_over_instances_per_dispatch")
    samples = N_CAP * N_SAMP
    out = {"workload": f"k_wave<2,true,true,false,2>, one grid of {N_CAP} captures x {N_SAMP} cu8 samples (tools/kbench.py --nodevs), lazy tiles on",
           "raw": allr,
           "how": "every counter summed over its instances (XCD x SE ...), mean over the two dispatches of a pass; separate --pmc passes; "
                  "producers_alone = the same launch with R433_DEBUG_SKIP_DETECT (consumer wavefronts return from every tile at once)"}
    try:
        valu, salu, lds = s(w, "SQ_INSTS_VALU"), s(w, "SQ_INSTS_SALU"), s(w, "SQ_INSTS_LDS")
        p = allr["producers_alone"]
        pv, ps = s(p, "SQ_INSTS_VALU"), s(p, "SQ_INSTS_SALU")
        dur = w.get("_duration_ns")
        tot = valu + salu + (lds or 0)
        out["derived"] = {
            "wave_instructions_per_launch": {"valu": valu, "salu": salu, "lds": lds, "all": tot},
            "wave_instructions_per_iq_sample": round(tot / samples, 3),
            "of_which_producers": {"valu": pv, "salu": ps, "share_of_valu_plus_salu": round((pv + ps) / (valu + salu), 3)} if pv and ps else None,
            "kernel_duration_ms_under_counters": round(dur / 1e6, 3) if dur else None,
            "simd_ipc": round(tot / (dur * 1e-9 * CLK * SIMDS), 3) if dur else None,
            "simd_ipc_formula": f"all wave-instructions / (duration x {CLK / 1e9} GHz x {SIMDS} SIMDs)",
            "valu_only_time_ms_at_2_clocks_per_wave64_instruction": round(valu * 2 / SIMDS / CLK * 1e3, 3),
            "note": "a CDNA4 SIMD issues a wave64 VALU instruction over 2 clocks (MI355X_MICROARCH.md): at that rate the VALU work alone would take the "
                    "time above; the kernel takes several times that because a wavefront issues one instruction per 6-8 clocks (dependent scalar / "
                    "cross-lane chains) and at most three are resident per SIMD, one or two of them waiting at a tile hand-off"}
    except Exception as e:
        out["derived_error"] = str(e)
    os.makedirs(OUT, exist_ok=True)
    json.dump(out, open(os.path.join(OUT, "issue.json"), "w"), indent=1)
    print(json.dumps(out.get("derived", out), indent=1))


if __name__ == "__main__":
    main()
