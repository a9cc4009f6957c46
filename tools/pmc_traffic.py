"""HBM traffic of the dominant kernel per launch from rocprofv3 PMC passes, the way /opt/skills/guides/MI355X_MICROARCH.md
(HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one), counters only
(--kernel-trace, no other tracing domain), FETCH_SIZE doubled on gfx950 (128-byte requests of wide coalesced streaming
reads are tallied at 64 bytes), WRITE_SIZE as it is (uncalibrated).  Run on the GPU box:

    python tools/pmc_traffic.py [keys...]  -> gpurun_out/r04_pmc/traffic.json   (copy to profiles/r04_pmc_traffic.json)
"""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", os.environ.get("R433_PMC_TAG", "r05_pmc"))

WORKLOADS = {
    # key: (command, kernel name pattern, algorithmic bytes per launch, what a launch is)
    "config2": ([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--reps", "2"], "k_wave<2", 2 * 1024 * 65536,
                "k_wave<2,true,true>, 1024 captures x 65536 cu8 samples (tools/kbench.py, all decoders), one launch"),
    "config3": ([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--quick", "--steps", "2", "--warmup", "1"], ("k_wave<4", "k_tile_max", "k_frame_sums"), 4 * (64 << 20),
                "every kernel that reads the stream in one pass of bench.py --config 3 (one 64 Mi-sample cs16 stream): the cut-planning estimate k_tile_max and k_wave<4,...> over the verified segments, all launches"),
    "config4": ([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--reps", "2", "--nodevs", "--streams", "8192", "--bench-batch"], ("k_wave<2", "k_capture_weight"), 2 * 8192 * 65536,
                "the detection pass (k_wave<2,true,true,..>: producers, consumers, run-again) and the look at the captures that orders its grid (k_capture_weight), one pass over 8192 captures x 65536 cu8 samples of bench.py's own batch (what bench.py --config 4 launches eight times per step, and the default bench once per step)"),
    "config5": ([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--quick", "--steps", "2", "--warmup", "1"], ("k_wave<2", "k_tile_max", "k_frame_sums"), 2 * (256 << 20),
                "every kernel that reads the stream in one pass of bench.py --config 5 (one 256 Mi-sample 2 MS/s cu8 stream, -Y autolevel): k_frame_sums (the levels of every frame have to be known before detection), k_tile_max, k_wave<2,...> over the segments"),
}


def one_pass(tag, counter, cmd):
    d = os.path.join(OUT, f"{tag}_{counter}")
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "r", "--"] + cmd, cwd="/tmp", env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=400)
    dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs:
        return None
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name = ? group by name, dispatch_id", (counter,)).fetchall()
    for f in dbs:
        if os.path.getsize(f) > (8 << 20):
            os.remove(f)
    return rows


def main():
    res = {}
    want = sys.argv[1:] or list(WORKLOADS)
    for key, (cmd, pat, alg, what) in WORKLOADS.items():
        if key not in want:
            continue
        got = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = one_pass(key, counter, cmd)
            if rows is None:
                got = None
                break
            pats = pat if isinstance(pat, tuple) else (pat,)
            mine = [(n, d, v) for n, d, v in rows if any(q in n for q in pats)]
            got[counter] = mine
        if not got:
            continue
        n_disp = len(got["FETCH_SIZE"])
        if key in ("config2", "config4"):
            # every detection pass is one launch of the workload: since round 5 a large grid goes out as three k_wave launches
            # (FORM 4 producers, FORM 5 consumers, FORM 2 run-again) -- the consumers' launches count the passes
            cons = sum(1 for n, _, _ in got["FETCH_SIZE"] if "k_wave" in n and n.split(">")[0].rstrip().endswith(", 5"))
            per = cons or sum(1 for n, _, _ in got["FETCH_SIZE"] if "k_wave" in n)
        else:
            per = 4  # bench.py --config 3 / 5 --steps 2 --warmup 1: three passes over the stream + the untimed checksum pass behind them (round 6)
        fetch_kb = sum(v for _, _, v in got["FETCH_SIZE"]) / per
        write_kb = sum(v for _, _, v in got["WRITE_SIZE"]) / per
        res[key] = dict(kernel=what, dispatches_seen=n_disp, launches_or_passes=per,
                        FETCH_SIZE_kb_raw=round(fetch_kb, 1), WRITE_SIZE_kb_raw=round(write_kb, 1),
                        correction="gfx950 FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane coalesced streaming reads as 64 bytes "
                                   "(MI355X_MICROARCH.md, HBM section): x2; WRITE_SIZE uncalibrated, taken as is; separate --pmc passes",
                        hbm_bytes_per_launch=int(fetch_kb * 1024 * 2 + write_kb * 1024), algorithmic_bytes_per_launch=alg)
        print(key, res[key])
    os.makedirs(OUT, exist_ok=True)
    json.dump(res, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
