"""Where the detection time of the single-stream workloads goes (configs[2] / configs[4]): ONE table per config from a rocprofv3
kernel trace of `bench.py --config N --quick` -- per pass over the stream every kernel of the detection part with its launches,
grid sizes and busy time, and the time the device idles between them (the host's stitch: states back, cuts verified, the next
round planned).  Measure before building.

    python tools/stream_phases.py 3 5      -> stdout (tools/gpu_r6_d.sh keeps it as profiles/r06_stream_phases.txt)
"""
import os, re, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "r06_stream_phases")


def short(name):
    n = name.replace("void ", "").replace("r433::(anonymous namespace)::", "")
    m = re.match(r"(k_\w+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


def one(cfg):
    d = os.path.join(OUT, f"config{cfg}")
    os.makedirs(d, exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(cfg), "--quick", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(["rocprofv3", "--kernel-trace", "-d", d, "-o", "r", "--"] + cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=False, timeout=600)
    line = next((ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")), "")
    dbs = [os.path.join(p, f) for p, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs:
        print(f"config {cfg}: no trace"); return
    c = sqlite3.connect(dbs[0])
    rows = [(short(n), s, e, g) for n, s, e, g in c.execute("select name, start, end, grid_x from kernels order by start")]
    # a pass = from one k_tile_max (the cut planning opens every pass over a split stream) to the kernel before the next
    opens = [i for i, r in enumerate(rows) if r[0].startswith("k_tile_max")]
    opens = [i - 1 if i > 0 and rows[i - 1][0].startswith("k_frame_sums") else i for i in opens]  # (-Y autolevel: the frame levels come first)
    passes = [rows[a:b] for a, b in zip(opens, opens[1:] + [len(rows)])]
    passes = passes[1:-1] if len(passes) > 3 else passes  # (steady passes: not the first, not the checksum pass behind the timed region)
    import json
    try:
        bj = json.loads(line)
        print(f"== config {cfg}: bench says detect_ms {bj['breakdown_ms'].get('detect_ms')} of ms_per_step {bj['ms_per_step']}; split {bj['config'].get('split')}")
    except Exception:
        print(f"== config {cfg}")
    det_names = ("k_tile_max", "k_frame_sums", "k_wave", "k_capture_weight", "k_order", "k_pkg_scan")
    for pi, p in enumerate(passes):
        det = [r for r in p if r[0].startswith(det_names)]
        if not det:
            continue
        last_wave = max(i for i, r in enumerate(det) if r[0].startswith("k_wave"))
        det = det[: last_wave + 2]  # ... up to the package scan behind the last detection launch
        t0, t1 = det[0][1], det[-1][2]
        busy = sum(e - s for _, s, e, _ in det)
        print(f"-- pass {pi}: detection part {1e-6 * (t1 - t0):.3f} ms from first to last kernel, kernels busy {1e-6 * busy:.3f} ms, device idle between them {1e-6 * (t1 - t0 - busy):.3f} ms")
        print(f"   {'kernel':<44} {'grid':>7} {'start_ms':>9} {'ms':>8}")
        for n, s, e, g in det:
            print(f"   {n:<44} {g:>7} {1e-6 * (s - t0):>9.3f} {1e-6 * (e - s):>8.3f}")
        rest = [r for r in p if r not in det]
        by = {}
        for n, s, e, g in rest:
            by.setdefault(n, [0, 0.0])
            by[n][0] += 1
            by[n][1] += 1e-6 * (e - s)
        print("   behind it (slicers, offsets, copies): " + ", ".join(f"{n} x{k} {t:.3f} ms" for n, (k, t) in sorted(by.items(), key=lambda x: -x[1][1])[:8]))
    for f in dbs:
        if os.path.getsize(f) > (8 << 20):
            os.remove(f)


if __name__ == "__main__":
    for cfg in (sys.argv[1:] or ["3", "5"]):
        one(int(cfg))
