"""Turns a rocprofv3 results .db (rocpd sqlite) into the plain-text per-kernel summary kept under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    lines = []
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    lines.append(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  kernel   (durations are in microseconds)")
    for name, calls, total, avg, pct in cur:
        lines.append(f"{calls:>6} {total:>12.1f} {avg:>12.2f} {pct:>7.2f}  {name}")
    # the dominant kernel dispatch by dispatch: alone on the device, or with kernels of other queues running beside it
    # (under rocprofv3 a host -> device copy is a blit kernel, __amd_rocclr_copyBuffer, that takes CUs away)
    try:
        import re
        rows = list(c.execute("select name, start, end from kernels order by start"))
        tops = [r[0] for r in c.execute("select name from top_kernels")]
        top = next((n for n in tops if "r433::" in n), tops[0])  # the library's heaviest kernel (under the profiler copies are kernels too)
        mine = [r for r in rows if r[0] == top]
        alone, beside = [], {}
        for k in mine:
            who = set()
            for r in rows:
                if r is not k and min(r[2], k[2]) - max(r[1], k[1]) > 0.02 * (k[2] - k[1]):
                    who.add(re.search(r"(k_\w+|__amd_\w+|\w+)", r[0].replace("void ", "").replace("r433::(anonymous namespace)::", "")).group(0))
            if who:
                beside.setdefault(", ".join(sorted(who)), []).append((k[2] - k[1]) / 1e3)
            else:
                alone.append((k[2] - k[1]) / 1e3)
        lines.append("")
        import re
        m = re.search(r"k_\w+(<[^(]*>)?", top)
        lines.append(f"{m.group(0) if m else top}: dispatch by dispatch")
        if alone:
            lines.append(f"  {len(alone):>4} dispatches with nothing else on the device: mean {sum(alone) / len(alone):.2f} us (min {min(alone):.2f}, max {max(alone):.2f})")
        for who, d in beside.items():
            lines.append(f"  {len(d):>4} dispatches beside [{who}]: mean {sum(d) / len(d):.2f} us (min {min(d):.2f}, max {max(d):.2f})")
    except (sqlite3.Error, TypeError) as e:
        lines.append(f"(no per-dispatch split: {e})")
    try:
        cur = c.execute("select name, min(vgpr_count), min(sgpr_count), min(lds_size), min(scratch_size), min(grid_x), min(workgroup_x) "
                        "from kernels group by name")
        lines.append("")
        lines.append("per-dispatch resources: vgpr sgpr lds scratch grid_x wg_x  kernel")
        for r in cur:
            lines.append(f"{r[1]:>5} {r[2]:>5} {r[3]:>7} {r[4]:>6} {r[5]:>9} {r[6]:>5}  {r[0]}")
    except sqlite3.Error as e:
        lines.append(f"(no per-dispatch resource table: {e})")
    try:
        cur = c.execute("select name, count(*), avg(value), sum(value) from pmc_events group by name order by name")
        rows = list(cur)
        if rows:
            lines.append("")
            lines.append("PMC counters (name, samples, mean per dispatch, sum)")
            for r in rows:
                lines.append(f"{r[0]:<28} {r[1]:>6} {r[2]:>18.1f} {r[3]:>20.1f}")
    except sqlite3.Error:
        pass
    # The detection pass of a large grid is three launches (k_wave<..., 4> producers, <..., 5> consumers, <..., 2> run-again):
    # what bench.py times as `k_wave_timed_region` and prices as `roofline.achieved` is their sum, pass by pass.
    try:
        rows = list(c.execute("select name, start, end from kernels where name like '%k_wave%' order by start"))

        def form(n):
            return n[n.index("k_wave<"):].split(">")[0].split(",")[-1].strip()
        passes, cur = [], None
        for n, s, e in rows:
            f = form(n)
            if f == "4":
                cur = {"4": e - s, "t0": s}
            elif cur is not None and f in ("5", "2"):
                cur[f] = cur.get(f, 0) + (e - s)
                cur["t1"] = e
                if f == "2":
                    passes.append(cur)
                    cur = None
        if passes:
            lines.append("")
            lines.append(f"detection pass = producers + consumers + run-again, {len(passes)} passes:")
            for k, nm in (("4", "producers  k_wave<..,4>"), ("5", "consumers  k_wave<..,5>"), ("2", "run-again  k_wave<..,2>")):
                v = [p.get(k, 0) / 1e3 for p in passes]
                lines.append(f"  {nm}: mean {sum(v) / len(v):9.2f} us (min {min(v):.2f}, max {max(v):.2f})")
            tot = [(p.get("4", 0) + p.get("5", 0) + p.get("2", 0)) / 1e3 for p in passes]
            span = [(p["t1"] - p["t0"]) / 1e3 for p in passes]
            best = sorted(tot)[: max(1, len(tot) // 2)]
            lines.append(f"  sum of the three: mean {sum(tot) / len(tot):.2f} us, the faster half of the passes (nothing beside them) {sum(best) / len(best):.2f} us; "
                         f"first start to last end: mean {sum(span) / len(span):.2f} us")
    except (sqlite3.Error, ValueError, KeyError) as e:
        lines.append(f"(no per-pass sum: {e})")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
