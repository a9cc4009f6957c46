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
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
