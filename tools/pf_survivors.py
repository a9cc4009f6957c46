"""What still reaches the decoders behind the device-side pre-filter (CPU: the wave emulator + the plugin library's real
decoders): per decoder the records that crossed, their shape (rows, row lengths) and what decode_fn answered.

    python tools/pf_survivors.py [captures] [seed0]

Used to decide which first-line tests of the decoders are worth learning next (NOTES.md)."""
import collections
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rtl_433_amd import _lib, plugins  # noqa: E402
from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table  # noqa: E402
from tests.emu import host  # noqa: E402


def records(blob):
    at = 0
    while at + 16 <= len(blob):
        total, pkg = np.frombuffer(blob, "<u4", 2, at)
        dev, ordinal, num_rows, free_row = np.frombuffer(blob, "<u2", 4, at + 8)
        rows, p = [], at + 16
        if int(num_rows) == 0xffff:  # a stub: a refusal of a later-level decoder, booked by the replay (kPfStub)
            at += int(total)
            continue
        for _ in range(int(num_rows)):
            bits, syncs, nbytes, _r = np.frombuffer(blob, "<u2", 4, p)
            rows.append((int(bits), int(syncs), bytes(blob[p + 8:p + 8 + int(nbytes)])))
            p += 8 + ((int(nbytes) + 3) & ~3)
        yield int(pkg), int(dev), int(num_rows), int(free_row), rows
        at += int(total)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    devs, protos, names = load_device_table()
    plug = plugins.Plugins()
    iqs = [bench._synth_one(s) for s in range(seed0, seed0 + n)]
    out = {}
    for mode in ("plain", "filtered"):
        eng = BatchEngine(flow_cfg(2, 250000), devs, library=host.emu_lib())
        eng.probe_prefilter(plug.devices, helper=(plug.helper_probe() if os.environ.get("PF_HELPER", "1") == "1" else None))
        eng.set_prefilter(1 if mode == "filtered" else 0)
        eng.run_host(iqs)
        ev, nev = eng.events()
        out[mode] = (bytes(ev), nev)
        eng.close()
    print(f"{n} captures: {out['plain'][1]} records, {out['filtered'][1]} behind the pre-filter")
    # the decoders' answers for what crossed
    RD = bench._RDevice()
    BB = _lib.BitBuffer if hasattr(_lib, "BitBuffer") else None
    by_dev = collections.Counter()
    shape = collections.defaultdict(collections.Counter)
    for pkg, dev, num_rows, free_row, rows in records(out["filtered"][0]):
        by_dev[dev] += 1
        key = (num_rows, tuple(r[0] for r in rows[:3]), tuple(min(r[1], 1) for r in rows[:2]))
        shape[dev][key] += 1
    total = sum(by_dev.values())
    acc = 0
    for dev, cnt in by_dev.most_common(40):
        acc += cnt
        top = ", ".join(f"{k[0]}r{list(k[1])}s{list(k[2])}:{c}" for k, c in shape[dev].most_common(6))
        print(f"{cnt:7d} {100 * cnt / total:5.1f}% cum {100 * acc / total:5.1f}%  [{dev:3d}] mod {devs[dev]['modulation']:2d} {names[dev][:28]:28s} {len(shape[dev]):4d} shapes: {top}")
    # all decoders together: rows histogram, first-row length histogram
    allshape = collections.Counter()
    for dev in shape:
        for k, c in shape[dev].items():
            allshape[(k[0], "len0<=14" if k[1] and k[1][0] <= 14 else "len0<=40" if k[1] and k[1][0] <= 40 else "longer" if k[1] else "none")] += c
    for k, c in sorted(allshape.items(), key=lambda x: -x[1])[:20]:
        print(f"   rows {k[0]:2d} {k[1]:9s} {c:7d} {100 * c / total:5.1f}%")


if __name__ == "__main__":
    main()


def answers(blob, plug, names, limit=None):
    """decode_fn's answer for every record of `blob` (outputs swallowed): Counter[(dev, num_rows, bits0)][code]"""
    RD = bench._RDevice()
    SWALLOW_OUT = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)(lambda d, x: None)
    SWALLOW_LOG = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)(lambda d, l, x: None)
    objs = [C.cast(p, C.POINTER(RD)).contents for p in plug.devices]
    fns = []
    for o in objs:
        o.output_fn = C.cast(SWALLOW_OUT, C.c_void_p).value
        o.log_fn = C.cast(SWALLOW_LOG, C.c_void_p).value
        fns.append(C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)(o.decode_fn))
    buf = (C.c_uint8 * 6604)()
    view = np.frombuffer(buf, dtype=np.uint8)
    u16 = np.frombuffer(buf, dtype="<u2", count=102)
    got = collections.defaultdict(collections.Counter)
    keep = (SWALLOW_OUT, SWALLOW_LOG)
    for k, (pkg, dev, num_rows, free_row, rows) in enumerate(records(blob)):
        if limit and k >= limit:
            break
        view[:] = 0
        u16[0], u16[1] = num_rows, free_row
        for r, (bits, syncs, data) in enumerate(rows):
            u16[2 + r] = bits
            u16[52 + r] = syncs
            at = 204 + 128 * r
            view[at:at + len(data)] = np.frombuffer(data, dtype=np.uint8)[:6604 - at]
        ret = fns[dev](plug.devices[dev], C.addressof(buf))
        got[(dev, num_rows, rows[0][0] if rows else -1)][ret] += 1
    return got, keep
