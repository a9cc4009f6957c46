"""Makes rtl_433_amd/data/protocol_frames.json: frames the reference's real decoders accept, for rtl_433_amd/protocols.py.

    python tools/gen_protocol_frames.py [name ...]        (needs oracle/_ref/libr433ref.so, i.e. /root/reference)

For every protocol a TEMPLATE says what a transmission looks like to its decoder -- how many rows of how many bits, which
bits are fixed (preambles, type nibbles), which fields are free -- and nothing about its integrity fields: those are found
by handing candidate bitbuffers to the decoder's own decode_fn (taken from the reference build, src/devices/*.c) and
trying the bytes of the row until it says yes.  A decoder that also looks at value ranges gets fresh random payloads until
one passes.  The frames are therefore the decoder's own idea of a valid message, and this file restates no checksum.

Every frame is then keyed (rtl_433_amd.protocols), run through the reference's whole path (oracle/_ref) and kept only if
the decoder really fires on the signal.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from rtl_433_amd import protocols as P  # noqa: E402


class BitBuffer(C.Structure):  # bitbuffer_t, reference include/bitbuffer.h:34-40
    _fields_ = [("num_rows", C.c_uint16), ("free_row", C.c_uint16), ("bits_per_row", C.c_uint16 * 50),
                ("syncs_before_row", C.c_uint16 * 50), ("bb", (C.c_uint8 * 128) * 50)]


DECODE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(BitBuffer))


class Decoder:
    """decode_fn of one protocol of the reference build, callable on bitbuffers made here"""

    def __init__(self, protocol_num):
        self.ref = po.Ref(protocols=[protocol_num], call_real=True, record=False)
        plain = self.ref.plain_devices()
        assert len(plain) >= 1, f"protocol {protocol_num} not registered"
        self.dev = plain[0]
        fn_addr = C.cast(self.dev + 48, C.POINTER(C.c_void_p)).contents.value  # r_device.decode_fn (tests/test_abi.py pins the offset)
        self.fn = DECODE_FN(fn_addr)
        self.bb = BitBuffer()

    def __call__(self, rows):
        """rows: list of (nbits, bytes) -> decode_fn's return value"""
        bb = self.bb
        C.memset(C.byref(bb), 0, C.sizeof(bb))
        bb.num_rows = bb.free_row = len(rows)
        for r, (nbits, data) in enumerate(rows):
            bb.bits_per_row[r] = nbits
            C.memmove(bb.bb[r], bytes(data), min(len(data), 128))
        return self.fn(self.dev, C.byref(bb))


def pack(bits):
    n = len(bits)
    by = bytearray((n + 7) // 8)
    for i, b in enumerate(bits):
        if b:
            by[i >> 3] |= 0x80 >> (i & 7)
    return n, by


def unpack(nbits, by):
    return [(by[i >> 3] >> (7 - (i & 7))) & 1 for i in range(nbits)]


def solve(dec, rows_bits, same_rows=True, fixed=(), max_pairs=True):
    """Make decode_fn accept: try every value of every byte of row 0 (mirrored into the other rows when they repeat it),
    then every pair of neighbouring bytes.  `fixed`: byte indices that must stay.  -> rows as bit lists, or None."""
    rows = [pack(r) for r in rows_bits]

    def attempt():
        return dec([(n, bytes(b)) for n, b in rows])

    def put(pos, v):
        for n, b in (rows if same_rows else rows[:1]):
            if pos < len(b):
                b[pos] = v
    if attempt() > 0:
        return [unpack(n, b) for n, b in rows]
    n0, b0 = rows[0]
    last_mask = 0xff << ((8 - n0 % 8) % 8) & 0xff  # bits of the last byte that belong to the row
    for pos in range(len(b0)):
        if pos in fixed:
            continue
        keep = b0[pos]
        for v in range(256):
            if pos == len(b0) - 1 and (v & ~last_mask):
                continue
            put(pos, v)
            if attempt() > 0:
                return [unpack(n, b) for n, b in rows]
        put(pos, keep)
    if max_pairs:
        for pos in range(len(b0) - 1):
            if pos in fixed or pos + 1 in fixed:
                continue
            k0, k1 = b0[pos], b0[pos + 1]
            for v in range(65536):
                if pos + 1 == len(b0) - 1 and ((v & 0xff) & ~last_mask):
                    continue
                put(pos, v >> 8)
                put(pos + 1, v & 0xff)
                if attempt() > 0:
                    return [unpack(n, b) for n, b in rows]
            put(pos, k0)
            put(pos + 1, k1)
    return None


def check_signal(name, frame, seed=0):
    """the keyed frame through the reference CLI itself (oracle/_ref/rtl_433_ref, all default decoders): does the model
    come out?  -> number of JSON lines with this protocol's model"""
    import subprocess
    import tempfile
    p = P.PROTOCOLS[name]
    rng = np.random.default_rng(seed)
    iq = P.render_cu8(p["schedule"](frame), p["rate"], rng, fsk=p["fsk"], **p.get("render", {}))
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, P.file_name(name, seed, p["rate"], p["freq"]))
        with open(fn, "wb") as f:
            f.write(iq.tobytes())
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "rtl_433_ref"), "-r", fn, "-F", "json"] + p.get("cli", []),
                           capture_output=True, text=True)
    hits = 0
    for line in r.stdout.splitlines():
        try:
            hits += json.loads(line).get("model") == p["model"]
        except ValueError:
            pass
    return hits


# ---------------------------------------------------------------- templates: name -> fn(rng) -> rows of bits the DECODER sees

TEMPLATES = {}


def template(name, **kw):
    def deco(fn):
        TEMPLATES[name] = dict(fn=fn, **kw)
        return fn
    return deco


def rnd_bits(rng, n):
    return [int(x) for x in rng.integers(0, 2, n)]


def from_int(v, n):
    return [(v >> (n - 1 - i)) & 1 for i in range(n)]


def load_templates():
    import tools.protocol_templates  # noqa: F401  (fills TEMPLATES and P.PROTOCOLS' frame -> decoder-row maps)


def main(names):
    load_templates()
    out = {}
    if os.path.exists(P.FRAMES_PATH):
        out = json.load(open(P.FRAMES_PATH))
    todo = names or sorted(TEMPLATES)
    for name in todo:
        t = TEMPLATES[name]
        p = P.PROTOCOLS[name]
        dec = Decoder(p["protocol"])
        rng = np.random.default_rng(sum(map(ord, name)))
        good = []
        tries = 0
        while len(good) < t.get("count", 6) and tries < t.get("tries", 60):
            tries += 1
            rows = t["fn"](rng)
            if t.get("raw"):  # a transmission that is not one bitbuffer (several packages): only the signal check applies
                sol = rows
                frame = ["".join(map(str, r)) for r in rows]
            else:
                sol = solve(dec, rows, same_rows=t.get("same_rows", True), fixed=t.get("fixed", ()), max_pairs=t.get("pairs", True))
                if sol is None:
                    continue
                frame = t["to_frame"](sol) if "to_frame" in t else "".join(map(str, sol[0]))
            if check_signal(name, frame, seed=len(good)) <= 0:
                print(f"  {name}: the decoder takes the bitbuffer but not the signal ({frame if isinstance(frame, str) else frame[0]})")
                continue
            good.append(frame)
        print(f"{name}: {len(good)} frames in {tries} tries")
        if good:
            out[name] = good
    with open(P.FRAMES_PATH, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    from tools import gen_protocol_frames as _self  # the templates register with the module they import, not with __main__
    _self.main(sys.argv[1:])
