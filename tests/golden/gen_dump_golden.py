"""Golden vectors for the -w dump formats from the REAL reference (oracle/_ref/rtl_433_ref, built from
/root/reference).  TEST INFRASTRUCTURE; run in the build container only:

    python tests/golden/gen_dump_golden.py

Writes tests/golden/dumps.json: for two seeded synthetic captures (tests/cases.py dump_capture) the SHA-256 and
length of every file `rtl_433_ref -r capture -w out.<fmt>` produced (reference src/r_flow.c:385-489)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from tests.cases import dump_capture  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    out = {}
    for name in ("cu8", "cs16"):
        iq, ss, fname = dump_capture(name)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, fname)
            iq.tofile(path)
            out[name] = {}
            for fmt in po.DUMP_FORMATS:
                # one dumper per run: the reference converts into demod->buf.temp, which is a union with buf.fm
                # (include/r_private.h:32-36), so an IQ dumper listed before fm.s16 / fm.f32 clobbers what those write
                subprocess.run([po.REF_CLI, "-r", path, "-w", os.path.join(td, "o." + fmt)], check=True,
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
                data = open(os.path.join(td, "o." + fmt), "rb").read()
                out[name][fmt] = {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()}
    with open(os.path.join(ROOT, "tests", "golden", "dumps.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1)[:400])


if __name__ == "__main__":
    main()
