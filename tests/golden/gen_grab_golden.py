"""Golden vectors for the sample grabber (`-S all`) from the REAL reference CLI (oracle/_ref/rtl_433_ref).
TEST INFRASTRUCTURE; run in the build container only:

    python tests/golden/gen_grab_golden.py

Writes tests/golden/grabs.json: for the captures of tests/cases.py grab_capture the g###_433.92M_250k.cu8 files the
reference saves (reference src/r_flow.c:342-362, src/samp_grab.c:100-165): length, SHA-256 and where in the input
those bytes sit."""
import glob
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests.cases import grab_capture  # noqa: E402


def main():
    gold = {}
    for name in ("one_frame", "two_frames"):
        iq = grab_capture(name)
        raw = iq.tobytes()
        with tempfile.TemporaryDirectory() as td:
            iq.tofile(os.path.join(td, "in_433.92M_250k.cu8"))
            r = subprocess.run([po.REF_CLI, "-r", "in_433.92M_250k.cu8", "-S", "all"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            files = sorted(glob.glob(os.path.join(td, "g[0-9]*")))
            gold[name] = []
            for f in files:
                data = open(f, "rb").read()
                gold[name].append({"file": os.path.basename(f), "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                                   "offset_in_input": raw.find(data)})
            said = [ln for ln in r.stderr.decode(errors="replace").splitlines() if "Saving signal" in ln]
            # the same grabs in the SigMF container (`-S sigmf:all`, src/samp_grab.c:166-232, src/sigmf.c): deterministic bytes
            for f in files:
                os.remove(f)
            subprocess.run([po.REF_CLI, "-r", "in_433.92M_250k.cu8", "-S", "sigmf:all"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            for k, f in enumerate(sorted(glob.glob(os.path.join(td, "g[0-9]*.sigmf")))):
                data = open(f, "rb").read()
                gold[name][k]["sigmf_bytes"] = len(data)
                gold[name][k]["sigmf_sha256"] = hashlib.sha256(data).hexdigest()
        print(name, gold[name], said)
    with open(os.path.join(ROOT, "tests", "golden", "grabs.json"), "w") as f:
        json.dump(gold, f, indent=1)


if __name__ == "__main__":
    main()
