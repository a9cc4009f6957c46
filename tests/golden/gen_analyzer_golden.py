"""Golden text of the pulse analyzer (`-A`) from the REAL reference CLI (oracle/_ref/rtl_433_ref).
TEST INFRASTRUCTURE; run in the build container only:

    python tests/golden/gen_analyzer_golden.py

Writes tests/golden/analyzer.json: per capture of tests/cases.py the analyzer's text block of every package, from
"Analyzing pulses..." through the flex-decoder suggestion (reference src/pulse_analyzer.c:279-556); what the
trial demodulation prints after that is not part of the vector."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests.cases import make_case  # noqa: E402

# no stock decoder, but one FSK flex decoder: FM demodulation stays enabled like in a default run (src/rtl_433.c:1515-1526)
FSK_FLEX = "n=f,m=FSK_PCM,s=52,l=52,r=800"
MAX_BLOCKS = 12  # per capture (the `random` capture has 88 packages)

CASES = ["kat", "ook0", "ook1", "ook2", "ook3", "ook4", "ook5", "ook6", "ook7", "ook_long", "fsk_cu8", "fsk_cu8_minmax", "fsk_cs16",
         "fsk_cs16_pcm", "random", "noise"]


def blocks_of(stderr_text):
    out, cur = [], None
    for ln in stderr_text.splitlines():
        if ln == "Analyzing pulses...":
            cur = [ln]
            continue
        if cur is not None:
            if ln == "" or ln.startswith("\x1b["):
                out.append(cur)
                cur = None
            else:
                cur.append(ln)
    if cur:
        out.append(cur)
    return out


def main():
    gold = {}
    for name in CASES:
        iq, ss, rate, freq = make_case(name)
        fname = "g_%dHz_%dsps.%s" % (freq, rate, "cu8" if ss == 2 else "cs16")
        with tempfile.TemporaryDirectory() as td:
            iq.tofile(os.path.join(td, fname))
            r = subprocess.run([po.REF_CLI, "-r", fname, "-f", str(freq), "-s", str(rate), "-A", "-R", "0", "-X", FSK_FLEX], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        blocks = blocks_of(r.stderr.decode(errors="replace"))
        gold[name] = {"packages": len(blocks), "blocks": blocks[:MAX_BLOCKS]}
        print(name, len(blocks), "packages")
    with open(os.path.join(ROOT, "tests", "golden", "analyzer.json"), "w") as f:
        json.dump(gold, f, indent=0)


if __name__ == "__main__":
    main()
