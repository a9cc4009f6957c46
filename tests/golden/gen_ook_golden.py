"""Golden vectors for the pulse-data (`.ook`) side door from the REAL reference CLI (oracle/_ref/rtl_433_ref).
TEST INFRASTRUCTURE; run in the build container only:

    python tests/golden/gen_ook_golden.py

Writes
  tests/golden/kat.ook          `rtl_433_ref -r nice_250k.cu8 -w kat.ook` (reference src/pulse_data.c:178-224)
  tests/golden/kat.vcd          the same package through the VCD writer (`-w kat.vcd`)
  tests/golden/ook_flex.json    what `rtl_433_ref -r kat.ook -R 0 -X 'n=raw,m=OOK_PWM,...' -F json` decodes from that
                                text again (reference src/rtl_433.c:1755-1794, src/pulse_data.c:122-176)"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FLEX = "n=raw,m=OOK_PWM,s=500,l=1000,r=5000,g=2000,t=100,y=1500"
PCM_FLEX = "n=raw,m=OOK_PCM,s=100,l=100,r=20000"


def main():
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(os.path.join(GOLD, "nice_250k.cu8"), os.path.join(td, "nice_250k.cu8"))
        subprocess.run([po.REF_CLI, "-r", "nice_250k.cu8", "-w", "kat.ook"], check=True, cwd=td,
                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        shutil.copy(os.path.join(td, "kat.ook"), os.path.join(GOLD, "kat.ook"))
        subprocess.run([po.REF_CLI, "-r", "nice_250k.cu8", "-w", "kat.vcd"], check=True, cwd=td,
                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        shutil.copy(os.path.join(td, "kat.vcd"), os.path.join(GOLD, "kat.vcd"))  # src/pulse_data.c:77-120
        out = subprocess.run([po.REF_CLI, "-s", "250k", "-r", "kat.ook", "-R", "0", "-X", FLEX, "-F", "json"], check=True, cwd=td,
                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    events = [json.loads(line) for line in out.splitlines() if line.startswith("{")]
    # RfRaw lines inside a pulse file (src/rfraw.c): the analyzer's own "view at ...#AAB1..." / "AAB0...+AAB0..." strings
    # of two captures (tests/golden/analyzer.json), read back by the CLI at 1 MS/s and sliced as PWM / as PCM
    ana = json.load(open(os.path.join(GOLD, "analyzer.json")))
    rfraw = {}
    for name, flex in (("kat", FLEX), ("ook_long", PCM_FLEX), ("fsk_cs16_pcm", PCM_FLEX)):
        lines = [ln.split("#", 1)[1] for blk in ana[name]["blocks"] for ln in blk if ln.startswith("view at")]
        text = ";pulse data\n" + "".join(ln + "\n;end\n" for ln in lines)
        with tempfile.TemporaryDirectory() as td:
            open(os.path.join(td, "r.ook"), "w").write(text)
            out = subprocess.run([po.REF_CLI, "-s", "1000k", "-r", "r.ook", "-R", "0", "-X", flex, "-F", "json"], check=True, cwd=td,
                    stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        rfraw[name] = {"text": text, "flex": flex, "events": [json.loads(line) for line in out.splitlines() if line.startswith("{")]}
        print(name, len(lines), "rfraw lines ->", len(rfraw[name]["events"]), "events")
    with open(os.path.join(GOLD, "ook_flex.json"), "w") as f:
        json.dump({"flex": FLEX, "events": events, "rfraw": rfraw}, f, indent=1)
    print(open(os.path.join(GOLD, "kat.ook")).read()[:600])
    print(events)


if __name__ == "__main__":
    main()
