"""Decoded-JSON parity on protocol-valid traffic (VERDICT r2 missing #3 / task 6): transmissions of 22 real protocols over nine
modulations (rtl_433_amd/protocols.py; every frame is one its decoder accepts) through the drop-in CLI and through the
stock reference binary, all default decoders registered, `-F json -M level -M protocol -M stats`: stdout byte for byte.
Covers data_acquired_handler's fields (src/r_api.c:632-840: mod, freq / freq1 / freq2, rssi, snr, noise, protocol), the
priority rule on SUCCESSFUL events (src/r_api.c:442-451: a Rubicson frame never reaches the Nexus decoder, a Nexus frame
does), a decoder that keeps state between calls (src/devices/secplus_v1.c:142: two packages make one message), the
per-decoder statistics with the device-side pre-filter on, and mixed sample rates / frequencies in one file list."""
import json
import os

import pytest

from rtl_433_amd import protocols as P
from tests.emu import build_emu
from tests.test_dropin import EMU, HIP, REF, _ensure_built, file_args, run_cli

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/rtl_433_ref not built (needs /root/reference once)")


def write_corpus(d, n_files, names=None):
    """n_files transmissions, the protocols taking turns.  The files of the stateful decoder go FIRST: secplus_v1 pairs the
    halves it is handed within 800 ms of WALL CLOCK (src/devices/secplus_v1.c:140,196-215), and a Generic-Remote burst
    happens to look like one half to it -- with such a stray half pending, what the reference itself prints for the next
    Secplus file depends on how fast the machine is."""
    # (Generic-Remote stays out of the long lists for the same reason: two of its bursts, files apart, pair up as one
    # Secplus message or do not, by the clock.  It is in the short list of the emulator test, where nothing follows it.)
    names = names or [n for n in sorted(P.PROTOCOLS) if n != "generic_remote"]
    files, want = [], []
    for k in range(n_files):
        name = names[k % len(names)]
        seed = k // len(names)
        iq, meta = P.transmission(name, seed)
        fn = P.file_name(name, seed, meta["rate"], meta["freq"])
        iq.tofile(os.path.join(d, fn))
        files.append(fn)
        want.append(meta["model"])
    first = [f for f in files if f.startswith("p_secplus")]
    return first + [f for f in files if not f.startswith("p_secplus")], want


def models_of(stdout):
    out = []
    for line in stdout.splitlines():
        try:
            m = json.loads(line).get("model")
        except ValueError:
            continue
        if m:
            out.append(m)
    return out


def check_corpus(binary, tmp_path, n_files, min_models, extra_env=None, names=None):
    files, want = write_corpus(tmp_path, n_files, names)
    args = file_args(files) + ["-F", "json", "-M", "level", "-M", "protocol", "-M", "stats", "-K", "FILE"]
    ref = run_cli(REF, args, tmp_path)
    got = run_cli(binary, args, tmp_path, extra_env)
    seen = set(models_of(ref))
    assert len(seen & set(want)) >= min_models, sorted(seen)
    assert got == ref
    return ref, seen


def test_emu_corpus_subset(tmp_path):
    """nine protocols, one file each, on the emulator build: the stateful decoder, the priority pair, OSv1, DMC, an FSK one"""
    if not build_emu.available():
        pytest.skip("wave emulator needs x86-64")
    _ensure_built(EMU)
    names = ["rubicson", "nexus", "secplus_v1", "generic_remote", "oregon_v1", "wt450", "ambient_f007th", "efergy_e2", "bresser_3ch"]
    ref, seen = check_corpus(EMU, tmp_path, len(names), 9, {"RTL433_HIP_PREFILTER": "1"}, names=names)
    assert "Rubicson-Temperature" in seen and "Nexus-TH" in seen and "Secplus-v1" in seen
    # the Rubicson file decoded at priority 0, so Nexus (priority 10) never saw it; the Nexus file has no Rubicson event
    lines = [json.loads(x) for x in ref.splitlines() if x.startswith('{"tag"') or '"model"' in x]
    by_tag = {}
    for x in lines:
        if "model" in x and "tag" in x:
            by_tag.setdefault(x["tag"], []).append(x["model"])
    rub = [t for t in by_tag if t.startswith("p_rubicson")][0]
    nex = [t for t in by_tag if t.startswith("p_nexus")][0]
    assert set(by_tag[rub]) == {"Rubicson-Temperature"} and set(by_tag[nex]) == {"Nexus-TH"}


@pytest.mark.gpu
def test_hip_corpus_256_files(tmp_path):
    """256 files, 22 protocols, nine modulations through rtl_433_hip (device-side pre-filter on) and the stock binary"""
    _ensure_built(HIP)
    ref, seen = check_corpus(HIP, tmp_path, 256, 20, {"RTL433_HIP_PREFILTER": "1"})  # (by itself the CLI asks only from 8 GiB of samples on)
    assert "Secplus-v1" in seen and "Rubicson-Temperature" in seen and "Nexus-TH" in seen
    assert '"mod" : "FSK"' in ref and '"mod" : "ASK"' in ref


@pytest.mark.gpu
def test_hip_corpus_without_prefilter_and_per_file(tmp_path):
    """the same with every record crossing to the host, and with one GPU pass per file"""
    _ensure_built(HIP)
    check_corpus(HIP, tmp_path, 66, 20, {"RTL433_HIP_PREFILTER": "0"})
    check_corpus(HIP, tmp_path, 44, 20, {"RTL433_HIP_BATCH": "1"})
