"""dropin/pipeline_host.c: a C host that keeps several engines in flight over a list of capture files (file reading, GPU legs
and the ordered replay into the reference's own decoders overlap).  What it prints -- the decoders' messages through the
reference's data_print_jsons, in list order -- must not depend on the number of engines or the size of a pass, and must be
what the stock binary decodes from the same list (modulo the `time` field and the CLI's float formatting, which are the
CLI's: src/r_api.c:632-840)."""
import json
import os
import subprocess

import pytest

from rtl_433_amd import protocols as P
from tests.emu import build_emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "rtl_433_ref")
EMU = os.path.join(ROOT, "dropin", "_build", "pipeline_host_emu")
HIP = os.path.join(ROOT, "dropin", "_build", "pipeline_host_hip")


def _files(d, rounds):
    files = []
    for seed in range(rounds):
        for name in sorted(P.PROTOCOLS):
            if name == "generic_remote":  # (pairs up with secplus_v1 by wall clock: tests/test_corpus.py)
                continue
            iq, meta = P.transmission(name, seed)
            if meta["rate"] != 250000 or meta["freq"] != 433920000 or iq.dtype.itemsize != 1:
                continue
            fn = P.file_name(name, seed, meta["rate"], meta["freq"])
            iq.tofile(os.path.join(d, fn))
            files.append(fn)
    first = [f for f in files if f.startswith("p_secplus")]
    return first + [f for f in files if not f.startswith("p_secplus")]


def _messages(text):
    out = []
    for line in text.splitlines():
        m = json.loads(line)
        m.pop("time", None)
        out.append({k: (round(v, 3) if isinstance(v, float) else v) for k, v in m.items()})
    return out


def _run(binary, args, files, cwd, env=None, want_stderr=None):
    p = subprocess.run([binary] + args + files, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-1500:]
    if want_stderr:
        assert want_stderr in p.stderr.decode(errors="replace"), p.stderr.decode(errors="replace")[-600:]
    return p.stdout.decode()


def _check(binary, tmp_path, rounds, shapes):
    if not (os.path.exists(binary) and os.path.exists(REF)):
        pytest.skip("dropin/_build/pipeline_host_* or oracle/_ref not built (needs /root/reference once: python __graft_entry__.py)")
    files = _files(str(tmp_path), rounds)
    ref = subprocess.run([REF] + sum((["-r", f] for f in files), []) + ["-F", "json"], cwd=tmp_path, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=1800)
    want = _messages(ref.stdout.decode())
    assert len(want) >= len(files) - 2 and len({m["model"] for m in want}) >= 15
    first = None
    for args in shapes:
        out = _run(binary, args, files, tmp_path)
        first = out if first is None else first
        assert out == first, args  # byte for byte, whatever the pipeline's shape
    assert _messages(first) == want


@pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")
def test_pipeline_host_on_the_emulator(tmp_path):
    _check(EMU, tmp_path, 1, [["-e", "1", "-b", "5", "-t", "4"], ["-e", "3", "-b", "4", "-t", "4"], ["-e", "2", "-b", "7", "-t", "2", "-p"]])


@pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")
def test_pipeline_host_over_two_devices_on_the_emulator(tmp_path):
    """-g N: engines on several GPUs (r433_batch_create_on), passes to the engines in turn, ONE ordered replay: the output of
    two (pretend) devices is the output of one, byte for byte, with and without the stateless decoders spread over threads."""
    if not (os.path.exists(EMU) and os.path.exists(REF)):
        pytest.skip("dropin/_build/pipeline_host_emu or oracle/_ref not built")
    files = _files(str(tmp_path), 1)
    one = _run(EMU, ["-g", "1", "-e", "2", "-b", "4", "-t", "4"], files, tmp_path, want_stderr="on 1 of 1 visible GPU")
    two_env = {"R433_EMU_DEVICES": "2"}
    two = _run(EMU, ["-g", "2", "-b", "4", "-t", "4"], files, tmp_path, env=two_env, want_stderr="4 engine(s) on 2 of 2 visible GPU")
    assert two == one
    assert _run(EMU, ["-g", "0", "-e", "3", "-b", "5", "-t", "3", "-p", "-o"], files, tmp_path, env=two_env, want_stderr="on 2 of 2 visible GPU") == one
    # a device that is not there is an error, not a silent fallback
    p = subprocess.run([EMU, "-g", "1", "-e", "1"] + files[:1], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, R433_EMU_DEVICES="1"))
    assert p.returncode == 0


@pytest.mark.gpu
def test_pipeline_host_on_the_gpu(tmp_path):
    _check(HIP, tmp_path, 12, [["-e", "1", "-b", "64"], ["-e", "3", "-b", "32"], ["-e", "4", "-b", "17", "-p"], ["-e", "2", "-b", "500", "-t", "32"],
                               ["-g", "0", "-b", "40", "-o"]])  # -g 0: every visible GPU (one on the test box: the same code path)
