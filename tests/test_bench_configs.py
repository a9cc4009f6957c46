"""BASELINE.json configs[2] and configs[4] at FULL size through bench.py on the GPU: one 64 Mi-sample 1024 kS/s cs16 FSK stream
(min/max detector, Manchester decoders) and one 256 Mi-sample 2 MS/s cu8 stream (-Y autolevel, -Y filter, every decoder),
each spread over the chip by verified cuts, the reference's real decoders (+ the config's flex decoder) behind the path -- their
JSON lines over the whole stream against the unmodified reference's (oracle/_ref) by SHA-256 and every bitbuffer by checksum, the
same line bench.py prints for `--config 3` / `--config 5`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("config, samples", [(3, 64 << 20), (5, 256 << 20)])
def test_full_size_stream_vs_reference(config, samples):
    from oracle import pyoracle as po
    if not po.have_ref():
        pytest.skip("oracle/_ref did not travel")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--steps", "2", "--warmup", "1"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["config"]["samples"] == samples
    # the reference's real decoders behind the path: their JSON lines over the whole stream by SHA-256, every bitbuffer by checksum
    assert line["parity"] == "decoded-json-sha256-match; bitbuffers: digest-match", line["parity"]
    assert line["parity_detail"]["gpu_sha256"] == line["parity_detail"]["cpu_sha256"]
    assert line["packages_per_step"] > 100 and line["bitbuffers_per_step"] > 1000 and line["decoded_messages_per_step"] > 0
