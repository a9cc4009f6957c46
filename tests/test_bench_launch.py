"""bench.py's launcher and collective where there is no GPU: `python bench.py --gpus 2 --backend emu ...` must start two ranks
by itself (re-executing under torch.distributed.run), shard the batches over them, gather every rank's decoded events on rank 0
(gloo here, RCCL on the GPU box: the same shard.gather_rank_records) and print ONE line whose n_gpus is the size of the process
group.  The kernels run on the CPU wave emulator of the test suite with shortened captures: nothing is measured.  Also: a
launcher that started another number of ranks than --gpus names is refused, and the timed path never loads the checker."""
import json
import os
import re
import subprocess
import sys

import pytest

from rtl_433_amd import plugins
from tests.emu import build_emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

pytestmark = pytest.mark.skipif(not build_emu.available() or not plugins.available(),
                                reason="needs the wave emulator (x86-64) and dropin/_build/libr433plugins.so")

TINY = ["--backend", "emu", "--streams", "2", "--batches", "1", "--steps", "2", "--warmup", "0", "--engines", "2", "--threads", "2",
        "--capture-samples", "6144", "--no-cpu-baseline"]


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def _line(p):
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_gathers_both():
    build_emu.build()
    one = _line(_run(["--gpus", "1"] + TINY))
    two = _line(_run(["--gpus", "2"] + TINY))
    assert one["n_gpus"] == 1 and one["config"]["ranks"]["world_size"] == 1
    assert two["n_gpus"] == 2 and two["config"]["ranks"] == {"world_size": 2, "launcher": "torch.distributed.run", "collective_backend": "gloo",
                                                              "devices": ["emu:0", "emu:1"]}
    assert "TEST" in two["backend"] and "not_baseline_workload" in two["config"]
    # weak scaling: every rank has its own batches; rank 0 of the pair ran what the single process ran
    g1, g2 = one["gathered"], two["gathered"]
    assert len(g1) == 1 and len(g2) == 2
    assert [g["first"] for g in g2] == [0, 2]
    assert g2[0]["packages"] == g1[0]["packages"] and g2[0]["events"] == g1[0]["events"]
    assert two["packages_per_step"] == sum(g["packages"] for g in g2) >= 2
    assert two["decoded_events_gathered"]["lines"] == sum(g["events"] for g in g2)


def test_a_launcher_with_another_world_size_is_refused():
    p = _run(["--gpus", "4"] + TINY, env=dict(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"), timeout=120)
    assert p.returncode != 0 and b"--gpus 4" in p.stderr and b"2 rank" in p.stderr


def test_the_timed_path_does_not_know_the_checker():
    """Every mention of the oracle in bench.py sits in a function of the CPU legs (cpu_baseline_*, the parity blocks that run
    after the timed region, real_decoders_leg's CPU side); the pipeline, the plugin loader and main() have none."""
    src = open(BENCH).read()
    assert "_RefPlugins" not in src
    funcs = re.split(r"\n(?=def |class )", src)
    allowed = ("def _ref_worker_init", "def cpu_baseline_config2", "def cpu_baseline_real_decoders", "def real_decoders_leg", "def run_batched", "def dropin_legs",
               "def run_stream", '"""bench.py')
    for f in funcs:
        if "pyoracle" in f or "_ref/" in f.split('"""')[-1]:
            assert f.startswith(allowed), f[:80]
    for name in ("class Pipeline", "def real_decoder_plugins", "def main", "class Backend", "def timed"):
        body = next(f for f in funcs if f.startswith(name))
        assert "pyoracle" not in body and "oracle import" not in body, name
    # run_batched / run_stream import the checker only in their after-the-clock parity blocks
    rb = next(f for f in funcs if f.startswith("def run_batched"))
    assert rb.index("elapsed, (det_ms, tot_ms, disp_s, n_pkgs) = timed(") < rb.index("pyoracle")
