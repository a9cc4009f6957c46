"""The detection pass in its launch forms -- producer / consumer pairs in one workgroup, and the two roles as two launches
(stream_kernels.hip FORM 4 / FORM 5) with the consumers' launch taking its captures heaviest first by the work the producers left
(StreamParams::cons_weight / k_order_falling): the order and the grouping of the workgroups are all that change.  EVERY form is
compared with the oracle's package records for the same captures, byte for byte (reference src/pulse_detect.c:199-483 is per
capture; src/rtl_433.c:1845-1854 resets the flow between files).
(Round 6 tried a third form -- the grid in slices, the consumers of slice k on a second stream beside the producers of slice
k + 1 -- bit-exact here and 1.1 ms SLOWER per extra slice on the MI355X: a launch of producers lasts as long as its heaviest
capture however few captures it holds; profiles/r06_b_roles_slices_dropped.txt.  Not kept.)"""
import os

import numpy as np
import pytest

import bench
from oracle import pyoracle as po
from rtl_433_amd import synth
from rtl_433_amd.engine import BatchEngine, flow_cfg
from tests.emu import build_emu

BACKENDS = [pytest.param("emu", marks=pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")),
            pytest.param("gpu", marks=pytest.mark.gpu)]
SPLIT_ROLES, FORCE_ORDER, NO_ORDER = 4194304, 128, 64  # R433_DEBUG_* (include/r433_hip.h)


def _oracle_packages(iqs):
    cfg = po.default_flow_cfg(2, 250000, fpdm=0)
    blob, base = b"", 0
    for s, a in enumerate(iqs):
        o = po.oracle_flow(a, None, cfg, stream_index=s, pkg_base=base)
        blob += o["packages"]
        base += o["n_packages"]
    return base, blob


@pytest.mark.parametrize("backend", BACKENDS)
def test_every_launch_form_against_the_oracle(backend):
    n = 20 if backend == "emu" else 600
    iqs = [bench._synth_one(s) for s in range(n)] + [synth.noise_cu8(5, 65536), synth.random_cu8(7, 65536), synth.ook_stream(3, 30000)[0]]
    want = _oracle_packages(iqs)
    got = {}
    for name, flags in (("pairs", 0), ("two launches", SPLIT_ROLES), ("two launches, ordered", SPLIT_ROLES | FORCE_ORDER),
                        ("two launches, capture order", SPLIT_ROLES | NO_ORDER)):
        if backend == "gpu":
            eng = BatchEngine(flow_cfg(2, 250000), None)
        else:
            from tests.emu import host
            eng = BatchEngine(flow_cfg(2, 250000), None, library=host.emu_lib())
        if flags:
            eng.set_debug(flags)
        npk = eng.run_host(iqs)
        got[name] = (npk, bytes(eng.packages()[0]), eng.split_stats()["detect_form"])
        eng.close()
    assert got["pairs"][2] != 45 and got["two launches, ordered"][2] == 45
    for name, (npk, blob, _) in got.items():
        assert npk == want[0], (name, npk, want[0])
        assert blob == want[1], name
