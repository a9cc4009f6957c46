"""The detection pass as two launches (producers, consumers: stream_kernels.hip FORM 4 / FORM 5) with the consumers' launch taking
its captures heaviest first by the work the producers left (StreamParams::cons_weight, counted by the producers, / k_order_falling): the order of the workgroups
is all that changes -- package records are those of the one-launch form, byte for byte (reference src/pulse_detect.c:199-483
is per capture; src/rtl_433.c:1845-1854 resets the flow between files)."""
import zlib

import numpy as np
import pytest

import bench
from rtl_433_amd import synth
from rtl_433_amd.engine import BatchEngine, flow_cfg
from tests.emu import build_emu

BACKENDS = [pytest.param("emu", marks=pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")),
            pytest.param("gpu", marks=pytest.mark.gpu)]
SPLIT_ROLES, FORCE_ORDER, NO_ORDER = 4194304, 128, 64  # R433_DEBUG_* (include/r433_hip.h)


@pytest.mark.parametrize("backend", BACKENDS)
def test_consumers_in_their_own_order(backend):
    n = 20 if backend == "emu" else 600
    iqs = [bench._synth_one(s) for s in range(n)] + [synth.noise_cu8(5, 65536), synth.random_cu8(7, 65536), synth.ook_stream(3, 30000)[0]]
    got = {}
    for name, flags in (("one launch", 0), ("two launches", SPLIT_ROLES), ("two launches, ordered", SPLIT_ROLES | FORCE_ORDER),
                        ("two launches, capture order", SPLIT_ROLES | NO_ORDER)):
        if backend == "gpu":
            eng = BatchEngine(flow_cfg(2, 250000), None)
        else:
            from tests.emu import host
            eng = BatchEngine(flow_cfg(2, 250000), None, library=host.emu_lib())
        if flags:
            eng.set_debug(flags)
        npk = eng.run_host(iqs)
        got[name] = (npk, zlib.crc32(eng.packages()[0]), eng.split_stats()["detect_form"])
        eng.close()
    assert got["one launch"][2] != 45 and got["two launches, ordered"][2] == 45
    assert len({v[:2] for v in got.values()}) == 1, got
