"""Development aid (GPU box): how many tiles of the pieces of the two single-stream workloads go by unfiltered (lazy tiles inside
split captures), on shortened streams:  python tools/split_quiet_stats.py [Mi samples]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rtl_433_amd.engine import BatchEngine, flow_cfg
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, host, cfg in (("config 3 (cs16 FSK, min/max)", bench.fsk_stream_config3(mi << 20), flow_cfg(4, 1024000, fpdm=1, center_frequency=868000000)),
                        ("config 5 (cu8 mixed, autolevel)", bench.mixed_stream_config5(2 * mi << 20), flow_cfg(2, 2000000, fpdm=0, auto_level=1.0, fm_low_pass=0.15))):
    d = torch.from_numpy(host.view(np.uint8)).cuda().reshape(1, -1)
    for flags, what in ((0, "lazy"), (262144, "R433_DEBUG_NO_LAZY")):
        eng = BatchEngine(cfg, None, profiling=True)
        if flags:
            eng.set_debug(flags)
        for rep in range(3):
            n = eng.run(d)
        st = eng.split_stats()
        slots = st["segments"]
        buf = np.zeros(slots * 64, dtype=np.int32)
        sz = eng.L.r433_batch_debug_state(eng.h, C.c_void_p(buf.ctypes.data), buf.nbytes)
        s = buf[: slots * sz // 4].reshape(slots, sz // 4)
        tiles = host.size // (2 if "cs16" in name else 2) // 2048
        print(f"{name}: {what}: detect {eng.timing()['detect_ms']:.3f} ms, {n} packages, {st}, unfiltered tiles {int(s[:, 2].sum())} of ~{tiles} per variant, pieces run twice {int((s[:, 3] > 0).sum())}")
        eng.close()
