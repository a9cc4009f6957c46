"""The N>1 path on CPU: two processes (gloo), captures sharded contiguously, every rank decodes its
shard (through the emulator build of the kernels -- there is no GPU here), records gathered to rank 0,
rank 0 checks the merged stream against one oracle pass over the whole list."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.emu import build_emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_caps, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import struct

    import torch.distributed as dist
    from oracle import pyoracle as po
    from rtl_433_amd import shard, synth
    from rtl_433_amd.engine import load_device_table
    from tests.emu import host
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        devs = load_device_table()[0][:50]
        caps = [synth.ook_stream(300 + i, 9000 + 531 * i)[0] for i in range(n_caps)]
        b = shard.partition(n_caps, world)
        mine = caps[b[rank]:b[rank + 1]]
        g = host.emu_run(mine, 2, 250000, devs)
        pk, npk = g["packages"]
        ev, _ = g["events"]
        payload = struct.pack("<III", b[rank], npk, len(pk)) + pk + ev
        got = shard.gather_bytes(payload, dst=0)
        if rank == 0:
            per_rank = []
            for blob in got:
                first, n, lp = struct.unpack_from("<III", blob)
                per_rank.append((first, n, blob[12:12 + lp], blob[12 + lp:]))
            pk_all, ev_all = shard.merge_rank_records(per_rank)
            cfg = po.default_flow_cfg(2, 250000)
            pk_o, ev_o, base = b"", b"", 0
            for s, a in enumerate(caps):
                o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
                pk_o += o["packages"]
                ev_o += o["events"]
                base += o["n_packages"]
            out_q.put((pk_all == pk_o, ev_all == ev_o, base))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    import torch.multiprocessing as mp
    build_emu.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_pk, ok_ev, n = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert n >= 5 and ok_pk and ok_ev


def test_partition_is_contiguous_and_complete():
    from rtl_433_amd import shard
    for n in (0, 1, 7, 1024, 65536):
        for w in (1, 2, 3, 8):
            b = shard.partition(n, w)
            assert b[0] == 0 and b[-1] == n and len(b) == w + 1
            assert all(0 <= b[i + 1] - b[i] <= (n + w - 1) // w for i in range(w))


def _worker_bench_gather(rank, world, port, out_q):
    """bench.py's collective, exactly: shard.pack_rank_record -> shard.gather_rank_records (config 2: real package
    records travel; config 4: counts + checksums), two ranks over gloo, kernels on the emulator."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import pyoracle as po
    from rtl_433_amd import shard, synth
    from rtl_433_amd.engine import load_device_table
    from tests.emu import host
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        devs = load_device_table()[0][:40]
        n_list = 6
        caps = [synth.ook_stream(500 + i, 8000 + 700 * i)[0] for i in range(n_list)]
        b = shard.partition(n_list, world)
        g = host.emu_run(caps[b[rank]:b[rank + 1]], 2, 250000, devs)
        pk, npk = g["packages"]
        ev, nev = g["events"]
        dg = po.events_digest2(ev)[0]
        got = shard.gather_rank_records(shard.pack_rank_record(b[rank], npk, nev, dg, len(pk), pk), True, dst=0)
        if rank == 0:
            cfg = po.default_flow_cfg(2, 250000)
            pk_o, base = b"", 0
            for s, a in enumerate(caps):
                o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
                pk_o += o["packages"]
                base += o["n_packages"]
            per = got["per_rank"]
            out_q.put((got["merged"] == pk_o, [p["first"] for p in per] == b[:-1], sum(p["packages"] for p in per) == base,
                       all(p["extra"] == len(p["pk"]) for p in per)))
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_bench_record_gather():
    import torch.multiprocessing as mp
    build_emu.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench_gather, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(res), res


def _worker_decoded_gather(rank, world, port, out_q):
    """configs[3] as bench.py --config 4 runs it: every rank decodes its shard of the list with the reference's real
    decoders (dropin/_build/libr433plugins.so) behind the ordered replay, serialises what they report as JSON lines and the
    ONE collective moves those bytes; rank 0's concatenation must be what a single process prints for the whole list."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from rtl_433_amd import plugins, protocols, shard, synth
    from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table
    from tests.emu import host
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        devs = load_device_table()[0]
        caps = [protocols.bench_capture(3 * i + 2)[0] if i % 2 else synth.ook_stream(700 + i)[0] for i in range(6)]

        def decode(part):
            plug = plugins.Plugins()
            eng = BatchEngine(flow_cfg(2, 250000), devs, library=host.emu_lib())
            eng.probe_prefilter(plug.devices, helper=plug.helper_probe())
            n = eng.run_host(part)
            eng.dispatch_ordered(plug.devices, None, 2)
            text, n_msg = plug.take()
            eng.close()
            plug.close()
            return n, n_msg, text
        b = shard.partition(len(caps), world)
        npk, n_msg, text = decode(caps[b[rank]:b[rank + 1]])
        got = shard.gather_rank_records(shard.pack_rank_record(b[rank], npk, n_msg, 0, len(text), text), True, dst=0, tail="text")
        if rank == 0:
            _, n_all, text_all = decode(caps)
            per = got["per_rank"]
            out_q.put((got["merged"] == text_all, sum(p["events"] for p in per) == n_all, n_all >= 3, text_all.count(b'"model"') == n_all))
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_decoded_event_gather():
    from rtl_433_amd import plugins
    if not plugins.available():
        pytest.skip("dropin/_build/libr433plugins.so not built (needs /root/reference once)")
    import torch.multiprocessing as mp
    build_emu.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_decoded_gather, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=900)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(res), res
