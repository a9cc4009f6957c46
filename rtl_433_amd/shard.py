"""Multi-GPU batch semantics: independent captures shard contiguously over the ranks (the reference
processes `-r a -r b ...` sequentially with reset_sdr_flow between files, src/rtl_433.c:1703,1854, so
a capture never depends on another) and the only collective is the final variable-length gather of
per-rank result records to rank 0 (RCCL over xGMI on the GPU box, gloo in the CPU tests).

One process per GPU; torch.distributed is plumbing only.
"""
from __future__ import annotations

import numpy as np


def partition(n_items: int, world: int):
    """Contiguous split: rank r gets [bounds[r], bounds[r+1]).  Canonical order is preserved, so
    concatenating per-rank outputs in rank order gives the single-process output."""
    base, extra = divmod(n_items, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


def gather_bytes(local, dst=0, device=None, group=None):
    """Variable-length gather of one byte string per rank to `dst`.
    Returns the list of per-rank bytes on dst, None elsewhere.  Two collectives: an all_gather of the
    lengths and one padded gather of the payloads."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    dev = torch.device("cpu") if device is None else device
    buf = (np.frombuffer(bytes(local), dtype=np.uint8) if not isinstance(local, np.ndarray) else local.view(np.uint8).ravel()).copy()
    n = torch.tensor([buf.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x.item()) for x in sizes]
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if buf.size:
        send[:buf.size] = torch.from_numpy(np.ascontiguousarray(buf)).to(dev)
    recv = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [recv[r][:sizes[r]].cpu().numpy().tobytes() for r in range(world)]


def rebase_packages(blob: bytes, stream_delta: int) -> bytes:
    """Shift the capture index of every package record (include/r433_records.h) by stream_delta."""
    a = np.frombuffer(blob, dtype=np.uint8).copy()
    at = 0
    while at + 64 <= a.size:
        total = int(a[at:at + 4].view(np.uint32)[0])
        a[at + 4:at + 8].view(np.uint32)[0] += np.uint32(stream_delta)
        at += total
    return a.tobytes()


def rebase_events(blob: bytes, pkg_delta: int) -> bytes:
    """Shift the package index of every event record by pkg_delta."""
    a = np.frombuffer(blob, dtype=np.uint8).copy()
    at = 0
    while at + 16 <= a.size:
        total = int(a[at:at + 4].view(np.uint32)[0])
        a[at + 4:at + 8].view(np.uint32)[0] += np.uint32(pkg_delta)
        at += total
    return a.tobytes()


def merge_rank_records(per_rank):
    """per_rank: list over ranks of (first_stream, n_packages, package_blob, event_blob), each numbered
    from 0 inside its rank.  Returns the (package_blob, event_blob) of the whole batch in canonical
    order, identical to what one process produces for the concatenated capture list."""
    pk_all, ev_all, pkg_base = [], [], 0
    for first_stream, n_pkgs, pk, ev in per_rank:
        pk_all.append(rebase_packages(pk, first_stream))
        ev_all.append(rebase_events(ev, pkg_base))
        pkg_base += n_pkgs
    return b"".join(pk_all), b"".join(ev_all)


# ---- the per-rank record that crosses the one collective of the path (bench.py, tests/test_dist_gloo.py) ----

def pack_rank_record(first_stream: int, n_packages: int, n_events: int, digest: int, extra: int, packages: bytes = b"") -> bytes:
    """What a rank sends to rank 0: where its shard starts in the capture list, how many packages / bitbuffers it
    produced, the checksum of the bitbuffers, one caller-defined word, and (optionally) its package records."""
    import struct
    return struct.pack("<QQQQQ", first_stream, n_packages, n_events, digest & 0xFFFFFFFFFFFFFFFF, extra & 0xFFFFFFFFFFFFFFFF) + packages


def unpack_rank_record(blob: bytes) -> dict:
    import struct
    f, npk, nev, dsum, x = struct.unpack_from("<QQQQQ", blob)
    return dict(first=f, packages=npk, events=nev, digest=dsum, extra=x, pk=blob[40:])


def gather_rank_records(payload: bytes, dist_on: bool, dst=0, device=None, tail="packages"):
    """All ranks call this with their packed record; rank `dst` gets the list of unpacked records in rank order, the
    others None.  tail = "packages": the records' tails are package records, merged into the canonical stream of the
    whole list under "merged"; tail = "text": the tails are the ranks' decoded events as JSON lines, which concatenate
    as they are (capture order is rank order)."""
    got = gather_bytes(payload, dst=dst, device=device) if dist_on else (list(payload) if isinstance(payload, (list, tuple)) else [payload])
    if got is None:
        return None
    # (a list of payloads: shards that ran inside one process -- engines on several streams or GPUs -- merge the same way)
    per = [unpack_rank_record(b) for b in got]
    if tail == "text":
        return dict(per_rank=per, merged=b"".join(p["pk"] for p in per))
    merged, _ = merge_rank_records([(p["first"], p["packages"], p["pk"], b"") for p in per])
    return dict(per_rank=per, merged=merged)
