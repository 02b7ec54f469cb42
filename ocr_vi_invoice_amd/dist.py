"""Multi-GPU plumbing for the sharded path: one process per GPU, images sharded by rank, and exactly one
collective -- a broadcast of the flattened weights from rank 0 at start-up (RCCL over xGMI on GPUs; gloo in CPU tests)."""
from __future__ import annotations

from typing import List, MutableMapping, Sequence, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of ``total`` items: ranks < total % world get one extra."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def flatten_state_dicts(sds: Sequence[MutableMapping[str, torch.Tensor]]) -> torch.Tensor:
    return torch.cat([v.detach().to(torch.float32).reshape(-1) for sd in sds for v in sd.values()])


def unflatten_into(sds: Sequence[MutableMapping[str, torch.Tensor]], flat: torch.Tensor) -> None:
    off = 0
    flat = flat.detach().cpu()
    for sd in sds:
        for k, v in sd.items():
            n = v.numel()
            sd[k] = flat[off:off + n].reshape(v.shape).to(v.dtype)
            off += n
    assert off == flat.numel()


def broadcast_weights(sds: List[MutableMapping[str, torch.Tensor]], device, dist) -> float:
    """Rank 0's tensors replace every rank's (key order and shapes must agree, which the deterministic schema guarantees).
    Returns the broadcast wall time in ms."""
    import time
    flat = flatten_state_dicts(sds).to(device)
    if dist.get_rank() != 0:
        flat.zero_()
    if flat.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(flat, 0)
    if flat.is_cuda:
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    unflatten_into(sds, flat)
    return ms


def broadcast_blobs(blobs, device, dist):
    """The one collective of the sharded path: rank 0's packed weight blobs (``weights.pack_blob`` output: BN already
    folded, one flat byte string per model) go to every rank in ONE broadcast, so no other rank builds, folds or packs anything.
    ``blobs``: list of bytes on rank 0, ignored elsewhere.  Returns (list of bytes, broadcast wall time in ms)."""
    import time
    rank = dist.get_rank()
    sizes = torch.zeros(8, dtype=torch.int64, device=device)
    if rank == 0:
        assert len(blobs) <= 7
        sizes[0] = len(blobs)
        for i, b in enumerate(blobs):
            sizes[1 + i] = len(b)
    dist.broadcast(sizes, 0)
    n = int(sizes[0].item())
    lens = [int(v) for v in sizes[1:1 + n].tolist()]
    if rank == 0:
        import numpy as np
        flat = torch.from_numpy(np.frombuffer(b"".join(blobs), dtype=np.uint8).copy()).to(device)
    else:
        flat = torch.empty(sum(lens), dtype=torch.uint8, device=device)
    if flat.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(flat, 0)
    if flat.is_cuda:
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    raw = flat.cpu().numpy().tobytes()
    out, off = [], 0
    for ln in lens:
        out.append(raw[off:off + ln])
        off += ln
    return out, ms


def max_over_ranks(seconds: float, device, dist) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, device, dist) -> List[float]:
    """Every rank's value, in rank order, on every rank (the per-rank step times bench.py reports beside their maximum)."""
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]
