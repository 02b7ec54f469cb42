"""Multi-GPU plumbing for the sharded path: one process per GPU, images sharded by rank, exactly one collective -- a
broadcast of the packed weight blobs from rank 0 at start-up (RCCL over xGMI on GPUs; gloo in CPU tests) -- and, for unequal pages, a
host-side per-node queue of whole pages (no device collective): a rank that has finished its own shard takes chunks of pages from the
rank that has most left."""
from __future__ import annotations

from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of ``total`` items: ranks < total % world get one extra."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


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


class PageQueue:
    """Per-node queue of whole pages, in chunks (SURVEY.md 8e "host-side work stealing of whole images"; the serial loop it replaces is
    src/pipeline/pipeline2.py:279).  The node's chunks 0 .. n_chunks-1 are split into one contiguous range per rank (``shard_range``:
    the static shard); every range has an atomic head counter in a key-value store that all ranks of the node reach (the process
    group's TCPStore: ``store.add(key, n)`` is an atomic fetch-and-add on the store's host).  ``take()`` hands out the rank's own chunks
    first (locality: they are the pages its host thread prepared) and, once those are gone, chunks from the rank with the most left.
    Every chunk is handed out exactly once, whoever asks; nothing moves between devices -- the caller keeps (or loads) the pages of
    whatever chunk it is given.  One queue object serves one step; ``for_step`` derives the next."""

    def __init__(self, store, name: str, n_chunks: int, rank: int, world: int):
        self.store, self.name, self.n, self.rank, self.world = store, name, int(n_chunks), int(rank), int(world)
        self.ranges = [shard_range(self.n, r, world) for r in range(world)]
        self.taken_own = 0
        self.taken_stolen = 0

    def for_step(self, step) -> "PageQueue":
        return PageQueue(self.store, f"{self.name.split('#')[0]}#{step}", self.n, self.rank, self.world)

    def _key(self, r: int) -> str:
        return f"pagequeue/{self.name}/{r}"

    def _claim(self, r: int) -> Optional[int]:
        lo, hi = self.ranges[r]
        if lo >= hi:
            return None
        i = int(self.store.add(self._key(r), 1)) - 1          # atomic: two ranks never see the same value
        return lo + i if lo + i < hi else None

    def remaining(self, r: int) -> int:
        lo, hi = self.ranges[r]
        return max(0, (hi - lo) - int(self.store.add(self._key(r), 0)))

    def take(self) -> Optional[int]:
        """Next chunk for this rank, or None when the node's queue is empty."""
        c = self._claim(self.rank)
        if c is not None:
            self.taken_own += 1
            return c
        while True:                                            # steal from whoever has most left (re-read after every miss)
            left = sorted(((self.remaining(r), r) for r in range(self.world) if r != self.rank), reverse=True)
            if not left or left[0][0] == 0:
                return None
            c = self._claim(left[0][1])
            if c is not None:
                self.taken_stolen += 1
                return c

    def __iter__(self) -> Iterator[int]:
        while True:
            c = self.take()
            if c is None:
                return
            yield c


def default_store(dist):
    """The key-value store behind the default process group (env:// rendezvous: a TCPStore on MASTER_ADDR:MASTER_PORT)."""
    from torch.distributed import distributed_c10d
    return distributed_c10d._get_default_store()


def drain_queue(queue: PageQueue, launch: Callable[[int], object], wait: Callable[[object], None], depth: int = 2) -> List[int]:
    """Work loop of one rank for one step: take a chunk, ``launch`` it (asynchronous: returns a ticket), and before taking the next one
    ``wait`` for the ticket issued ``depth`` chunks ago -- without this back-pressure a rank whose launches only ENQUEUE device work would
    empty the whole queue in microseconds.  Returns the chunks this rank processed, in order."""
    tickets, mine = [], []
    for c in queue:
        if len(tickets) >= depth:
            wait(tickets.pop(0))
        tickets.append(launch(c))
        mine.append(c)
    for t in tickets:
        wait(t)
    return mine
