"""CPU, gloo, world size 2: the N > 1 plumbing of the sharded path (weight broadcast from rank 0, contiguous image
sharding, max-over-ranks timing).  The data path itself has no collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ocr_vi_invoice_amd import weights
from ocr_vi_invoice_amd.dist import broadcast_weights, flatten_state_dicts, max_over_ranks, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank builds weights from a DIFFERENT seed; after the broadcast all must equal rank 0's
        sds = [weights.make_rec_state_dict("tiny", seed=100 + rank)]
        want = flatten_state_dicts([weights.make_rec_state_dict("tiny", seed=100)])
        ms = broadcast_weights(sds, "cpu", dist)
        same = bool(torch.equal(flatten_state_dicts(sds), want))
        dtype_ok = sds[0]["stem.bn1.num_batches_tracked"].dtype == torch.long
        slow = max_over_ranks(1.0 + rank, "cpu", dist)      # the slowest rank defines the step time
        lo, hi = shard_range(13, rank, world)
        q.put((rank, same, dtype_ok, slow, lo, hi, ms >= 0))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_sharding_and_timing_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "weights differ from rank 0 after broadcast"
    assert all(r[2] for r in res)
    assert [r[3] for r in res] == [2.0, 2.0]
    assert [(r[4], r[5]) for r in res] == [(0, 7), (7, 13)]


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
