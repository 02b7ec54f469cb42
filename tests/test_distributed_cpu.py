"""CPU, gloo, world size 2: the N > 1 plumbing of the sharded path (packed-blob broadcast from rank 0, contiguous image sharding,
max-over-ranks timing, the per-node page queue for unequal pages).  The data path itself has no collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ocr_vi_invoice_amd import weights
from ocr_vi_invoice_amd.dist import PageQueue, broadcast_blobs, default_store, drain_queue, gather_over_ranks, max_over_ranks, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        same = dtype_ok = True
        ms = 0.0
        slow = max_over_ranks(1.0 + rank, "cpu", dist)      # the slowest rank defines the step time
        per_rank = gather_over_ranks(1.0 + rank, "cpu", dist)   # ... and every rank's own time is reported beside it (bench.py: ms_per_step_by_rank)
        lo, hi = shard_range(13, rank, world)
        # the designed collective: rank 0 folds + packs, everybody receives the same bytes; other ranks build nothing
        mine = None
        if rank == 0:
            mine = [weights.pack_blob(weights.fold_rec(weights.make_rec_state_dict("tiny", seed=100), "tiny")), b"second-blob"]
        got, bms = broadcast_blobs(mine, "cpu", dist)
        want_blob = weights.pack_blob(weights.fold_rec(weights.make_rec_state_dict("tiny", seed=100), "tiny"))
        blob_ok = got[0] == want_blob and got[1] == b"second-blob" and bms >= 0
        q.put((rank, same, dtype_ok, slow, lo, hi, ms >= 0 and blob_ok and per_rank == [1.0, 2.0]))
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
    assert all(r[6] for r in res), "packed-blob broadcast differs from rank 0's bytes"


def test_bench_launcher_starts_n_ranks(monkeypatch):
    """`python bench.py --gpus N` from a plain shell must start N ranks itself (torch.distributed.run, 127.0.0.1 rendezvous) and
    never fall through to a single-rank run; inside a rank a --gpus / WORLD_SIZE mismatch is fatal."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
        raise AssertionError("launcher must exit with the children's status")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # inside a rank: WORLD_SIZE must equal --gpus
    monkeypatch.setenv("WORLD_SIZE", "2")
    try:
        bench.main()
        raise AssertionError("mismatch must be fatal")
    except SystemExit as e:
        assert "WORLD_SIZE=2" in str(e.code)


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- per-node page queue (host-side work stealing of whole images; replaces the serial loop of pipeline2.py:279)
def _queue_worker(rank, world, port, q, lines, chunk_pages, balanced):
    """One step over the node's pages: page j costs `lines[j]` ms of (simulated) device time.  Static mode: every rank walks its own
    shard.  Queue mode: chunks come from the PageQueue.  Reports (rank, busy seconds, pages processed)."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_chunks = len(lines) // chunk_pages
        cost = lambda c: sum(lines[c * chunk_pages:(c + 1) * chunk_pages]) * 1e-3
        done_at = {}

        def launch(c):               # "enqueue" the chunk: the device is busy until its finish time
            start = max(time.perf_counter(), max(done_at.values(), default=0.0))
            done_at[c] = start + cost(c)
            return c

        def wait(c):
            time.sleep(max(0.0, done_at[c] - time.perf_counter()))

        dist.barrier()
        t0 = time.perf_counter()
        if balanced:
            pq = PageQueue(default_store(dist), "cpu-test", n_chunks, rank, world).for_step(0)
            mine = drain_queue(pq, launch, wait, depth=2)
            stolen = pq.taken_stolen
        else:
            lo, hi = shard_range(n_chunks, rank, world)
            mine = [launch(c) for c in range(lo, hi)]
            for c in mine:
                wait(c)
            stolen = 0
        busy = time.perf_counter() - t0
        step = max_over_ranks(busy, "cpu", dist)
        per_rank = gather_over_ranks(busy, "cpu", dist)
        pages = [p for c in mine for p in range(c * chunk_pages, (c + 1) * chunk_pages)]
        q.put((rank, busy, step, per_rank, pages, stolen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_queue_case(balanced, lines, chunk_pages=2):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_queue_worker, args=(r, world, port, q, lines, chunk_pages, balanced)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_page_queue_balances_unequal_pages_world2():
    """32 pages in chunks of 2; rank 0's shard holds short pages (10 lines -> 10 ms each), rank 1's long ones (40 lines).  Static
    sharding leaves rank 0 idle for most of the step; with the queue it takes chunks from rank 1.  Every page is processed exactly
    once either way, the step time (max over ranks) drops and the spread between the ranks' own times shrinks."""
    lines = [10] * 16 + [40] * 16
    static = _run_queue_case(False, lines)
    queued = _run_queue_case(True, lines)
    for res in (static, queued):
        pages = sorted(p for r in res for p in r[4])
        assert pages == list(range(32)), "every page exactly once"
    s_busy, q_busy = [r[1] for r in static], [r[1] for r in queued]
    s_step, q_step = static[0][2], queued[0][2]
    assert static[0][3] == static[1][3] and len(static[0][3]) == 2          # every rank sees every rank's time
    assert s_busy[1] > 3.0 * s_busy[0]                                       # static: 0.16 s vs 0.64 s
    assert q_step < 0.75 * s_step, (q_step, s_step)                          # ideal 0.40 s against 0.64 s
    assert max(q_busy) - min(q_busy) < 0.35 * (max(s_busy) - min(s_busy)), (q_busy, s_busy)
    assert queued[0][5] > 0 and queued[1][5] == 0                            # rank 0 stole, rank 1 never had to
    assert len(queued[0][4]) > len(queued[1][4])


def test_page_queue_hands_out_every_chunk_once_without_a_process_group():
    """The queue only needs an atomic add: exercised here against a dict-backed stand-in, ranks interleaved by hand, odd sizes."""
    class Store:
        def __init__(self):
            self.d = {}

        def add(self, k, n):
            self.d[k] = self.d.get(k, 0) + n
            return self.d[k]

    for n_chunks, world in ((7, 3), (1, 4), (0, 2), (16, 2), (5, 8)):
        st = Store()
        qs = [PageQueue(st, "t", n_chunks, r, world).for_step(3) for r in range(world)]
        got, alive, turn = [], list(range(world)), 0
        while alive:
            r = alive[turn % len(alive)]
            # (rank 0 is "fast": it asks three times per round)
            for _ in range(3 if r == 0 else 1):
                c = qs[r].take()
                if c is None:
                    alive.remove(r)
                    break
                got.append(c)
            turn += 1
        assert sorted(got) == list(range(n_chunks)), (n_chunks, world, got)
