"""CPU, gloo, world size 2: the N > 1 plumbing of the sharded path (weight broadcast from rank 0, contiguous image
sharding, max-over-ranks timing).  The data path itself has no collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ocr_vi_invoice_amd import weights
from ocr_vi_invoice_amd.dist import broadcast_blobs, broadcast_weights, flatten_state_dicts, gather_over_ranks, max_over_ranks, shard_range


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
