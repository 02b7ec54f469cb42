#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: end-to-end invoice images/s (detect -> post-process -> crop -> recognise).

Contract: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment this process starts the N
ranks itself (`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before anything touches the GPU) and relays rank 0's
JSON line; when the driver has already launched the ranks (WORLD_SIZE set) it is simply rank RANK of them.

One step (workload e2e = BASELINE.json configs[3]) = one pass over this rank's batch of `--batch` synthetic 960x1280 invoices that
are resident in HBM as uint8:
  det stream   per chunk of `--det-chunk` pages: normalise -> DBNet++ forward (all five maps) -> probability map -> pinned host memory
  host         DB post-processing of every page (threshold, contours, polygon, score, unclip), boxes -> crop rectangles, on a pool of
               host threads, pipelined one chunk behind the detector (src/pipeline/pipeline2.py:320-343 does this on the host too)
  rec stream   rectangles -> device; crop + resize + normalise from the resident pages -> SVTRv2-base in batches of `--rec-batch` ->
               greedy CTC on device -> ids to pinned host memory -> strings
The recogniser therefore consumes exactly the boxes the post-processor produced from the detector's output of the same step.  With
seeded random weights the detector's map carries no text structure, so each page's map is blended on the device with that page's
synthetic text kernels (ground-truth line boxes shrunk by the DB shrink rule): inside a kernel 0.75 + 0.25 p, elsewhere 0.25 p, p = the
detector's `binary` output.  The whole chain (D2H, contours, unclip, rectangles, H2D, padding of the last recogniser batch) is inside
the timed region.  Images shard across ranks with no data-path collective (weak scaling); RCCL is used once, to broadcast the packed
weight blobs from rank 0.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline / cpu_baseline / parity definitions).
"""
import argparse
import json
import os
import queue
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Dense MFMA peaks (MI355X_MICROARCH.md), in TFLOP/s of the ALGORITHMIC product 2 M N K.  f16x2 forms every product from three 16-bit
# partial products (hi hi, hi lo, lo hi: three v_mfma_f32_16x16x32_f16 per 32 k), so its matrix-pipe roof for
# algorithmic FLOPs is a third of the 16-bit peak; the fp32 MFMA peak (what the exact-fp32 mode is priced against) is quoted beside it.
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3, "f16x2": 2500.0 / 3}
PEAK_HBM_GBS = 8000.0
PROFILE_ROUND = "r04"
PARITY_MODES = ("f16x2", "f32")       # modes whose CTC strings equal the CPU reference's (tests/test_gpu_parity_modes.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16", "f32", "f16x2"],
                    help="operand type of both models for the headline.  Default: f16x2 (fp32-equivalent operands as two fp16 halves on the 16-bit "
                         "matrix pipe: CTC strings equal the CPU reference's), followed by the --also modes; giving --dtype runs that one mode only")
    ap.add_argument("--also", default="f32,f16",
                    help="comma list of further modes measured after the headline in the same process on the same inputs (f32 = exact fp32 MFMA, "
                         "reported as `exact_fp32_mode`; f16 / bf16 = 16-bit throughput mode, reported as `throughput_mode`), or `none`")
    ap.add_argument("--workload", default="e2e", choices=["e2e", "det", "rec"])
    ap.add_argument("--batch", type=int, default=64, help="invoices per rank per step (BASELINE.json configs[3])")
    ap.add_argument("--lines", type=int, default=30)
    ap.add_argument("--det-chunk", type=int, default=16)
    ap.add_argument("--rec-batch", type=int, default=256)
    ap.add_argument("--height", type=int, default=960)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--boxes", default="detected", choices=["detected", "synthetic-gt"],
                    help="detected: crops come from DB post-processing of the (blended) detector map inside the timed region; "
                         "synthetic-gt: round-1 variant, ground-truth rectangles, no post-processing")
    ap.add_argument("--post", choices=["host", "device"], default="host",
                    help="DB post-processing: host = D2H of the whole map, everything on the host (as the reference); device = threshold + "
                         "component labelling + box packing on the GPU, only mask / table / box values cross PCIe, host finishes")
    ap.add_argument("--post-threads", type=int, default=0, help="host threads for DB post-processing (0 = cores available / ranks, at most 16)")
    ap.add_argument("--balance", choices=["static", "queue"], default="static",
                    help="static: every rank walks its own contiguous shard of pages (the contract's weak-scaling run).  queue: every rank keeps "
                         "the whole node's pages resident and takes detector chunks from a per-node host-side queue (ocr_vi_invoice_amd.dist."
                         "PageQueue: own shard first, then from the rank with most left) -- for unequal pages; no device collective either way")
    ap.add_argument("--lines-skew", type=float, default=0.0,
                    help="with N > 1 ranks: text lines per page vary linearly over the ranks' shards from lines*(1-skew) to lines*(1+skew) "
                         "(an artificial imbalance for --balance queue to level)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cer-crops", type=int, default=256,
                    help="crops of the last step whose strings are compared with the CPU oracle's (cer_vs_cpu_ref, strings_differ_vs_cpu)")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the fp32-mode re-run of the last step's crops (CER of the benchmarked dtype)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernel launches with HIP events")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying captured HIP graphs")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one stream, one step at a time: every kernel runs alone on the chip (the variant the rocprofv3 --stats summary under "
                         "profiles/ is taken from, so that its per-kernel averages can be compared with the HIP-event figures)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args):
    """`python bench.py --gpus N` from a plain shell: start N fresh rank processes and relay their output.  Runs before this process
    has imported torch.cuda state or libocrvi, and never re-execs: the children are ordinary subprocesses."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------ helpers
def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def cer(hyp, ref):
    """src/rec2/val.py:14-24: sum of edit distances / sum of reference lengths."""
    num = sum(edit_distance(h, r) for h, r in zip(hyp, ref))
    return num / max(sum(len(r) for r in ref), 1)


def usable_cores(per_rank_cap=16):
    """Host cores this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one, and by the box's
    per-GPU share (16).  os.cpu_count()/affinity alone can name every core of the host while the container is throttled to a few,
    and an oversubscribed thread pool runs the CPU oracle tens of times slower."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, per_rank_cap))


def shrink_box(x, y, w, h, unclip_ratio=1.6):
    """Synthetic text kernel of a line box: the box shrunk by the offset d that the pipeline's unclip (distance = A * ratio / L of the
    kernel, src/det/test.py:37-43 with pipeline2.py's ratio 1.6) maps back onto the line, i.e. the integer d minimising
    |A_k * ratio / L_k - d| for the kernel (w - 2d) x (h - 2d).  (A trained DB head emits such kernels: its labels are the boxes
    shrunk by A (1 - r^2) / L.)"""
    best, best_err = 0, None
    for d in range(0, max(1, (min(w, h) - 3) // 2 + 1)):
        kw, kh = w - 2 * d, h - 2 * d
        err = abs(kw * kh * unclip_ratio / (2.0 * (kw + kh)) - d)
        if best_err is None or err < best_err:
            best, best_err = d, err
    d = best
    return x + d, y + d, w - 2 * d, h - 2 * d


_STREAMS = {}


class E2E:
    """The timed region.  Everything on the device goes through libocrvi's C ABI; the host stage is ocrvi_db_boxes_batch."""

    def __init__(self, args, dev, det_blob, rec_blob, n_ranks_on_host):
        import torch
        from ocr_vi_invoice_amd import DBNetPP, SVTRv2, _lib
        from ocr_vi_invoice_amd.pipeline import DBPostProcessor
        self.torch, self.args, self.dev, self.L = torch, args, dev, _lib
        self.lib = _lib.load()
        self.det = DBNetPP(pretrained=False, blob=det_blob, dtype=args.dtype, device=dev) if args.workload != "rec" else None
        self.rec = SVTRv2("base", blob=rec_blob, dtype=args.dtype, device=dev) if args.workload != "det" else None
        self.devi = torch.device(dev).index or 0
        # pipeline2.py:213-216,254-259 defaults: thresh 0.3, box_thresh 0.5, unclip 1.6, min_area 10
        self.pp = DBPostProcessor(thresh=0.3, box_thresh=0.5, max_candidates=1000, unclip_ratio=1.6)
        # host threads of this rank's post-processing stage: this rank's share of the cores the job may really use (affinity mask cut by
        # the cgroup quota), at most 16 -- eight ranks on one host must not each take sixteen
        self.post_threads = args.post_threads or max(2, min(16, usable_cores(1 << 20) // max(1, n_ranks_on_host)))
        self.detected = args.boxes == "detected" and args.workload == "e2e"
        self.queue = None               # dist.PageQueue in --balance queue mode (set by run_mode)
        self.last_chunks = None

    # ---- inputs
    def load_inputs(self, images_u8, gt_boxes):
        import numpy as np
        torch, a = self.torch, self.args
        self.images = torch.from_numpy(images_u8).to(self.dev)                 # [B,H,W,3] uint8, resident in HBM
        self.gt_rects = np.ascontiguousarray(gt_boxes, dtype=np.int32)          # [B*lines,5] (page, x, y, w, h)
        self.x = torch.empty((a.det_chunk, 3, a.height, a.width), dtype=torch.float32, device=self.dev)
        self.nchunk = (a.batch + a.det_chunk - 1) // a.det_chunk
        if self.detected:
            add = np.zeros((a.batch, 1, a.height, a.width), np.float32)
            for pg, x, y, w, h in self.gt_rects:
                sx, sy, sw, sh = shrink_box(int(x), int(y), int(w), int(h))
                add[pg, 0, sy:sy + sh, sx:sx + sw] = 0.75
            self.kernel_add = torch.from_numpy(add).to(self.dev)
            self.prob = [torch.empty((min(a.det_chunk, a.batch - c * a.det_chunk), 1, a.height, a.width), dtype=torch.float32, device=self.dev)
                         for c in range(self.nchunk)]
            # two steps of pinned host maps in flight
            self.h_prob = [[torch.empty(p.shape, dtype=torch.float32).pin_memory() for p in self.prob] for _ in range(2)]
            self.ev_map = [[torch.cuda.Event() for _ in self.prob] for _ in range(2)]
            self.dcomp = None
            if getattr(a, "post", "host") == "device":
                from ocr_vi_invoice_amd.pipeline import DBComponents
                self.dcomp = [DBComponents(p.shape[0], a.height, a.width, self.dev, cap=4096, pack_frac=0.35, host_slots=2) for p in self.prob]
                self.d2h_bytes = 0
        if self.rec is not None:
            rb = a.rec_batch
            self.d_rects = torch.zeros((rb, 5), dtype=torch.int32, device=self.dev)
            self.crops = torch.empty((rb, 3, 48, 320), dtype=torch.float32, device=self.dev)
            nslots = 2 * ((a.batch * (a.lines + 8) + rb - 1) // rb + self.nchunk + 2)
            self.h_rects = [torch.zeros((rb, 5), dtype=torch.int32).pin_memory() for _ in range(nslots)]
            self.h_ids = [torch.empty((rb, 80), dtype=torch.int32).pin_memory() for _ in range(nslots)]
            self.h_lens = [torch.empty((rb,), dtype=torch.int32).pin_memory() for _ in range(nslots)]
            self.ev_rec = [torch.cuda.Event() for _ in range(nslots)]
            self.slot = 0

    # ---- device work as plain enqueue-only calls (capturable: no allocation, no sync inside libocrvi)
    def _det_chunk(self, c):
        a, L, lib, torch = self.args, self.L, self.lib, self.torch
        i = c * a.det_chunk
        n = min(a.det_chunk, a.batch - i)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        L.check(lib.ocrvi_normalize_u8(self.devi, self.images[i:i + n].data_ptr(), n, a.height, a.width, self.x.data_ptr(), stream))
        out = self.det(self.x[:n])                                             # all five maps, as DBNetPP.forward returns
        if self.detected:                                                      # synthetic text kernels over the random-weight map
            torch.add(self.kernel_add[i:i + n], out["binary"], alpha=0.25, out=self.prob[c])
            if self.dcomp is not None:
                self.dcomp[c].run(self.prob[c], self.pp.thresh)
        return out

    def _rec_batch(self):
        a, L, lib, torch = self.args, self.L, self.lib, self.torch
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        L.check(lib.ocrvi_crop_resize_normalize(self.devi, self.images.data_ptr(), a.batch, a.height, a.width, self.d_rects.data_ptr(),
                                                a.rec_batch, 48, 320, self.crops.data_ptr(), stream))
        return self.rec._run(self.crops, False, True)[2:]                      # (ids, lens) on device

    def capture(self):
        torch, a = self.torch, self.args
        # one pair of streams per process, shared by the modes that run one after the other: every new HIP stream takes the next
        # hardware queue round-robin, and a second pair ends up sharing queues (measured: the second mode lost its det || rec overlap,
        # 134 instead of 110 ms per step)
        if self.dev not in _STREAMS:
            _STREAMS[self.dev] = (torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev))
        self.s_det = _STREAMS[self.dev][0]
        self.s_rec = self.s_det if getattr(a, "no_overlap", False) else _STREAMS[self.dev][1]
        self.g_det, self.g_rec = None, None
        if self.det is not None:
            with torch.cuda.stream(self.s_det):
                for c in range(self.nchunk):
                    self._det_chunk(c)                                         # warm-up outside capture: sizes the workspace
            torch.cuda.synchronize()
            if not a.no_graph:
                self.g_det = []
                for c in range(self.nchunk):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.s_det):
                        self._det_chunk(c)
                    self.g_det.append(g)
        if self.rec is not None:
            with torch.cuda.stream(self.s_rec):
                self.g_ids, self.g_lens = self._rec_batch()
            torch.cuda.synchronize()
            if not a.no_graph:
                self.g_rec = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_rec, stream=self.s_rec):
                    self.g_ids, self.g_lens = self._rec_batch()
        self.jobs = queue.Queue()
        self.room = threading.Semaphore(2)      # the detector may run at most two steps ahead of the post-processor (pinned map ring)
        self.results = queue.Queue()
        self.worker = threading.Thread(target=self._worker, daemon=True)
        self.worker.start()
        self.nstep = 0

    # ---- host side: post-process chunk by chunk, feed the recogniser, collect strings
    # Crops are batched ACROSS steps: a step's last, partial recogniser batch is not padded and launched but carried over and filled
    # with the first crops of the next step (the pages stay resident, so its rectangles remain valid); only a flush -- the end of the
    # timed region -- pads.  Every launched batch remembers which step each of its rows belongs to, and a step's result is emitted
    # once all its rows are decoded, in row order.
    def _feed(self, step, rects):
        """Append step's rectangles to the carry-over and launch every full batch."""
        import numpy as np
        rb = self.args.rec_batch
        ids = np.full(len(rects), step, np.int64)
        self.pend_rects = rects if not len(self.pend_rects) else np.concatenate([self.pend_rects, rects], 0)
        self.pend_steps = ids if not len(self.pend_steps) else np.concatenate([self.pend_steps, ids], 0)
        off = 0
        while len(self.pend_rects) - off >= rb:
            self._launch_rec(self.pend_rects[off:off + rb], self.pend_steps[off:off + rb])
            off += rb
        self.pend_rects, self.pend_steps = self.pend_rects[off:], self.pend_steps[off:]

    def _launch_rec(self, rects, steps):
        torch, a = self.torch, self.args
        nvalid = len(rects)
        k = self.slot
        self.slot = (self.slot + 1) % len(self.h_rects)
        hr = self.h_rects[k].numpy()
        hr[:nvalid] = rects
        if nvalid < a.rec_batch:
            hr[nvalid:] = 0                      # w = h = 0: the all-zero tensor of pipeline2.py:154-156; its string is dropped
        with torch.cuda.stream(self.s_rec):
            self.d_rects.copy_(self.h_rects[k], non_blocking=True)
            if self.g_rec is not None:
                self.g_rec.replay()
            else:
                self.g_ids, self.g_lens = self._rec_batch()
            self.h_ids[k].copy_(self.g_ids, non_blocking=True)
            self.h_lens[k].copy_(self.g_lens, non_blocking=True)
            self.ev_rec[k].record(self.s_rec)
        self.inflight.append((k, nvalid, steps.copy()))

    def _decode_until(self, step):
        """Decode launched batches (oldest first) until every row of `step` has its string.  False if some are not launched yet."""
        rec = self.open_steps[step]
        while len(rec["texts"]) < rec["nrows"]:
            if not self.inflight:
                return False
            k, nvalid, steps = self.inflight.popleft()
            self.ev_rec[k].synchronize()
            ids, lens = self.h_ids[k][:nvalid].tolist(), self.h_lens[k][:nvalid].tolist()
            texts = self.rec.tokenizer.decode([row[:n] for row, n in zip(ids, lens)])
            for st, t in zip(steps.tolist(), texts):
                self.open_steps[st]["texts"].append(t)
        return True

    def _emit_complete(self, upto):
        """Emit, in order, every open step < upto whose rows are all decoded."""
        for st in sorted(self.open_steps):
            if st >= upto:
                break
            if self.rec is not None and not self._decode_until(st):
                break
            rec = self.open_steps.pop(st)
            self.results.put((rec["rects"], rec["texts"] if self.rec is not None else None, rec["counts"]))

    def _worker(self):
        try:
            self._worker_loop()
        except BaseException as e:  # noqa: BLE001  (surface it in the main thread instead of hanging the run)
            self.results.put(e)
            for _ in range(4):
                self.room.release()

    def _worker_loop(self):
        import collections
        import numpy as np
        from ocr_vi_invoice_amd.pipeline import db_boxes_batch
        a = self.args
        self.pend_rects, self.pend_steps = np.zeros((0, 5), np.int32), np.zeros((0,), np.int64)
        self.inflight, self.open_steps = collections.deque(), {}
        while True:
            job = self.jobs.get()
            if job is None:
                break
            if job == "flush":               # end of a timed region: the carried-over partial batch goes out padded, everything is decoded
                if self.rec is not None and len(self.pend_rects):
                    self._launch_rec(self.pend_rects, self.pend_steps)
                    self.pend_rects, self.pend_steps = self.pend_rects[:0], self.pend_steps[:0]
                self._emit_complete(1 << 62)
                assert not self.open_steps and not self.inflight
                self.results.put("flushed")
                continue
            step, chunk_q = job if isinstance(job, tuple) else (job, None)
            all_rects, counts = [], []
            self.open_steps[step] = rec = {"rects": None, "texts": [], "counts": counts, "nrows": 1 << 62}
            if self.detected:
                # static: the step's chunks are all of this rank's; queue mode: whatever the main thread took from the node's queue, as it takes them
                for c in (range(self.nchunk) if chunk_q is None else iter(chunk_q.get, None)):
                    self.ev_map[step & 1][c].synchronize()
                    if self.dcomp is not None:   # (no fallback map is handed over: an overflowing page raises instead of being guessed)
                        rects, cnt, _ = self.dcomp[c].boxes(self.pp, None, 1.0, 1.0, (a.height, a.width), page_base=c * a.det_chunk,
                                                            threads=self.post_threads, slot=step & 1, cap_per_page=256)
                    else:
                        maps = self.h_prob[step & 1][c]
                        rects, cnt, _ = db_boxes_batch(maps.view(maps.shape[0], a.height, a.width), self.pp, 1.0, 1.0, (a.height, a.width),
                                                       page_base=c * a.det_chunk, threads=self.post_threads, cap_per_page=256)
                    all_rects.append(rects)
                    counts.extend(int(v) for v in cnt)
                    if self.rec is not None:
                        self._feed(step, rects)
                self.room.release()
            else:
                rects = self.gt_rects
                all_rects.append(rects)
                if self.rec is not None:
                    with self.torch.cuda.stream(self.s_rec):
                        self.s_rec.wait_event(self.ev_det_done[step & 1])
                    self._feed(step, rects)
                self.room.release()
            rec["rects"] = np.concatenate(all_rects, 0) if all_rects else None
            rec["nrows"] = len(rec["rects"]) if (rec["rects"] is not None and self.rec is not None) else 0
            self._emit_complete(step)            # earlier steps whose tail rode on this step's first batch: their strings while this step's batches run

    def step(self):
        """Enqueue one step: the detector's chunks on its stream; everything downstream is driven by the worker thread."""
        torch, a = self.torch, self.args
        self.room.acquire()
        k = self.nstep & 1
        if self.queue is not None:       # --balance queue: chunks come from the node's page queue (detected-boxes e2e workload only)
            from ocr_vi_invoice_amd.dist import drain_queue
            chunk_q = queue.Queue()
            self.jobs.put((self.nstep, chunk_q))     # the worker post-processes chunk by chunk while later ones are still being taken

            def launch(c):
                with torch.cuda.stream(self.s_det):
                    if self.g_det is not None:
                        self.g_det[c].replay()
                    else:
                        self._det_chunk(c)
                    if self.dcomp is not None:
                        self.d2h_bytes += self.dcomp[c].copy_async(slot=k)
                    else:
                        self.h_prob[k][c].copy_(self.prob[c], non_blocking=True)
                    self.ev_map[k][c].record(self.s_det)
                chunk_q.put(c)
                return c

            q = self.queue.for_step(self.nstep)
            self.last_chunks = drain_queue(q, launch, lambda c: self.ev_map[k][c].synchronize(), depth=2)
            self.stolen = getattr(self, "stolen", 0) + q.taken_stolen
            chunk_q.put(None)
            self.nstep += 1
            return
        if self.det is not None:
            with torch.cuda.stream(self.s_det):
                for c in range(self.nchunk):
                    if self.g_det is not None:
                        self.g_det[c].replay()
                    else:
                        self._det_chunk(c)
                    if self.detected:
                        if self.dcomp is not None:
                            self.d2h_bytes += self.dcomp[c].copy_async(slot=k)
                        else:
                            self.h_prob[k][c].copy_(self.prob[c], non_blocking=True)
                        self.ev_map[k][c].record(self.s_det)
                if not self.detected:
                    if not hasattr(self, "ev_det_done"):
                        self.ev_det_done = [torch.cuda.Event(), torch.cuda.Event()]
                    self.ev_det_done[k].record(self.s_det)
        elif not hasattr(self, "ev_det_done"):
            self.ev_det_done = [torch.cuda.Event(), torch.cuda.Event()]
            for e in self.ev_det_done:
                e.record(self.s_rec)
        self.jobs.put(self.nstep)
        self.nstep += 1

    def finish(self):
        """Drain: returns the list of (rects, strings, boxes-per-page) of every step since the last finish()."""
        self.jobs.put("flush")
        out = []
        while True:
            r = self.results.get()
            if isinstance(r, BaseException):
                raise r
            if isinstance(r, str) and r == "flushed":
                break
            out.append(r)
        self.torch.cuda.synchronize()
        return out

    def close(self):
        self.jobs.put(None)
        self.worker.join(timeout=30)

    # ---- one sequential eager pass of the same device work (per-kernel HIP-event timing; each kernel alone on the chip)
    def eager_pass(self, rects):
        import numpy as np
        torch, a = self.torch, self.args
        if self.det is not None:
            for c in (range(self.nchunk) if self.last_chunks is None else self.last_chunks):
                self._det_chunk(c)
        if self.rec is not None and rects is not None:
            rb = a.rec_batch
            for off in range(0, len(rects), rb):
                part = np.zeros((rb, 5), np.int32)
                n = min(rb, len(rects) - off)
                part[:n] = rects[off:off + n]
                self.d_rects.copy_(torch.from_numpy(part))
                self._rec_batch()
        torch.cuda.synchronize()


def cpu_baseline(args, det_sd, rec_sd, image_u8, crops_f32, gpu_texts, prob_page=None):
    """Oracle (CPU restatement, `kind: port`) timed on this host as SURVEY.md 8d prescribes: ONE full page through the detector and
    32-crop batches through the recogniser (the reference's batching: 1 image per det forward pipeline2.py:279-317, 32 crops per rec
    forward :221), one warm-up then the median of 3, torch.set_num_threads(cores available).  Also the CER of the GPU strings against
    the oracle's strings on those 32 crops (src/rec2/val.py:14-24 semantics)."""
    import numpy as np
    import torch
    from ocr_vi_invoice_amd import synth
    from ocr_vi_invoice_amd.vocab import Tokenizer
    from oracle import dbnet_cpu, svtrv2_cpu
    avail = usable_cores(16)
    torch.set_num_threads(avail)
    t_det = t_rec = t_post = t_pre = 0.0
    sample = []
    if args.workload != "rec":
        x = torch.from_numpy(synth.normalize_chw(image_u8))[None]
        ts = []
        for i in range(4):
            t0 = time.perf_counter()
            dbnet_cpu.forward(det_sd, x)
            ts.append(time.perf_counter() - t0)
        t_det = float(np.median(ts[1:]))
        sample.append(f"detector: one full {args.height}x{args.width} page, 1 warm-up + median of 3")
    if args.workload == "e2e" and prob_page is not None:
        # the host stages between the two models, as the reference runs them per image (pipeline2.py:320-343, src/det/test.py:20-130;
        # pipeline2.py:92-128): DB post-processing of the page's (blended) probability map and crop + resize + normalise of its boxes, by
        # the oracle's restatements of cv2 / pyclipper (single-threaded Python + numpy, like the reference's own loop)
        from oracle import dbpost_cpu, preproc_cpu
        ts, tp = [], []
        for i in range(3):
            t0 = time.perf_counter()
            polys, _ = dbpost_cpu.db_postprocess(prob_page, 0.3, 0.5, 1000, 1.6)
            rects = [dbpost_cpu.rescale_and_rect(q, 1.0, 1.0, args.height, args.width)[1] for q in polys]
            t1 = time.perf_counter()
            for r in rects:
                preproc_cpu.preprocess_for_recognition(preproc_cpu.crop_image(image_u8, r), (48, 320))
            ts.append(t1 - t0)
            tp.append(time.perf_counter() - t1)
        t_post, t_pre = float(np.median(ts)), float(np.median(tp))
        sample.append(f"DB post-processing + crop pre-processing of that page ({len(rects)} boxes): oracle restatements of cv2 / pyclipper, median of 3")
    cer_cpu = None
    if args.workload != "det":
        n = min(32, crops_f32.shape[0])
        xc = crops_f32[:n].cpu()
        ts, ref_txt = [], None
        for i in range(4):
            t0 = time.perf_counter()
            lp = svtrv2_cpu.forward(rec_sd, xc, "base")
            ref_txt = Tokenizer().decode(svtrv2_cpu.greedy_ids(lp))
            ts.append(time.perf_counter() - t0)
        t_rec = float(np.median(ts[1:])) * args.lines / n
        sample.append(f"recogniser: one {n}-crop 48x320 batch, 1 warm-up + median of 3, scaled to {args.lines} crops/invoice")
        # string identity against the oracle on a larger sample than the timing needs: the first `cer_crops` crops of the step, 32 at a
        # time (untimed; ~1.5 s per batch on 16 cores)
        n_cer = min(args.cer_crops, crops_f32.shape[0])
        for i in range(n, n_cer, 32):
            lp = svtrv2_cpu.forward(rec_sd, crops_f32[i:min(i + 32, n_cer)].cpu(), "base")
            ref_txt = ref_txt + Tokenizer().decode(svtrv2_cpu.greedy_ids(lp))
        n_cer = len(ref_txt)
        cer_cpu = {"cer": cer(gpu_texts[:n_cer], ref_txt), "crops": n_cer,
                   "strings_differ": sum(a != b for a, b in zip(gpu_texts[:n_cer], ref_txt))}
    total = t_det + t_post + t_pre + t_rec
    if args.workload == "rec":
        val, unit = args.lines / total, "crops/s"
    else:
        val, unit = 1.0 / total, "images/s"
    return {"value": round(val, 4), "unit": unit, "cores": avail, "kind": "port", "sample": "; ".join(sample),
            "det_s": round(t_det, 3), "post_s": round(t_post, 4), "crop_prep_s": round(t_pre, 4), "rec_s": round(t_rec, 3)}, cer_cpu


def run_mode(args, dtype, dev, cdev, det_blob, rec_blob, images_u8, boxes, local_world, dist, lib):
    """Warm-up + the timed region + one profiled eager pass for one compute dtype.  Returns a dict of raw results."""
    import copy
    import torch
    from ocr_vi_invoice_amd import _lib
    a = copy.copy(args)
    a.dtype = dtype
    queue_mode = args.balance == "queue"
    if queue_mode:
        a.batch = images_u8.shape[0]         # every rank holds the node's pages; the queue decides who runs which chunk
    pipe = E2E(a, dev, det_blob, rec_blob, local_world)
    if queue_mode:
        from ocr_vi_invoice_amd.dist import PageQueue, default_store
        if not pipe.detected:
            raise SystemExit("bench.py: --balance queue needs the e2e workload with --boxes detected")
        world = dist.get_world_size() if dist else 1
        rank = dist.get_rank() if dist else 0

        class _LocalStore:               # one rank: the queue only needs an atomic add
            def __init__(self):
                self.d = {}

            def add(self, k, n):
                self.d[k] = self.d.get(k, 0) + n
                return self.d[k]

        nchunk_node = (a.batch + a.det_chunk - 1) // a.det_chunk
        pipe.queue = PageQueue(default_store(dist) if dist else _LocalStore(), f"bench-{dtype}", nchunk_node, rank, world)
    pipe.load_inputs(images_u8, boxes)
    pipe.capture()
    for _ in range(a.warmup):
        pipe.step()
    pipe.finish()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = []
    for _ in range(a.steps):
        pipe.step()
        if a.no_overlap:
            done.extend(pipe.finish())
    done.extend(pipe.finish())           # the last step's post-processing, recogniser batches and strings: inside the timed region
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    rank_dt = [dt]
    if dist:
        from ocr_vi_invoice_amd.dist import gather_over_ranks, max_over_ranks
        rank_dt = gather_over_ranks(dt, cdev, dist)      # every rank's own time: load imbalance would show here
        dt = max_over_ranks(dt, cdev, dist)
    assert len(done) == a.steps, (len(done), a.steps)
    last_rects, texts, counts = done[-1]
    prof = {}
    if not a.no_prof:
        # graph nodes cannot carry per-kernel events, and under the two-stream pipeline a kernel's duration depends on what shares the
        # chip with it: time the same kernels, shapes and buffers in ONE extra sequential eager pass right after the timed region
        _lib.check(lib.ocrvi_prof_reset())
        _lib.check(lib.ocrvi_prof_enable(1))
        pipe.eager_pass(last_rects)
        _lib.check(lib.ocrvi_prof_enable(0))
        prof = _lib.prof_report()
    out = {"dt": dt, "rank_dt": rank_dt, "rects": last_rects, "texts": texts, "counts": counts, "prof": prof, "detected": pipe.detected,
           "post_threads": pipe.post_threads, "images": pipe.images}
    if queue_mode:
        mine = float(sum(len(r[2]) for r in done))           # pages this rank processed over the timed steps
        stolen = float(getattr(pipe, "stolen", 0))
        if dist:
            from ocr_vi_invoice_amd.dist import gather_over_ranks
            out["pages_by_rank"] = gather_over_ranks(mine, cdev, dist)
            out["chunks_stolen_by_rank"] = gather_over_ranks(stolen, cdev, dist)
        else:
            out["pages_by_rank"], out["chunks_stolen_by_rank"] = [mine], [stolen]
    if pipe.detected:   # bytes the post-processing stage pulls over PCIe per page (whole map, or mask + component table + box values)
        out["d2h_bytes_per_page"] = (int(pipe.d2h_bytes / ((a.steps + a.warmup) * a.batch)) if getattr(pipe, "dcomp", None) is not None
                                     else a.height * a.width * 4)
    pipe.close()
    images = pipe.images
    pipe.__dict__.clear()            # graphs, handles, pinned rings: give everything back before the next mode is set up
    out["images"] = images
    del pipe
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def roofline_of(prof, dtype):
    import re

    def roof(name, d):
        secs = d["ms"] / 1e3
        if d["flops"] > 0:
            ach, peak, u, bound = d["flops"] / secs / 1e12, PEAK_TFLOPS[dtype], "TFLOP/s", "mfma"
            # conv_gemm's 64- and 32-column f16x2 tiles keep the four-product chunk form (a.b + swap(a).b): their matrix-pipe roof is a quarter
            # of the 16-bit peak, not a third (the direct offset conv, 128 x 32, runs three products)
            if dtype == "f16x2" and re.search(r"_128x(64|32)_f16x2$", name) and not name.startswith("dcn_offset"):
                peak = 2500.0 / 4
        else:
            ach, peak, u, bound = d["bytes"] / secs / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
        return {"kernel": name, "bound": bound, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": u, "frac": round(ach / peak, 4)}
    name, d = max(prof.items(), key=lambda kv: kv[1]["ms"])
    traffic = None   # HBM-side bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 runs)
    for rnd in (PROFILE_ROUND, "r01"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")))["kernels"].get(name)
            if pmc:
                traffic = pmc["traffic_bytes_per_launch"]
                break
        except (OSError, ValueError, KeyError):
            pass
    r = roof(name, d)
    if dtype == "f16x2" and r["bound"] == "mfma":
        r.update({"peak": round(r["peak"], 1), "peak_note": "2500 TFLOP/s dense 16-bit MFMA / 3 partial products per product",
                  "executed_16bit_mfma_tflops": round(3 * r["achieved"], 1), "frac_of_fp32_mfma_peak": round(r["achieved"] / 157.3, 4)})
    # the same launches against the OTHER roof (a GEMM on 4-byte f16x2 elements sits close to the ridge: both fractions are of interest)
    r["hbm_frac_algorithmic"] = round(d["bytes"] / (d["ms"] / 1e3) / 1e9 / PEAK_HBM_GBS, 4)
    if traffic:
        r["hbm_frac_traffic"] = round(traffic * d["launches"] / (d["ms"] / 1e3) / 1e9 / PEAK_HBM_GBS, 4)
    r.update({"traffic": traffic, "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"], 1), "launches": d["launches"],
              "avg_ms": round(d["ms"] / d["launches"], 4), "algorithmic_flops_per_launch": round(d["flops"] / d["launches"], 1),
              "timed_by": "HIP events on the launch stream around every launch, one sequential eager pass after the timed region"})
    by = {k: {"frac": roof(k, v)["frac"], "bound": roof(k, v)["bound"], "ms": round(v["ms"], 3)}
          for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    tot = sum(v["ms"] for v in prof.values())
    mf = sum(v["flops"] for v in prof.values())
    return r, by, round(mf / (tot / 1e3) / 1e12, 2)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    if os.environ.get("OCRVI_BENCH_WATCHDOG"):   # development aid: dump every thread's stack if the run takes longer than N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["OCRVI_BENCH_WATCHDOG"]), repeat=True, file=sys.stderr)
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    # The headline is quoted in a mode whose CTC strings equal the CPU reference's: f16x2 (fp32-equivalent operands on the 16-bit matrix
    # pipe; the gate it passed is in DESIGN.md section 4).  The exact-fp32 MFMA mode and a plain 16-bit throughput mode run in the same
    # process on the same inputs and are reported beside it.
    primary = args.dtype or "f16x2"
    others = [] if (args.dtype is not None or args.also == "none") else [m for m in args.also.split(",") if m and m != primary]
    for m in others:
        if m not in PEAK_TFLOPS:
            raise SystemExit(f"bench.py: unknown mode {m!r} in --also")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # Rehearsal switch for a 1-GPU box: OCRVI_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo for the (CPU-side) collectives.
    rehearse = os.environ.get("OCRVI_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    elif torch.cuda.device_count() <= local:      # (device_count() does not initialise the GPU)
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} device(s) are visible: one rank per GPU")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    cdev = "cpu" if rehearse else dev           # device the collectives run on
    from ocr_vi_invoice_amd import _lib, synth, weights
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))  # RCCL over xGMI

    # ---- weights: built, BN-folded and packed on rank 0 only; the packed blobs are broadcast once over RCCL (the only collective on
    #      the path).  Every rank hands the same bytes to ocrvi_*_create.
    det_sd = rec_sd = None
    blobs = None
    if rank == 0:
        det_sd = weights.make_det_state_dict(seed=1234)
        rec_sd = weights.make_rec_state_dict("base", seed=1234)
        blobs = [weights.pack_blob(weights.fold_det(det_sd)), weights.pack_blob(weights.fold_rec(rec_sd, "base"))]
    bcast_ms = None
    if world > 1:
        from ocr_vi_invoice_amd.dist import broadcast_blobs
        blobs, bcast_ms = broadcast_blobs(blobs, cdev, dist)
    det_blob, rec_blob = blobs

    # ---- synthetic inputs (this rank's shard)
    imgs, boxes = [], []
    owners = range(world) if args.balance == "queue" else [rank]      # queue mode: every rank holds the node's pages (64 x 3.7 MB per rank)
    for r in owners:
        lines_r = args.lines
        if world > 1 and args.lines_skew:
            lines_r = max(1, int(round(args.lines * (1.0 + args.lines_skew * (2.0 * r / (world - 1) - 1.0)))))
        for i in range(args.batch):
            im, bx = synth.make_invoice(1000 * r + i, args.height, args.width, lines_r)
            imgs.append(im)
            boxes.append(np.concatenate([np.full((len(bx), 1), len(imgs) - 1, np.int32), bx], 1))
    images_u8 = np.stack(imgs)
    boxes = np.ascontiguousarray(np.concatenate(boxes, 0), dtype=np.int32)
    lib = _lib.load()

    m1 = run_mode(args, primary, dev, cdev, det_blob, rec_blob, images_u8, boxes, local_world, dist, lib)
    extra = [(m, run_mode(args, m, dev, cdev, det_blob, rec_blob, images_u8, boxes, local_world, dist, lib)) for m in others]

    if rank == 0:
        last_rects, texts, counts = m1["rects"], m1["texts"], m1["counts"]
        n_crops = 0 if last_rects is None else len(last_rects)
        units_per_step = args.batch if args.workload != "rec" else n_crops
        metric = {"e2e": "invoice images/sec end-to-end (det+rec)", "det": "detection images/sec", "rec": "recognition crops/sec"}[args.workload]
        unit = "crops/s" if args.workload == "rec" else "images/s"
        boxes_note = ("boxes = DB post-processing (host, in the timed region) of the detector's binary map blended with synthetic text kernels"
                      if m1["detected"] else "boxes=synthetic-gt, DB post-processing not timed")

        def rank_spread(m):      # every rank's own time for the timed region (ms per step): imbalance across ranks shows here
            per = [round(t / args.steps * 1e3, 3) for t in m["rank_dt"]]
            return {"min": min(per), "max": max(per), "per_rank": per}

        res = {
            "metric": metric, "value": round(world * units_per_step * args.steps / m1["dt"], 3), "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(m1["dt"] / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": primary, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {args.batch} invoices/rank {args.height}x{args.width} -> DBNet++(ResNet-50-DCN, 5 maps) -> D2H -> "
                                   f"DB post-process -> crop rects -> H2D -> crop/resize @48x320 -> SVTRv2-base (batches of {args.rec_batch}) -> CTC greedy "
                                   f"(BASELINE.json configs[3]); {boxes_note}",
                       "global_batch": world * args.batch, "det_chunk": args.det_chunk, "rec_batch": args.rec_batch,
                       "crops_per_step_rank0": n_crops, "boxes_per_page_min_max": [min(counts), max(counts)] if counts else None,
                       "weights": "seeded synthetic (no checkpoint ships)",
                       "arithmetic": {"f16x2": "every GEMM operand as two fp16 halves (x = hi + lo), three partial products per product (hi hi + hi lo + lo hi; "
                                               "lo lo <= 2^-22 relative is dropped; narrow conv_gemm tiles compute all four) on v_mfma_f32_16x16x32_f16, "
                                               "fp32 accumulation; fp32 inputs, outputs, biases, residual stream, softmax and LayerNorm (DESIGN.md section 4)",
                                      "f32": "fp32 operands on v_mfma_f32_16x16x4_f32 (an fp32 fmaf chain)",
                                      "f16": "plain fp16 operands, fp32 accumulation", "bf16": "plain bf16 operands, fp32 accumulation"}[primary],
                       "launch": ("eager" if args.no_graph else "hipGraph replay") +
                                 (", one stream, one step at a time (--no-overlap)" if args.no_overlap else
                                  ", det stream || host post-processing || rec stream; recogniser batches filled across steps, one padded batch at the flush"),
                       "post_process": {"in_timed_region": bool(m1["detected"]), "host_threads": m1["post_threads"], "mode": args.post,
                                        "d2h_bytes_per_page": m1.get("d2h_bytes_per_page")},
                       "parallelism": f"replicas x{world}, images sharded, no collective" +
                                      (", per-node page queue (whole pages, host side)" if args.balance == "queue" else "")},
            "ms_per_step_by_rank": rank_spread(m1),
        }
        if args.balance == "queue":
            res["page_queue"] = {"pages_by_rank": m1.get("pages_by_rank"), "chunks_stolen_by_rank": m1.get("chunks_stolen_by_rank"),
                                 "lines_skew": args.lines_skew}
        if bcast_ms is not None:
            res["weight_broadcast_ms"] = round(bcast_ms, 2)
            res["weight_broadcast_bytes"] = len(det_blob) + len(rec_blob)
        if m1["prof"]:
            res["roofline"], res["roofline_by_kernel"], res["model_mfma_tflops"] = roofline_of(m1["prof"], primary)
        # ---- parity of the headline dtype
        crops_all = None
        if args.workload != "det" and last_rects is not None and (not args.no_parity_check or not args.no_cpu_baseline):
            from ocr_vi_invoice_amd.pipeline import preprocess_crops
            crops_all = preprocess_crops(m1["images"], last_rects, (48, 320))
        have_f32 = any(m == "f32" for m, _ in extra)
        if args.workload != "det" and not args.no_parity_check and crops_all is not None and primary != "f32" and not have_f32:
            from ocr_vi_invoice_amd import SVTRv2
            ref = SVTRv2("base", blob=rec_blob, dtype="f32", device=dev)
            t32 = []
            for i in range(0, crops_all.shape[0], args.rec_batch):
                t32 += ref.decode_greedy(crops_all[i:i + args.rec_batch])
            res["cer_vs_f32_mode"] = round(cer(texts, t32), 5)
            res["strings_differ_vs_f32_mode"] = [sum(a != b for a, b in zip(texts, t32)), len(t32)]
            del ref
        for mode, m2 in extra:
            tm = {"dtype": mode, "value": round(world * units_per_step * args.steps / m2["dt"], 3), "unit": unit,
                  "ms_per_step": round(m2["dt"] / args.steps * 1e3, 3), "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step_by_rank": rank_spread(m2),
                  "note": ("same process, same inputs, same timed region as the headline; " +
                           ("fp32 operands on the fp32 MFMA: the arithmetic the reference's CPU path uses" if mode == "f32" else
                            "fp32-equivalent operands as two fp16 halves on the 16-bit matrix pipe (DESIGN.md section 4)" if mode in PARITY_MODES else
                            "NOT string-identical to the CPU reference on the random-weight model (DESIGN.md section 4)"))}
            if m2["prof"]:
                tm["roofline"], tm["roofline_by_kernel"], tm["model_mfma_tflops"] = roofline_of(m2["prof"], mode)
            if args.workload != "det" and m2["texts"] is not None and texts is not None:
                same_rects = m2["rects"] is not None and last_rects is not None and np.array_equal(m2["rects"], last_rects)
                tm["crop_rects_equal_headline"] = bool(same_rects)
                if same_rects:
                    tm["cer_vs_headline_strings"] = round(cer(m2["texts"], texts), 5)
                    tm["strings_differ_vs_headline"] = [sum(x != y for x, y in zip(m2["texts"], texts)), len(texts)]
            res["exact_fp32_mode" if mode == "f32" else ("throughput_mode" if mode not in PARITY_MODES else mode + "_mode")] = tm
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            prob0 = None
            if m1["detected"]:   # page 0's map as the post-processor saw it: the synthetic text kernels over a structureless detector map
                prob0 = np.zeros((args.height, args.width), np.float32)
                for pg, x, y, w, h in boxes:
                    if pg == 0:
                        sx, sy, sw, sh = shrink_box(int(x), int(y), int(w), int(h))
                        prob0[sy:sy + sh, sx:sx + sw] = 0.75
                prob0 += np.float32(0.25) * np.random.default_rng(0).random((args.height, args.width), dtype=np.float32)
            cb, cer_cpu = cpu_baseline(args, det_sd, rec_sd, images_u8[0], crops_all, texts, prob0)
            res["cpu_baseline"] = cb
            if cer_cpu is not None:
                res["cer_vs_cpu_ref"] = round(cer_cpu["cer"], 5)
                res["cer_vs_cpu_ref_crops"] = cer_cpu["crops"]
                res["strings_differ_vs_cpu"] = [cer_cpu["strings_differ"], cer_cpu["crops"]]
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
