#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: end-to-end invoice images/s (detect -> crop -> recognise).

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by torch.distributed.run, one rank per GPU).
One step = one pass over one batch per rank: `--batch` synthetic 960x1280 invoices (uint8, resident in HBM) ->
normalise -> DBNet++ forward (all five maps, chunks of `--det-chunk`) -> crop+resize+normalise of the ground-truth line
boxes (`--lines` per invoice) -> SVTRv2-base forward in batches of `--rec-batch` -> greedy CTC on device -> ids to host ->
strings.  Images shard across ranks with no data-path collective (weak scaling); RCCL is used once, to broadcast the
packed weights from rank 0.

With random (seeded) weights the probability map carries no text structure, so crop boxes come from the synthetic
generator's ground truth (`boxes=synthetic-gt`); DB post-processing (contours/unclip) is a host
stage listed as the next row in DESIGN.md and is NOT inside the timed region.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline / cpu_baseline definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--workload", default="e2e", choices=["e2e", "det", "rec"])
    ap.add_argument("--batch", type=int, default=64, help="invoices per rank per step (BASELINE.json configs[3])")
    ap.add_argument("--lines", type=int, default=30)
    ap.add_argument("--det-chunk", type=int, default=16)
    ap.add_argument("--rec-batch", type=int, default=256)
    ap.add_argument("--height", type=int, default=960)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernel launches with HIP events")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying captured HIP graphs")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run detector and recogniser back to back on one stream instead of pipelining det(i+1) with rec(i) on two")
    return ap.parse_args()


class Pipeline:
    """The timed region.  Everything here goes through libocrvi's C ABI."""

    def __init__(self, args, dev, det_sd, rec_sd):
        from ocr_vi_invoice_amd import DBNetPP, SVTRv2, _lib
        self.args, self.dev, self.L = args, dev, _lib
        self.lib = _lib.load()
        self.det = DBNetPP(pretrained=False, state_dict=det_sd, dtype=args.dtype, device=dev) if args.workload != "rec" else None
        self.rec = SVTRv2("base", state_dict=rec_sd, dtype=args.dtype, device=dev) if args.workload != "det" else None
        self.devi = torch.device(dev).index or 0

    def load_inputs(self, images_u8, boxes):
        a = self.args
        self.images = torch.from_numpy(images_u8).to(self.dev)                 # [B,H,W,3] uint8, resident in HBM
        self.boxes = torch.from_numpy(boxes).to(self.dev)                      # [B*lines,5] int32
        self.x = torch.empty((a.det_chunk, 3, a.height, a.width), dtype=torch.float32, device=self.dev)
        self.crops = torch.empty((boxes.shape[0], 3, 48, 320), dtype=torch.float32, device=self.dev)

    # ---- the device work of one step, as plain enqueue-only calls (capturable: no allocation, no sync inside libocrvi)
    def _det_chunk(self, i, n):
        a, L, lib = self.args, self.L, self.lib
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        L.check(lib.ocrvi_normalize_u8(self.devi, self.images[i:i + n].data_ptr(), n, a.height, a.width, self.x.data_ptr(), stream))
        return self.det(self.x[:n])                                            # all five maps, as DBNetPP.forward returns

    def _crop(self):
        a, L, lib = self.args, self.L, self.lib
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        L.check(lib.ocrvi_crop_resize_normalize(self.devi, self.images.data_ptr(), a.batch, a.height, a.width, self.boxes.data_ptr(),
                                                self.boxes.shape[0], 48, 320, self.crops.data_ptr(), stream))

    def _rec_chunk(self, i):
        return self.rec._run(self.crops[i:i + self.args.rec_batch], False, True)[2:]   # (ids, lens) on device

    def _device_step(self):
        a = self.args
        out, dec = None, []
        if self.det is not None:
            for i in range(0, a.batch, a.det_chunk):
                out = self._det_chunk(i, min(a.det_chunk, a.batch - i))
        if self.rec is not None:
            self._crop()
            for i in range(0, self.boxes.shape[0], a.rec_batch):
                dec.append(self._rec_chunk(i))
        return out, dec

    def capture(self):
        """Capture one whole step into a HIP graph (hipGraph replay removes ~150 host launches per recogniser forward)."""
        self._device_step()                      # warm-up outside capture: sizes the workspaces, builds lazy state
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.g_out, self.g_dec = self._device_step()

    # ---- two-stream pipeline: det(i+1) runs concurrently with rec(i); rec(i) waits for det(i) by event (as it would have to if its
    #      boxes came from det(i)'s map), ids go to pinned host memory on the rec stream, strings are built one step late.
    def capture_overlap(self):
        a = self.args
        self._device_step()
        torch.cuda.synchronize()
        self.s_det, self.s_rec = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        self.g_det, self.g_rec = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_det, stream=self.s_det):
            for i in range(0, a.batch, a.det_chunk):
                self.g_out = self._det_chunk(i, min(a.det_chunk, a.batch - i))
        with torch.cuda.graph(self.g_rec, stream=self.s_rec):
            self._crop()
            self.g_dec = [self._rec_chunk(i) for i in range(0, self.boxes.shape[0], a.rec_batch)]
            self.g_ids = torch.cat([d[0] for d in self.g_dec])
            self.g_lens = torch.cat([d[1] for d in self.g_dec])
        self.h_ids = [torch.empty(self.g_ids.shape, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.h_lens = [torch.empty(self.g_lens.shape, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.ev_det = [torch.cuda.Event() for _ in range(2)]
        self.ev_rec = [torch.cuda.Event() for _ in range(2)]
        self.pending = None
        self.nstep = 0

    def _collect(self, slot):
        self.ev_rec[slot].synchronize()
        ids, lens = self.h_ids[slot].tolist(), self.h_lens[slot].tolist()
        return self.rec.tokenizer.decode([row[:n] for row, n in zip(ids, lens)])

    def step_overlap(self):
        """Enqueue step i on both streams; returns the strings of step i-1 (None on the first call)."""
        k = self.nstep & 1
        with torch.cuda.stream(self.s_det):
            self.g_det.replay()
            self.ev_det[k].record(self.s_det)
        with torch.cuda.stream(self.s_rec):
            self.s_rec.wait_event(self.ev_det[k])
            self.g_rec.replay()
            self.h_ids[k].copy_(self.g_ids, non_blocking=True)
            self.h_lens[k].copy_(self.g_lens, non_blocking=True)
            self.ev_rec[k].record(self.s_rec)
        texts = self._collect(self.pending) if self.pending is not None else None
        self.pending = k
        self.nstep += 1
        return self.g_out, texts

    def flush_overlap(self):
        texts = self._collect(self.pending) if self.pending is not None else None
        self.pending = None
        return texts

    def step(self):
        if getattr(self, "g_det", None) is not None:
            return self.step_overlap()
        if getattr(self, "graph", None) is not None:
            self.graph.replay()
            out, dec = self.g_out, self.g_dec
        else:
            out, dec = self._device_step()
        texts = None
        if self.rec is not None:                 # ids -> host -> strings (Tokenizer.decode semantics), part of the step
            texts = []
            for ids, lens in dec:
                texts.extend(self.rec._ids_to_text(ids, lens))
        return out, texts


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def cpu_baseline(args, det_sd, rec_sd, image_u8, crops_dev, gpu_texts):
    """Oracle (CPU restatement, `kind: port`) timed on this host on a bounded sample: ONE invoice through the detector and
    `lines` of its crops through the recogniser in the reference's batching (1 image per det forward pipeline2.py:279-317,
    32 crops per rec forward :221).  Also the CER of the GPU strings against the oracle's strings on that sample
    (src/rec2/val.py:14-24 semantics)."""
    from ocr_vi_invoice_amd import synth
    from ocr_vi_invoice_amd.vocab import Tokenizer
    from oracle import dbnet_cpu, svtrv2_cpu
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))   # a 1-GPU box gets a 16-core share of the host; more threads only oversubscribe
    torch.set_num_threads(cores)
    t_det = t_rec = 0.0
    sample = []
    if args.workload != "rec":
        # bounded sample: the top-left quarter (H/2 x W/2, still a multiple of 32) of one invoice; the detector is fully
        # convolutional, so time scales with area -> x4 for one full page
        hh, ww = args.height // 64 * 32, args.width // 64 * 32
        x = torch.from_numpy(synth.normalize_chw(image_u8[:hh, :ww]))[None]
        t0 = time.perf_counter()
        dbnet_cpu.forward(det_sd, x)
        t_det = (time.perf_counter() - t0) * (args.height * args.width) / (hh * ww)
        sample.append(f"detector on a {hh}x{ww} quarter of one invoice, time scaled x{args.height * args.width / (hh * ww):.0f} to a full page")
    cer = None
    if args.workload != "det":
        n = min(args.lines, 8)
        xc = crops_dev[:n].cpu()
        t0 = time.perf_counter()
        ref_txt = []
        for i in range(0, n, 32):
            lp = svtrv2_cpu.forward(rec_sd, xc[i:i + 32], "base")
            ref_txt += Tokenizer().decode(svtrv2_cpu.greedy_ids(lp))
        t_rec = (time.perf_counter() - t0) * args.lines / n
        sample.append(f"{n} crops 48x320 through the recogniser, time scaled to {args.lines} crops/invoice")
        num = sum(edit_distance(g, r) for g, r in zip(gpu_texts[:n], ref_txt))
        cer = num / max(sum(len(r) for r in ref_txt), 1)
    total = t_det + t_rec
    if args.workload == "rec":
        val, unit = args.lines / total, "crops/s"
    else:
        val, unit = 1.0 / total, "images/s"
    return {"value": round(val, 4), "unit": unit, "cores": cores, "kind": "port", "sample": "; ".join(sample),
            "det_s": round(t_det, 2), "rec_s": round(t_rec, 2)}, cer


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # Rehearsal switch for a 1-GPU box: OCRVI_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo for the (CPU-side) collectives.
    rehearse = os.environ.get("OCRVI_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
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

    # ---- weights: built on rank 0, broadcast once over RCCL (the only collective on the path)
    det_sd = weights.make_det_state_dict(seed=1234)
    rec_sd = weights.make_rec_state_dict("base", seed=1234)
    bcast_ms = None
    if world > 1:
        from ocr_vi_invoice_amd.dist import broadcast_weights
        bcast_ms = broadcast_weights([det_sd, rec_sd], cdev, dist)

    # ---- synthetic inputs (this rank's shard)
    imgs, boxes = [], []
    for i in range(args.batch):
        im, bx = synth.make_invoice(1000 * rank + i, args.height, args.width, args.lines)
        imgs.append(im)
        boxes.append(np.concatenate([np.full((len(bx), 1), i, np.int32), bx], 1))
    images_u8 = np.stack(imgs)
    boxes = np.ascontiguousarray(np.concatenate(boxes, 0), dtype=np.int32)

    pipe = Pipeline(args, dev, det_sd, rec_sd)
    pipe.load_inputs(images_u8, boxes)
    lib = _lib.load()

    overlap = not args.no_graph and not args.no_overlap and args.workload == "e2e"
    if overlap:
        pipe.capture_overlap()
    elif not args.no_graph:
        pipe.capture()
    for _ in range(args.warmup):
        pipe.step()
    if overlap:
        pipe.flush_overlap()
    torch.cuda.synchronize()
    graph_mode = getattr(pipe, "graph", None) is not None or overlap
    if not args.no_prof and not graph_mode:      # eager: bracket every launch of the timed region with HIP events
        _lib.check(lib.ocrvi_prof_reset())
        _lib.check(lib.ocrvi_prof_enable(1))
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    texts = None
    for _ in range(args.steps):
        _, t = pipe.step()
        texts = t if t is not None else texts
    if overlap:
        texts = pipe.flush_overlap()     # the last step's strings: still inside the timed region
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        from ocr_vi_invoice_amd.dist import max_over_ranks
        dt = max_over_ranks(dt, cdev, dist)
    prof, prof_steps = {}, args.steps
    if not args.no_prof:
        if graph_mode:
            # graph nodes cannot carry per-kernel events: time the same kernels on the same stream in ONE extra eager step
            _lib.check(lib.ocrvi_prof_reset())
            _lib.check(lib.ocrvi_prof_enable(1))
            pipe._device_step()   # sequential: each kernel alone on the chip, i.e. its intrinsic duration (under the two-stream
                                  # pipeline of the timed region a kernel's duration depends on what it happens to share the chip with)
            torch.cuda.synchronize()
            prof_steps = 1
        _lib.check(lib.ocrvi_prof_enable(0))
        prof = _lib.prof_report()

    if rank == 0:
        units_per_step = args.batch if args.workload != "rec" else boxes.shape[0]
        metric = {"e2e": "invoice images/sec end-to-end (det+rec)", "det": "detection images/sec", "rec": "recognition crops/sec"}[args.workload]
        unit = "crops/s" if args.workload == "rec" else "images/s"
        res = {
            "metric": metric, "value": round(world * units_per_step * args.steps / dt, 3), "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {args.batch} invoices/rank {args.height}x{args.width} -> DBNet++(ResNet-50-DCN, 5 maps) -> "
                                   f"{args.lines} GT-box crops/invoice @48x320 -> SVTRv2-base -> CTC greedy (BASELINE.json configs[3]); "
                                   f"boxes=synthetic-gt, DB post-processing not timed",
                       "global_batch": world * args.batch, "det_chunk": args.det_chunk, "rec_batch": args.rec_batch,
                       "weights": "seeded synthetic (no checkpoint ships)", "launch": "eager" if args.no_graph else ("hipGraph replay, det(i+1) || rec(i) on two streams" if overlap else "hipGraph replay"), "parallelism": f"replicas x{world}, images sharded, no collective"},
        }
        if bcast_ms is not None:
            res["weight_broadcast_ms"] = round(bcast_ms, 2)
        # ---- roofline of the dominant kernel (most accumulated device time over the timed region)
        if prof:
            dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
            name, d = dom
            secs = d["ms"] / 1e3
            if d["flops"] > 0:
                ach, peak, u, bound = d["flops"] / secs / 1e12, PEAK_TFLOPS[args.dtype], "TFLOP/s", "mfma"
            else:
                ach, peak, u, bound = d["bytes"] / secs / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
            traffic = None   # HBM-side bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 runs)
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"].get(name)
                traffic = pmc["traffic_bytes_per_launch"] if pmc else None
            except (OSError, ValueError, KeyError):
                pass
            res["roofline"] = {"kernel": name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": u, "frac": round(ach / peak, 4),
                               "traffic": traffic, "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"], 1),
                               "launches": d["launches"], "avg_ms": round(d["ms"] / d["launches"], 4),
                               "algorithmic_flops_per_launch": round(d["flops"] / d["launches"], 1),
                               "timed_by": "HIP events on the launch stream around every launch, " +
                                           ("one extra sequential eager step after the graph-replayed timed region (rocprof cross-check: profiles/r01_bench_e2e_bf16_no_overlap_kernel_stats.csv)" if graph_mode else "over the timed region")}
            tot = sum(v["ms"] for v in prof.values())
            res["kernel_time_ms_per_step"] = {k: round(v["ms"] / prof_steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
            mf = sum(v["flops"] for v in prof.values())
            res["model_mfma_tflops"] = round(mf / (tot / 1e3) / 1e12, 2)
        if world == 1 and args.workload == "e2e":
            # SURVEY 8(f) row 1 (not in the timed region): the host DB post-processor on a detector-like map of one
            # page (ground-truth line boxes rendered as blobs), through the C ABI, one core
            try:
                from ocr_vi_invoice_amd.pipeline import DBPostProcessor
                pm = np.random.default_rng(0).uniform(0.0, 0.2, (args.height, args.width)).astype(np.float32)
                for _, x, y, w, h in boxes[boxes[:, 0] == 0]:
                    pm[y + 1:y + h - 1, x + 1:x + w - 1] = 0.9
                pp = DBPostProcessor()
                nb = len(pp(pm[None])[0])
                t1 = time.perf_counter()
                for _ in range(10):
                    pp(pm[None])
                res["db_postprocess_host"] = {"ms_per_page": round((time.perf_counter() - t1) / 10 * 1e3, 3), "boxes": nb, "cores": 1,
                                              "in_timed_region": False}
            except Exception as e:  # noqa: BLE001  (a reported extra, never fatal)
                res["db_postprocess_host"] = {"error": str(e)[:120]}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            cb, cer = cpu_baseline(args, det_sd, rec_sd, images_u8[0], pipe.crops if args.workload != "det" else None, texts)
            res["cpu_baseline"] = cb
            if cer is not None:
                res["cer_vs_cpu_ref"] = round(cer, 4)
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
