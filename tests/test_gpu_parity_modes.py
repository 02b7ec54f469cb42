"""GPU: which compute mode reproduces the reference's CTC strings, at the bench workload's size.

* fp32 MFMA and f16x2 (the parity modes; f16x2 is the mode bench.py quotes its headline in, f32 the facades' default): strings identical
  to the CPU oracle on a full recogniser batch of bench-like crops, and identical to the reference-run goldens.
* f16 / bf16 (throughput modes): the log-prob error is bounded by 1.5x what was measured on MI355X in round 2
  (tools/precision_study.py: f16 0.026, bf16 0.164 over 1920 bench crops), and -- the string-level statement of the same bound -- every
  time step whose fp32 top-2 margin exceeds twice that budget decodes identically.  Full string identity on the bench's crops is NOT
  asserted for them: with seeded random weights 4 % of the time steps have a top-2 margin below 0.05 (smallest seen 3e-6), so no
  evaluation order other than fp32's can reproduce every decision; with a trained checkpoint margins are nats
  wide.  On the four reference-run golden crops every margin is wide enough, and there the strings are asserted in every mode.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BUDGET = {"f16": 0.04, "bf16": 0.25}      # 1.5 x the measured max |dlog-prob| over 1920 bench crops (0.0258 / 0.1639)
MIN_AGREE = {"f16": 0.997, "bf16": 0.985}  # per-step argmax agreement with the fp32 mode (measured 0.9983 / 0.9891)


def _bench_crops(n_pages, lines=30, seed0=0):
    from ocr_vi_invoice_amd import synth
    from ocr_vi_invoice_amd.pipeline import preprocess_crops
    imgs, rects = [], []
    for i in range(n_pages):
        im, bx = synth.make_invoice(seed0 + i, 960, 1280, lines)
        imgs.append(im)
        rects += [(i, int(x), int(y), int(w), int(h)) for x, y, w, h in bx]
    pages = torch.from_numpy(np.stack(imgs)).cuda()
    return preprocess_crops(pages, rects, (48, 320))


@pytest.mark.parametrize("dt", ["f32", "f16x2"])
def test_f32_mode_strings_equal_cpu_oracle_on_bench_crops(dt):
    """128 of the bench's own crops (4 pages x 32 lines): parity-mode strings == CPU-oracle strings, log-probs within 1e-3 (and within
    1e-4, the bar round 2's verdict set for calling f16x2 an fp32-equivalent mode; measured 3.6e-5 / 3.1e-5 over 1920 crops)."""
    from ocr_vi_invoice_amd import SVTRv2, weights
    from ocr_vi_invoice_amd.vocab import Tokenizer
    from oracle import svtrv2_cpu
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    sd = weights.make_rec_state_dict("base", seed=1234)
    crops = _bench_crops(4, lines=32)
    m = SVTRv2("base", state_dict=sd, dtype=dt)
    lp = m(crops)
    got = m.decode_probs(lp)
    xc = crops.cpu()
    want, worst = [], 0.0
    for i in range(0, xc.shape[0], 32):
        ref = svtrv2_cpu.forward(sd, xc[i:i + 32], "base")
        want += Tokenizer().decode(svtrv2_cpu.greedy_ids(ref))
        worst = max(worst, float((lp[:, i:i + 32].cpu() - ref).abs().max()))
    assert worst < 1e-4, worst
    assert got == want
    assert m.decode_greedy(crops) == want


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_lowp_error_budget_and_decidable_steps_on_a_full_batch(dt):
    from ocr_vi_invoice_amd import SVTRv2
    crops = _bench_crops(8, lines=32)                      # 256 crops: one full recogniser batch of the bench
    ref = SVTRv2("base", dtype="f32", seed=1234)
    lp32 = ref(crops)
    m = SVTRv2("base", dtype=dt, seed=1234)
    lp = m(crops)
    err = float((lp - lp32).abs().max())
    top2 = lp32.topk(2, -1).values
    margin = top2[..., 0] - top2[..., 1]                   # [T, B]
    flips = lp.argmax(-1) != lp32.argmax(-1)
    agree = 1.0 - float(flips.float().mean())
    print(f"\n[{dt}] 256 bench crops: max |dlog-prob| {err:.4f} (budget {BUDGET[dt]}), argmax agreement {agree:.5f}, "
          f"flipped steps {int(flips.sum())}, largest fp32 margin at a flip {float(margin[flips].max()) if flips.any() else 0.0:.4f}, "
          f"steps with margin < 2*budget: {int((margin < 2 * BUDGET[dt]).sum())} of {margin.numel()}")
    assert err < BUDGET[dt]
    assert agree > MIN_AGREE[dt]
    decidable = margin > 2 * BUDGET[dt]
    assert not bool((flips & decidable).any())             # every decision outside the precision band is reproduced
    # crops all of whose steps are decidable must decode to the fp32 mode's strings
    t32, t = ref.decode_probs(lp32), m.decode_probs(lp)
    clean = decidable.all(0).cpu().tolist()
    assert all(a == b for a, b, c in zip(t, t32, clean) if c)


@pytest.mark.parametrize("dt", ["f32", "f16x2", "f16", "bf16"])   # (the 16-bit modes too: every step of these four golden crops has a top-2 margin above their error)
@pytest.mark.parametrize("name", ["rec_base_48x320", "rec_tiny_32x256"])
def test_strings_equal_reference_goldens(golden_dir, name, dt):
    from ocr_vi_invoice_amd import SVTRv2, weights
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    variant = str(g["variant"])
    m = SVTRv2(variant, state_dict=weights.make_rec_state_dict(variant, seed=int(g["seed"])), dtype=dt)
    assert m.decode_greedy(torch.from_numpy(g["x"]).cuda()) == [str(s) for s in g["strings"]]


@pytest.mark.parametrize("post,dt", [("host", "f32"), ("device", "f32"), ("host", "f16x2")])   # whole map to the host / device threshold + components, host finish
def test_bench_pipeline_small_f32_boxes_and_strings_match_the_oracle_chain(post, dt):
    """configs[3] at reduced size through bench.py's own E2E class (fp32 mode): 6 pages 320x480, 5 lines each, detector -> blended map
    -> D2H -> ocrvi_db_boxes_batch -> crop + SVTRv2-base -> strings, pipelined on two streams + a host thread, against the oracle chain
    on the same inputs (oracle detector -> same blend -> oracle post-processing -> rects -> oracle pre-processing -> oracle recogniser);
    and the strings of three pipelined steps (one-step-late hand-off) equal each other."""
    import argparse
    import importlib.util
    from ocr_vi_invoice_amd import synth, weights
    from ocr_vi_invoice_amd.vocab import Tokenizer
    from oracle import dbnet_cpu, dbpost_cpu, preproc_cpu, svtrv2_cpu
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_e2e", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    H, W, B, lines = 320, 480, 6, 5
    args = argparse.Namespace(dtype=dt, workload="e2e", batch=B, lines=lines, det_chunk=4, rec_batch=16, height=H, width=W,
                              boxes="detected", post_threads=3, no_graph=False, post=post)
    det_sd, rec_sd = weights.make_det_state_dict(seed=1234), weights.make_rec_state_dict("base", seed=1234)
    imgs, gts = [], []
    for i in range(B):
        im, bx = synth.make_invoice(50 + i, H, W, lines)
        imgs.append(im)
        gts.append(np.concatenate([np.full((len(bx), 1), i, np.int32), bx], 1))
    imgs, gts = np.stack(imgs), np.concatenate(gts, 0).astype(np.int32)
    pipe = bench.E2E(args, "cuda:0", weights.pack_blob(weights.fold_det(det_sd)), weights.pack_blob(weights.fold_rec(rec_sd, "base")), 1)
    pipe.load_inputs(imgs, gts)
    pipe.capture()
    for _ in range(3):
        pipe.step()
    done = pipe.finish()
    pipe.close()
    assert len(done) == 3
    rects, texts, counts = done[0]
    for r2, t2, c2 in done[1:]:                                   # pipelined steps reproduce each other exactly
        assert np.array_equal(r2, rects) and t2 == texts and c2 == counts
    assert sum(counts) == len(rects) == len(texts) and min(counts) >= 1
    # ---- oracle chain
    want_rects, want_crops = [], []
    for i in range(B):
        x = torch.from_numpy(synth.normalize_chw(imgs[i]))[None]
        p = dbnet_cpu.forward(det_sd, x)["binary"][0, 0].numpy()
        add = np.zeros((H, W), np.float32)
        for pg, bx, by, bw, bh in gts[gts[:, 0] == i]:
            sx, sy, sw, sh = bench.shrink_box(int(bx), int(by), int(bw), int(bh))
            add[sy:sy + sh, sx:sx + sw] = 0.75
        prob = (add + np.float32(0.25) * p).astype(np.float32)
        oboxes, _ = dbpost_cpu.db_postprocess(prob[None], thresh=0.3, box_thresh=0.5, unclip_ratio=1.6)
        for ob in oboxes:
            _, (rx, ry, rw, rh) = dbpost_cpu.rescale_and_rect(ob, 1.0, 1.0, H, W)
            want_rects.append((i, rx, ry, rw, rh))
            want_crops.append(preproc_cpu.preprocess_for_recognition(imgs[i][ry:ry + rh, rx:rx + rw], (48, 320)))
    assert np.array_equal(rects, np.asarray(want_rects, np.int32))
    lp = svtrv2_cpu.forward(rec_sd, torch.from_numpy(np.stack(want_crops)), "base")
    assert texts == Tokenizer().decode(svtrv2_cpu.greedy_ids(lp))
    # every ground-truth line was found (the unclipped kernel covers most of its line)
    assert len(rects) == B * lines


def test_bench_launcher_runs_two_ranks_from_a_plain_invocation():
    """`python bench.py --gpus 2` (no torchrun around it) must start two ranks and report n_gpus: 2.  On this one-GPU box both ranks
    share cuda:0 and the broadcast runs over gloo (OCRVI_BENCH_REHEARSE=1); on a multi-GPU node the same path uses RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OCRVI_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "4", "--det-chunk", "2",
           "--lines", "4", "--height", "320", "--width", "480", "--rec-batch", "16", "--no-cpu-baseline", "--no-prof", "--also", "none", "--dtype", "f16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 8 and res["value"] > 0
    assert res["weight_broadcast_ms"] >= 0 and res["weight_broadcast_bytes"] > 50e6
    assert res["config"]["post_process"]["in_timed_region"] is True


def test_bench_page_queue_two_ranks_process_every_page_once():
    """`bench.py --balance queue` (SURVEY.md 8e work stealing of whole pages; pipeline2.py:279 is the serial loop): two ranks on this box's
    one GPU (gloo rehearsal), rank 1's pages carrying more lines than rank 0's.  Every page of the node is processed exactly once per
    step whoever takes it, and the strings are unaffected by who ran which chunk (same pages, same strings as the static run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OCRVI_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--det-chunk", "2",
            "--lines", "6", "--lines-skew", "0.5", "--height", "320", "--width", "480", "--rec-batch", "16", "--no-cpu-baseline", "--no-prof",
            "--also", "none", "--dtype", "f16"]
    out = {}
    for mode in ("static", "queue"):
        r = subprocess.run(base + ["--balance", mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    q = out["queue"]
    assert q["n_gpus"] == 2 and q["config"]["global_batch"] == 16 and q["value"] > 0
    assert sum(q["page_queue"]["pages_by_rank"]) == 2 * 16                 # 2 timed steps x 16 pages: none dropped, none twice
    assert "page queue" in q["config"]["parallelism"] and "page_queue" not in out["static"]
