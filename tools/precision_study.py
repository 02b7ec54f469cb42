#!/usr/bin/env python3
"""Which compute dtype keeps the CTC strings of the bench workload identical to the fp32 parity mode?

Runs the recogniser over the bench's own crops (GT boxes of `--pages` synthetic invoices, cut by the device crop kernel) in
f32 / f16 / bf16 and prints, per dtype: strings that differ from f32's, per-step argmax agreement, max |dlog-prob|, and the
fp32 top-2 margin statistics (how close the random-weight model's decisions are).  Development tool; results are quoted in
DESIGN.md."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=64)
    ap.add_argument("--lines", type=int, default=30)
    ap.add_argument("--dtypes", default="f16,bf16")
    ap.add_argument("--det-pages", type=int, default=2)
    ap.add_argument("--cpu-det-pages", type=int, default=0, help="also compare the fp32 detector with the CPU oracle on N full-size pages")
    ap.add_argument("--cpu-crops", type=int, default=0, help="also run the CPU oracle (fp32) on the first N crops and compare the f32-mode strings")
    args = ap.parse_args()
    from ocr_vi_invoice_amd import DBNetPP, SVTRv2, synth, weights
    from ocr_vi_invoice_amd.pipeline import preprocess_crops
    dev = "cuda:0"
    imgs, rects = [], []
    for i in range(args.pages):
        im, bx = synth.make_invoice(i, 960, 1280, args.lines)
        imgs.append(im)
        rects += [(i, int(x), int(y), int(w), int(h)) for x, y, w, h in bx]
    pages = torch.from_numpy(np.stack(imgs)).to(dev)
    crops = preprocess_crops(pages, rects, (48, 320))
    sd = weights.make_rec_state_dict("base", seed=1234)
    out = {}

    def run(dt):
        m = SVTRv2("base", state_dict=sd, dtype=dt, device=dev)
        lps, txt = [], []
        for i in range(0, crops.shape[0], 256):
            lp = m(crops[i:i + 256])
            lps.append(lp.permute(1, 0, 2).contiguous().cpu())
            txt += m.decode_probs(lp)
        return torch.cat(lps), txt

    lp32, t32 = run("f32")
    top2 = lp32.topk(2, -1).values
    margin = (top2[..., 0] - top2[..., 1])
    out["f32"] = {"crops": len(t32), "min_top2_margin": float(margin.min()), "margin_p1": float(margin.flatten().kthvalue(max(1, margin.numel() // 100)).values),
                  "steps_with_margin_lt_0.05": int((margin < 0.05).sum()), "steps": int(margin.numel())}
    am32 = lp32.argmax(-1)
    ref_lp = ref_txt = None

    def vs_cpu(lp, txt):
        """one mode against the CPU oracle on the first n crops: strings, argmax decisions, max |dlog-prob|, mean |d(top-2 margin)|"""
        n = ref_lp.shape[0]
        bad = [i for i in range(n) if ref_txt[i] != txt[i]]
        rm = ref_lp.topk(2, -1)
        rmargin = rm.values[..., 0] - rm.values[..., 1]
        flips = ref_lp.argmax(-1) != lp[:n].argmax(-1)
        gm = lp[:n].gather(-1, rm.indices)           # this mode's log-probs at the oracle's top-2 classes
        return {"crops": n, "strings_differ": len(bad), "steps_argmax_differ": int(flips.sum()),
                "max_abs_err": float((ref_lp - lp[:n]).abs().max()), "mean_abs_err": float((ref_lp - lp[:n]).abs().mean()),
                "mean_abs_top2_margin_err": float(((gm[..., 0] - gm[..., 1]) - rmargin).abs().mean()),
                "largest_ref_margin_at_a_flip": float(rmargin[flips].max()) if flips.any() else 0.0}

    if args.cpu_crops:
        from oracle import svtrv2_cpu
        from ocr_vi_invoice_amd.vocab import Tokenizer
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        n = min(args.cpu_crops, crops.shape[0])
        xc = crops[:n].cpu()
        ref_lp, ref_txt = [], []
        for i in range(0, n, 32):
            lp = svtrv2_cpu.forward(sd, xc[i:i + 32], "base")
            ref_lp.append(lp.permute(1, 0, 2))
            ref_txt += Tokenizer().decode(svtrv2_cpu.greedy_ids(lp))
        ref_lp = torch.cat(ref_lp)
        out["f32_gpu_vs_cpu_oracle"] = vs_cpu(lp32, t32)
    for dt in args.dtypes.split(","):
        lp, t = run(dt)
        bad = [i for i, (a, b) in enumerate(zip(t, t32)) if a != b]
        flips = lp.argmax(-1) != am32
        out[dt] = {"strings_differ": len(bad), "argmax_agree": float((~flips).float().mean()),
                   "max_abs_err": float((lp - lp32).abs().max()), "steps_argmax_differ": int(flips.sum()),
                   "largest_f32_margin_at_a_flip": float(margin[flips].max()) if flips.any() else 0.0}
        if ref_lp is not None:
            out[dt + "_gpu_vs_cpu_oracle"] = vs_cpu(lp, t)
    # detector: binary-map error of the low-precision modes on full-size pages
    dsd = weights.make_det_state_dict(seed=1234)
    from ocr_vi_invoice_amd.pipeline import normalize_for_det
    x = normalize_for_det(pages[:args.det_pages])
    d32 = DBNetPP(pretrained=False, state_dict=dsd, dtype="f32", device=dev)(x)["binary"].cpu()
    out["det_f32_range"] = [float(d32.min()), float(d32.max())]
    for dt in args.dtypes.split(","):
        b = DBNetPP(pretrained=False, state_dict=dsd, dtype=dt, device=dev)(x)["binary"].cpu()
        out["det_" + dt] = {"max_abs_err": float((b - d32).abs().max()), "mean_abs_err": float((b - d32).abs().mean()),
                            "thr0.3_flips": int(((b > 0.3) != (d32 > 0.3)).sum())}
    if args.cpu_det_pages:
        # fp32 mode vs the CPU oracle at FULL size (the -m gpu model tests do this at 64x96 .. 160x96): probability maps and the crop
        # rectangles post-processing derives from the blended maps (bench.py's chain)
        from oracle import dbnet_cpu
        from ocr_vi_invoice_amd.pipeline import DBPostProcessor, db_boxes_batch
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        n = args.cpu_det_pages
        refs = [dbnet_cpu.forward(dsd, x[i:i + 1].cpu()) for i in range(n)]
        pp = DBPostProcessor(0.3, 0.5, 1000, 1.6)
        for dt in ["f32"] + [d for d in args.dtypes.split(",") if d != "f32"]:
            gpu = DBNetPP(pretrained=False, state_dict=dsd, dtype=dt, device=dev)(x[:n])
            worst, rect_equal = {}, True
            for i in range(n):
                ref = refs[i]
                for k in ("binary", "thresh", "thresh_binary"):
                    worst[k] = max(worst.get(k, 0.0), float((gpu[k][i:i + 1].cpu() - ref[k]).abs().max()))
                add = np.zeros((960, 1280), np.float32)
                for r in rects:
                    if r[0] == i:
                        sx, sy, sw, sh = bench.shrink_box(*r[1:])
                        add[sy:sy + sh, sx:sx + sw] = 0.75
                ma = (add + np.float32(0.25) * gpu["binary"][i, 0].cpu().numpy()).astype(np.float32)
                mb = (add + np.float32(0.25) * ref["binary"][0, 0].numpy()).astype(np.float32)
                ra, _, _ = db_boxes_batch(ma[None], pp)
                rb, _, _ = db_boxes_batch(mb[None], pp)
                rect_equal = rect_equal and np.array_equal(ra, rb)
            out[f"det_{dt}_gpu_vs_cpu_oracle_fullsize"] = {"pages": n, "max_abs_err": worst, "crop_rects_equal": bool(rect_equal)}
            del gpu
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
