"""GPU: BASELINE.json configs[0] -- one 640x640 page through the whole path (normalise -> DBNet++ -> crop + pre-process the
ground-truth line boxes -> SVTRv2 -> CTC strings), fp32-MFMA parity mode against the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_config0_single_640_page_end_to_end_matches_cpu_oracle():
    from ocr_vi_invoice_amd import DBNetPP, SVTRv2, pipeline, synth, weights
    from ocr_vi_invoice_amd.vocab import Tokenizer
    from oracle import dbnet_cpu, preproc_cpu, svtrv2_cpu
    torch.set_num_threads(16)
    img, boxes = synth.make_invoice(0, 640, 640, lines=12)
    det_sd, rec_sd = weights.make_det_state_dict(seed=1234), weights.make_rec_state_dict("tiny", seed=1234)
    det = DBNetPP(pretrained=False, state_dict=det_sd, dtype="f32")
    rec = SVTRv2("tiny", state_dict=rec_sd, dtype="f32")
    dimg = torch.from_numpy(img).cuda()
    # --- detection: device normalisation == reference arithmetic bit for bit, maps within 1e-3
    x = pipeline.normalize_for_det(dimg)
    xr = torch.from_numpy(preproc_cpu.normalize_det(img))[None]
    assert torch.equal(x.cpu(), xr)
    maps = det(x)
    ref = dbnet_cpu.forward(det_sd, xr)
    for k in ("binary", "thresh", "thresh_binary"):
        assert maps[k].shape == (1, 1, 640, 640)                                     # tests/test_model.py:131-138
        np.testing.assert_allclose(maps[k].cpu().numpy(), ref[k].numpy(), atol=1e-3, err_msg=k)
    # --- recognition on the page's line boxes: crops pre-processed on the device == oracle pre-processing; strings identical
    rects = [(0, int(b[0]), int(b[1]), int(b[2]), int(b[3])) for b in boxes]
    crops = pipeline.preprocess_crops(dimg[None], rects, (32, 256))
    want = np.stack([preproc_cpu.preprocess_for_recognition(img[y:y + h, x0:x0 + w], (32, 256)) for _, x0, y, w, h in rects])
    np.testing.assert_allclose(crops.cpu().numpy(), want, atol=3e-7)
    got = rec.decode_greedy(crops)
    lp = svtrv2_cpu.forward(rec_sd, torch.from_numpy(want), "tiny")
    assert got == Tokenizer().decode(svtrv2_cpu.greedy_ids(lp))
    np.testing.assert_allclose(rec(crops).cpu().numpy(), lp.numpy(), atol=1e-3)
    # --- the reference-named helpers give the same strings crop by crop (pipeline2.py:131-168)
    np_crops = [pipeline.crop_image(img, np.array([[x0, y], [x0 + w - 1, y], [x0 + w - 1, y + h - 1], [x0, y + h - 1]])) for _, x0, y, w, h in rects]
    assert [c.shape[:2] for c in np_crops] == [(h, w) for _, _, _, w, h in rects]
    assert pipeline.recognize_text_batch(rec, np_crops, "cuda:0", (32, 256), batch_size=5) == got
    assert pipeline.recognize_text(rec, np_crops[3], "cuda:0", (32, 256)) == got[3]
    assert pipeline.recognize_text_batch(rec, [np.zeros((0, 7, 3), np.uint8)], "cuda:0") == rec.decode_greedy(torch.zeros(1, 3, 32, 256))


def _render_prob(boxes, H, W, scale_h, scale_w, seed=0):
    """A detector-like probability map for a page: low noise background, one soft blob per line box (in resized coordinates)."""
    rng = np.random.default_rng(seed)
    prob = rng.uniform(0.0, 0.2, (H, W)).astype(np.float32)
    for x, y, w, h in boxes:
        x0, x1 = int(round(x * scale_w)) + 2, int(round((x + w) * scale_w)) - 2
        y0, y1 = int(round(y * scale_h)) + 2, int(round((y + h) * scale_h)) - 2
        prob[y0:y1, x0:x1] = rng.uniform(0.7, 0.99, (y1 - y0, x1 - x0))
    return prob


def test_detect_and_recognize_chain_matches_oracle_chain():
    """pipeline2.py:306-352 as one call.  The detector is a stand-in that returns a rendered map (random weights give no text structure),
    so the chain under test is resize -> [map] -> DBPostProcessor -> rescale -> crop + pre-process on the device -> SVTRv2 -> strings,
    against the oracle chain (oracle/dbpost_cpu + preproc_cpu + svtrv2_cpu) on the same page."""
    from ocr_vi_invoice_amd import DBNetPP, SVTRv2, pipeline, synth, weights
    from ocr_vi_invoice_amd.vocab import Tokenizer
    from oracle import dbpost_cpu, preproc_cpu, svtrv2_cpu
    torch.set_num_threads(16)
    img, gt = synth.make_invoice(3, 1000, 760, lines=10)
    rec_sd = weights.make_rec_state_dict("tiny", seed=77)
    rec = SVTRv2("tiny", state_dict=rec_sd, dtype="f32")
    resized, (scale_h, scale_w) = pipeline.resize_image_for_det(img, 640)
    RH, RW = resized.shape[:2]
    assert (RH, RW) == (640, 480) and np.array_equal(resized.cpu().numpy(), preproc_cpu.resize_linear_u8(img, (RW, RH)))
    prob = _render_prob(gt, RH, RW, scale_h, scale_w)
    seen = {}

    def fake_det(x):
        seen["shape"] = tuple(x.shape)
        return {"binary": torch.from_numpy(prob)[None, None].cuda()}

    pp = pipeline.DBPostProcessor()
    boxes, scores, texts = pipeline.detect_and_recognize(img, fake_det, rec, pp, "cuda:0", det_size=640, rec_size=(32, 256), rec_batch_size=4)
    assert seen["shape"] == (1, 3, RH, RW) and len(boxes) == len(gt) == len(texts) == len(scores)
    # oracle chain on the same page
    oboxes, oscores = dbpost_cpu.db_postprocess(prob[None])
    want_crops = []
    for b, ob in zip(boxes, oboxes):
        rb, (x, y, bw, bh) = dbpost_cpu.rescale_and_rect(ob, scale_w, scale_h, img.shape[0], img.shape[1])
        assert np.array_equal(b, rb)
        want_crops.append(preproc_cpu.preprocess_for_recognition(img[y:y + bh, x:x + bw], (32, 256)))
    np.testing.assert_allclose(scores, oscores, atol=1e-6)
    lp = svtrv2_cpu.forward(rec_sd, torch.from_numpy(np.stack(want_crops)), "tiny")
    assert texts == Tokenizer().decode(svtrv2_cpu.greedy_ids(lp))
    # each detected rectangle covers the line it came from (found bottom-up: cv2 returns the last-found contour first)
    for b, (x, y, w, h) in zip(boxes, gt[::-1]):
        assert b[:, 0].min() <= x + 4 and b[:, 0].max() >= x + w - 4 and b[:, 1].min() <= y + 4 and b[:, 1].max() >= y + h - 4
    # a real (random-weight) detector runs through the same call; whatever boxes its map yields, every box gets a string
    det = DBNetPP(pretrained=False, state_dict=weights.make_det_state_dict(seed=5), dtype="bf16")
    b2, s2, t2 = pipeline.detect_and_recognize(img, det, rec, pipeline.DBPostProcessor(max_candidates=50), "cuda:0", det_size=320)
    assert len(b2) == len(s2) == len(t2)
    # no text at all -> three empty lists (pipeline2.py:332-334)
    assert pipeline.detect_and_recognize(img, lambda x: torch.zeros(1, 1, RH, RW), rec, pp) == ([], [], [])
