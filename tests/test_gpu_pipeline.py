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
