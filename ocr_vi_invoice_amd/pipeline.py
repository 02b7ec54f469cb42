"""Host-side mirror of the reference pipeline's helpers (src/pipeline/pipeline2.py) on top of libocrvi.

Same names and argument meaning as the reference for the pieces either side of the two models:
``load_detection_model`` (:43), ``load_recognition_model`` (:72), ``preprocess_for_recognition`` (:92), ``recognize_text`` (:131),
``recognize_text_batch`` (:144), ``resize_image_for_det`` (:33), ``crop_image`` (src/det/test.py:123) plus ``normalize_for_det`` (the
inline code at :312-314), ``rescale_boxes`` (:324-328) and ``detect_and_recognize`` (steps 2-3 of the per-image loop, :306-352).
Image resizing and crop pre-processing run on the GPU (ocrvi_crop_resize_normalize / ocrvi_normalize_u8).

``DBPostProcessor`` (src/det/test.py:46-106) is the host C++ implementation behind ``ocrvi_db_postprocess``; like the reference's, it works on
the probability map after the device->host copy.  cv2 / pyclipper / shapely are absent from the build container, so its agreement with
those libraries is unpinned; it is checked bit-for-bit against the independent statement in oracle/dbpost_cpu.py.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import ctypes

import numpy as np
import torch

from . import _lib
from .det import DBNetPP
from .rec import SVTRv2


def _numpy_data_globals():
    """Data-only numpy reconstructors a reference-trained checkpoint carries next to ``model_state_dict``: the trainers store
    ``best_f1`` / ``best_acc`` and ``val_metrics`` as numpy scalars (src/det/val.py:111-115 -> src/det/train.py:266-272,
    src/rec2/train.py:237-260).  Allow-listing them keeps ``weights_only=True`` (nothing in the file is executed)."""
    allow = [np.dtype]
    try:
        from numpy._core.multiarray import scalar
    except ImportError:  # numpy < 2
        from numpy.core.multiarray import scalar
    allow.append(scalar)
    for name in ("Float64DType", "Float32DType", "Float16DType", "Int64DType", "Int32DType", "BoolDType", "UInt8DType"):
        t = getattr(getattr(np, "dtypes", None), name, None)
        if t is not None:
            allow.append(t)
    return allow


def load_checkpoint(model_path: str):
    """torch.load of a reference checkpoint file with the weights-only unpickler; returns whatever the file holds (a wrapped dict with
    ``model_state_dict`` or a bare state_dict -- ``weights.unwrap_checkpoint`` takes either, pipeline2.py:46-52,75-80)."""
    with torch.serialization.safe_globals(_numpy_data_globals()):
        return torch.load(model_path, map_location="cpu", weights_only=True)


def load_detection_model(model_path: str, device: str = "cuda:0", dtype: str = "f32") -> DBNetPP:
    """pipeline2.py:43-67.  The checkpoint is read with ``weights_only=True`` (nothing in the file is executed)."""
    return DBNetPP(pretrained=False, state_dict=load_checkpoint(model_path), device=device, dtype=dtype)


def load_recognition_model(model_path: str, device: str = "cuda:0", variant: str = "base", dtype: str = "f32") -> SVTRv2:
    """pipeline2.py:72-89."""
    return SVTRv2(variant=variant, in_channels=3, state_dict=load_checkpoint(model_path), device=device, dtype=dtype)


def _dev_index(device) -> int:
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def resize_image_for_det(image, image_size: int = 640):
    """pipeline2.py:33-40: scale the longer side to ``image_size``, round both sides to multiples of 32, cv2.resize (bilinear).
    ``image``: uint8 HWC numpy array or device tensor.  Returns (resized uint8 HWC device tensor, (scale_h, scale_w))."""
    img = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    img = img.cuda().contiguous() if not img.is_cuda else img.contiguous()
    h, w = img.shape[:2]
    scale = image_size / max(h, w)
    new_h = int(np.round(h * scale / 32) * 32)
    new_w = int(np.round(w * scale / 32) * 32)
    out = torch.empty((new_h, new_w, 3), dtype=torch.uint8, device=img.device)
    stream = torch.cuda.current_stream(img.device).cuda_stream
    _lib.check(_lib.load().ocrvi_resize_u8(_dev_index(img.device), img.data_ptr(), h, w, out.data_ptr(), new_h, new_w, stream))
    return out, (new_h / h, new_w / w)


def normalize_for_det(images_u8: torch.Tensor) -> torch.Tensor:
    """uint8 HWC [N,H,W,3] on the device -> float32 NCHW, exactly the arithmetic of pipeline2.py:312-314."""
    if images_u8.dim() == 3:
        images_u8 = images_u8[None]
    images_u8 = images_u8.contiguous()
    N, H, W, C = images_u8.shape
    assert C == 3 and images_u8.dtype == torch.uint8 and images_u8.is_cuda
    out = torch.empty((N, 3, H, W), dtype=torch.float32, device=images_u8.device)
    stream = torch.cuda.current_stream(images_u8.device).cuda_stream
    _lib.check(_lib.load().ocrvi_normalize_u8(_dev_index(images_u8.device), images_u8.data_ptr(), N, H, W, out.data_ptr(), stream))
    return out


class DBPostProcessor:
    """src/det/test.py:44-106: probability map -> unclipped text polygons + scores.  Same constructor, attributes and call signature."""

    def __init__(self, thresh=0.3, box_thresh=0.6, max_candidates=1000, unclip_ratio=1.5):
        self.thresh = thresh
        self.box_thresh = box_thresh
        self.max_candidates = max_candidates
        self.unclip_ratio = unclip_ratio
        self.min_size = 3
        self.min_area = 10

    def __call__(self, pred, is_output_polygon=False):
        """pred: (1, H, W) float array or tensor (a device tensor is copied to the host, as pipeline2.py:320 does).
        Returns (boxes, scores): boxes[i] is an (n_i, 2) int64 array of polygon vertices, scores[i] a float."""
        if isinstance(pred, torch.Tensor):
            pred = pred.detach().float().cpu().numpy()
        pred = np.asarray(pred)
        if pred.ndim == 3:
            pred = pred[0]
        if pred.ndim != 2:
            raise ValueError(f"DBPostProcessor expects (1, H, W) or (H, W), got shape {pred.shape}")
        prob = np.ascontiguousarray(pred, dtype=np.float32)
        H, W = prob.shape
        lib = _lib.load()
        cap_boxes = max(int(self.max_candidates), 1)
        cap_points = 4 * (H + W) + 4096
        while True:
            points = np.empty((cap_points, 2), np.int32)
            offs = np.empty(cap_boxes + 1, np.int32)
            scores = np.empty(cap_boxes, np.float32)
            n = ctypes.c_int32(0)
            rc = lib.ocrvi_db_postprocess(prob.ctypes.data, H, W, float(self.thresh), float(self.box_thresh), int(self.max_candidates),
                                          float(self.unclip_ratio), float(self.min_area), points.ctypes.data, cap_points, offs.ctypes.data,
                                          scores.ctypes.data, cap_boxes, ctypes.byref(n))
            if rc == -3 and cap_points < (1 << 28):
                cap_points *= 4          # more polygon vertices than guessed: retry with room
                continue
            _lib.check(rc)
            break
        boxes = [points[offs[i]:offs[i + 1]].astype(np.int64) for i in range(n.value)]
        return boxes, [float(v) for v in scores[:n.value]]


def db_boxes_batch(prob_maps, post_processor: "DBPostProcessor", scale_w: float = 1.0, scale_h: float = 1.0, orig_hw: Tuple[int, int] = None,
                   page_base: int = 0, threads: int = 8, cap_per_page: int = None, out=None):
    """The host middle of pipeline2.py:320-343 for a batch of pages, in one GIL-free call: ``post_processor`` on each map ->
    ``rescale_boxes`` -> ``crop_rect``.  ``prob_maps``: float32 [n,H,W] host array (numpy, or a CPU/pinned torch tensor).
    Returns (rects int32 [total,5] = (page, x, y, w, h) in page order, counts int32 [n], scores float32 [total])."""
    if isinstance(prob_maps, torch.Tensor):
        assert not prob_maps.is_cuda and prob_maps.dtype == torch.float32 and prob_maps.is_contiguous()
        n, H, W = prob_maps.shape[-3:] if prob_maps.dim() >= 3 else (1,) + tuple(prob_maps.shape)
        ptr = prob_maps.data_ptr()
    else:
        prob_maps = np.ascontiguousarray(prob_maps, dtype=np.float32)
        if prob_maps.ndim == 2:
            prob_maps = prob_maps[None]
        n, H, W = prob_maps.shape
        ptr = prob_maps.ctypes.data
    oh, ow = orig_hw if orig_hw is not None else (H, W)
    cap = int(cap_per_page or post_processor.max_candidates)
    if out is None:
        out = (np.empty((n, cap, 5), np.int32), np.empty((n, cap), np.float32), np.empty(n, np.int32))
    rects, scores, counts = out
    _lib.check(_lib.load().ocrvi_db_boxes_batch(ptr, n, H, W, float(post_processor.thresh), float(post_processor.box_thresh),
                                                int(post_processor.max_candidates), float(post_processor.unclip_ratio),
                                                float(post_processor.min_area), float(scale_w), float(scale_h), int(oh), int(ow), int(page_base),
                                                rects.ctypes.data, scores.ctypes.data, cap, counts.ctypes.data, int(threads)))
    keep = [rects[i, :counts[i]] for i in range(n)]
    return (np.concatenate(keep, 0) if keep else np.empty((0, 5), np.int32)), counts[:n].copy(), \
        np.concatenate([scores[i, :counts[i]] for i in range(n)], 0)


class DBComponents:
    """Device half of DB post-processing for a fixed page geometry: ``run(prob)`` thresholds DEVICE maps [n,H,W] and labels their
    8-connected components (``ocrvi_db_components``), leaving in HBM the 1-bit mask, the component table and the probability values
    inside the component boxes; ``to_host()`` (or the enqueue-only ``copy_async()``) copies those -- not the maps -- into one of
    ``host_slots`` pinned buffer sets, and ``boxes()`` finishes the stage on the host (``ocrvi_db_boxes_batch_sparse``), falling back to
    the full map for a page whose table or boxes overflowed.  Replaces the ``.cpu().numpy()`` of the whole map at pipeline2.py:320 plus the
    thresholding at src/det/test.py:57."""

    def __init__(self, n_pages: int, H: int, W: int, device="cuda:0", cap: int = 4096, pack_frac: float = 0.5, host_slots: int = 1):
        assert W % 32 == 0, "page width must be a multiple of 32"
        self.n, self.H, self.W, self.cap = n_pages, H, W, cap
        self.dev = torch.device(device)
        self.devi = _dev_index(self.dev)   # an index-less 'cuda' means the current device, as in DBNetPP / SVTRv2
        self.pack_cap = int(H * W * pack_frac)
        d = dict(device=self.dev)
        self.bits = torch.empty((n_pages, H, W // 32), dtype=torch.int32, **d)
        self.comps = torch.empty((n_pages, cap, 8), dtype=torch.int32, **d)
        self.counts = torch.empty((n_pages,), dtype=torch.int32, **d)
        self.offsets = torch.empty((n_pages, cap + 1), dtype=torch.int64, **d)
        self.packed = torch.empty((n_pages, self.pack_cap), dtype=torch.float32, **d)
        self.ws = torch.empty((_lib.load().ocrvi_db_components_workspace_bytes(n_pages, H, W),), dtype=torch.uint8, **d)
        self._dev_bufs = (self.counts, self.offsets, self.comps, self.bits, self.packed)
        self.host = [tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in self._dev_bufs) for _ in range(host_slots)]
        self._select(0)

    def _select(self, slot):
        self.h_counts, self.h_offsets, self.h_comps, self.h_bits, self.h_packed = self.host[slot]

    def run(self, prob: torch.Tensor, thresh: float, n_pages: int = None, stream=None):
        n = n_pages or self.n
        assert prob.is_cuda and prob.dtype == torch.float32 and prob.is_contiguous() and prob.numel() >= n * self.H * self.W and n <= self.n
        st = stream if stream is not None else torch.cuda.current_stream(self.dev)
        _lib.check(_lib.load().ocrvi_db_components(self.devi, prob.data_ptr(), n, self.H, self.W, float(thresh), self.bits.data_ptr(),
                                                   self.comps.data_ptr(), self.counts.data_ptr(), self.cap, self.offsets.data_ptr(),
                                                   self.packed.data_ptr(), self.pack_cap, self.ws.data_ptr(), st.cuda_stream))

    def copy_async(self, slot: int = 0, n_pages: int = None, pack_floats: int = None) -> int:
        """Enqueue-only: the tables, the mask and the first ``pack_floats`` (default: all) packed values of every page, on the current
        stream, into host slot ``slot``.  Returns the bytes enqueued.  (A page that packed more than was copied is caught by ``boxes``.)"""
        n = n_pages or self.n
        k = self.pack_cap if pack_floats is None else min(int(pack_floats), self.pack_cap)
        moved = 0
        for h, dv in zip(self.host[slot][:4], self._dev_bufs[:4]):
            h[:n].copy_(dv[:n], non_blocking=True)
            moved += h[:n].numel() * h.element_size()
        self.host[slot][4][:n, :k].copy_(self.packed[:n, :k], non_blocking=True)
        self._copied = k
        return moved + n * k * 4

    def to_host(self, n_pages: int = None, slot: int = 0) -> int:
        """Two-phase copy on the current stream: the small tables first, then only as much of each page's packed values as its table
        says were written.  Returns the bytes that crossed PCIe."""
        n = n_pages or self.n
        self._select(slot)
        moved = 0
        for h, dv in zip(self.host[slot][:4], self._dev_bufs[:4]):
            h[:n].copy_(dv[:n], non_blocking=True)
            moved += h[:n].numel() * h.element_size()
        torch.cuda.current_stream(self.dev).synchronize()
        for pg in range(n):
            c = int(self.h_counts[pg])
            tot = int(self.h_offsets[pg, min(c, self.cap)])
            if 0 < tot <= self.pack_cap and c <= self.cap:
                self.h_packed[pg, :tot].copy_(self.packed[pg, :tot], non_blocking=True)
                moved += tot * 4
        torch.cuda.current_stream(self.dev).synchronize()
        self._copied = self.pack_cap
        return moved

    def boxes(self, post_processor: "DBPostProcessor", prob: torch.Tensor = None, scale_w: float = 1.0, scale_h: float = 1.0,
              orig_hw: Tuple[int, int] = None, page_base: int = 0, threads: int = 8, n_pages: int = None, slot: int = 0, cap_per_page: int = None):
        """Host finish from the buffers of host slot ``slot``.  Returns what ``db_boxes_batch`` returns.  ``prob`` (the device maps) is
        only read for pages that overflowed the table or the packed buffer (or the part of it that was copied)."""
        n = n_pages or self.n
        self._select(slot)
        oh, ow = orig_hw if orig_hw is not None else (self.H, self.W)
        cap = int(cap_per_page or post_processor.max_candidates)
        rects, scores = np.empty((n, cap, 5), np.int32), np.empty((n, cap), np.float32)
        counts, skipped = np.empty(n, np.int32), np.zeros(n, np.int32)
        # pack_cap passed down = what was actually copied: a page with more packed values than that is skipped, not read past the copy
        limit = min(getattr(self, "_copied", self.pack_cap), self.pack_cap)
        lib = _lib.load()
        if limit == self.pack_cap:
            _lib.check(lib.ocrvi_db_boxes_batch_sparse(
                self.h_bits.data_ptr(), self.h_comps.data_ptr(), self.h_counts.data_ptr(), self.cap, self.h_offsets.data_ptr(),
                self.h_packed.data_ptr(), self.pack_cap, n, self.H, self.W, float(post_processor.box_thresh), int(post_processor.max_candidates),
                float(post_processor.unclip_ratio), float(post_processor.min_area), float(scale_w), float(scale_h), int(oh), int(ow), int(page_base),
                rects.ctypes.data, scores.ctypes.data, cap, counts.ctypes.data, int(threads), skipped.ctypes.data))
        else:   # rows of the packed buffer are pack_cap apart but only `limit` floats of each are valid: one page per call
            for pg in range(n):
                _lib.check(lib.ocrvi_db_boxes_batch_sparse(
                    self.h_bits[pg].data_ptr(), self.h_comps[pg].data_ptr(), self.h_counts[pg:].data_ptr(), self.cap, self.h_offsets[pg].data_ptr(),
                    self.h_packed[pg].data_ptr(), limit, 1, self.H, self.W, float(post_processor.box_thresh), int(post_processor.max_candidates),
                    float(post_processor.unclip_ratio), float(post_processor.min_area), float(scale_w), float(scale_h), int(oh), int(ow),
                    int(page_base) + pg, rects[pg].ctypes.data, scores[pg].ctypes.data, cap, counts[pg:].ctypes.data, 1, skipped[pg:].ctypes.data))
        for pg in np.nonzero(skipped)[0]:
            if prob is None:
                raise RuntimeError(f"db components: page {pg} overflowed the component table / packed buffer and no full map was given")
            full = prob.reshape(-1, self.H, self.W)[pg:pg + 1].cpu()
            r, c, sc = db_boxes_batch(full, post_processor, scale_w, scale_h, (oh, ow), page_base + int(pg), 1, cap_per_page=cap)
            counts[pg] = c[0]
            rects[pg, :c[0]] = r
            scores[pg, :c[0]] = sc
        keep = [rects[i, :counts[i]] for i in range(n)]
        return (np.concatenate(keep, 0) if keep else np.empty((0, 5), np.int32)), counts.copy(), \
            np.concatenate([scores[i, :counts[i]] for i in range(n)], 0)


def rescale_boxes(boxes, scale_w: float, scale_h: float) -> List[np.ndarray]:
    """pipeline2.py:324-328: the in-place true-divide into the integer array truncates toward zero; then ``astype(int32)``."""
    out = []
    for box in boxes:
        b = np.asarray(box).astype(np.int64)
        b[:, 0] = np.trunc(b[:, 0] / scale_w)
        b[:, 1] = np.trunc(b[:, 1] / scale_h)
        out.append(b.astype(np.int32))
    return out


def crop_rect(img_hw: Tuple[int, int], box) -> Tuple[int, int, int, int]:
    """The rectangle crop_image (src/det/test.py:123-130) slices: cv2.boundingRect of the integer box, clamped to the image."""
    h, w = img_hw
    pts = np.asarray(box).reshape(-1, 2)
    x0, y0 = int(pts[:, 0].min()), int(pts[:, 1].min())
    bw, bh = int(pts[:, 0].max()) - x0 + 1, int(pts[:, 1].max()) - y0 + 1     # boundingRect of integer points is inclusive
    x, y = max(0, x0), max(0, y0)
    return x, y, max(min(bw, w - x), 0), max(min(bh, h - y), 0)


def crop_image(img: np.ndarray, box) -> np.ndarray:
    x, y, bw, bh = crop_rect(img.shape[:2], box)
    return img[y:y + bh, x:x + bw]


def preprocess_crops(images_u8: torch.Tensor, rects: Sequence[Sequence[int]], img_size: Tuple[int, int] = (32, 256)) -> torch.Tensor:
    """Batched crop + preprocess_for_recognition (pipeline2.py:92-128) on the device.
    images_u8 [N,H,W,3] uint8 on the device; rects = (image index, x, y, w, h) -> float32 [B,3,h,w]."""
    images_u8 = images_u8.contiguous()
    N, H, W, _ = images_u8.shape
    r = torch.as_tensor(np.asarray(rects, dtype=np.int32).reshape(-1, 5), device=images_u8.device)
    out = torch.empty((r.shape[0], 3, img_size[0], img_size[1]), dtype=torch.float32, device=images_u8.device)
    stream = torch.cuda.current_stream(images_u8.device).cuda_stream
    _lib.check(_lib.load().ocrvi_crop_resize_normalize(_dev_index(images_u8.device), images_u8.data_ptr(), N, H, W, r.data_ptr(), r.shape[0],
                                                       img_size[0], img_size[1], out.data_ptr(), stream))
    return out


def preprocess_for_recognition(crop: np.ndarray, img_size: Tuple[int, int] = (32, 256), device: str = "cuda:0") -> torch.Tensor:
    """pipeline2.py:92-128 for one crop (HxWx3 uint8, or HxW grey which is replicated as cv2.COLOR_GRAY2RGB does) -> [3,h,w]."""
    if crop.ndim == 2:
        crop = np.repeat(crop[:, :, None], 3, axis=2)
    crop = np.ascontiguousarray(crop[:, :, :3])
    if crop.size == 0:
        return torch.zeros((3,) + tuple(img_size), device=device)
    img = torch.from_numpy(crop).to(device)[None]
    return preprocess_crops(img, [(0, 0, 0, crop.shape[1], crop.shape[0])], img_size)[0]


def recognize_text(model: SVTRv2, crop: np.ndarray, device: str = "cuda:0", img_size: Tuple[int, int] = (32, 256)) -> str:
    """pipeline2.py:131-141."""
    out = recognize_text_batch(model, [crop], device, img_size, 1)
    return out[0] if out else ""


def recognize_text_batch(model: SVTRv2, crops: List[np.ndarray], device: str = "cuda:0", img_size: Tuple[int, int] = (32, 256),
                         batch_size: int = 32) -> List[str]:
    """pipeline2.py:144-168: batches of ``batch_size`` crops -> forward -> greedy CTC strings.  Empty crops become the all-zero tensor
    of :154-156."""
    texts: List[str] = []
    for i in range(0, len(crops), batch_size):
        ts = [preprocess_for_recognition(c, img_size, device) for c in crops[i:i + batch_size]]
        texts.extend(model.decode_greedy(torch.stack(ts)))
    return texts


def detect_and_recognize(original_image, det_model, rec_model, post_processor: DBPostProcessor, device: str = "cuda:0", det_size: int = 640,
                         rec_size: Tuple[int, int] = (32, 256), rec_batch_size: int = 64):
    """Steps 2 and 3 of the reference's per-image loop (pipeline2.py:306-352) with every stage on this library: resize + normalise on the
    device -> ``det_model`` -> ``post_processor`` on the host copy of the binary map -> boxes rescaled to the original image -> the
    bounding rectangle of each box cropped, resized and normalised on the device straight from the uploaded page -> ``rec_model`` greedy
    CTC in batches of ``rec_batch_size``.  ``original_image``: RGB uint8 HxWx3 (numpy or device tensor).
    Returns (rescaled_boxes [int32 (n_i, 2)], scores, texts); empty crops decode the all-zero tensor as pipeline2.py:154-156 does."""
    page = original_image if isinstance(original_image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(original_image))
    page = page.to(device).contiguous()
    h, w = page.shape[:2]
    resized, (scale_h, scale_w) = resize_image_for_det(page, det_size)
    det_preds = det_model(normalize_for_det(resized))
    pred_binary = det_preds["binary"] if isinstance(det_preds, dict) else det_preds
    boxes, scores = post_processor(pred_binary[0])       # (copies the map to the host: the stream is synchronised from here on)
    if hasattr(det_model, "check_range"):
        det_model.check_range()                           # f16x2 only: OverflowError if an activation left fp16's range
    rescaled = rescale_boxes(boxes, scale_w, scale_h)
    rects = [(0,) + crop_rect((h, w), b) for b in rescaled]
    texts: List[str] = []
    for i in range(0, len(rects), rec_batch_size):
        chunk = rects[i:i + rec_batch_size]
        live = [r for r in chunk if r[3] > 0 and r[4] > 0]
        batch = torch.zeros((len(chunk), 3) + tuple(rec_size), device=page.device)
        if live:
            idx = torch.as_tensor([j for j, r in enumerate(chunk) if r[3] > 0 and r[4] > 0], device=page.device)
            batch[idx] = preprocess_crops(page[None], live, rec_size)
        texts.extend(rec_model.decode_greedy(batch))
    return rescaled, scores, texts
