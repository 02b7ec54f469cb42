"""SVTRv2 recogniser on libocrvi -- host-side mirror of the reference module's inference API
(model/rec2/svtrv2.py:410-569): same constructor arguments, ``forward`` / ``decode_probs`` /
``decode_greedy`` semantics and attributes (``tokenizer``, ``blank_id``, ``dims``)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib, weights
from .vocab import VOCAB, Tokenizer


class SVTRv2:
    def __init__(self, variant: str = "small", in_channels: int = 3, charset=VOCAB, dropout: float = 0.0,
                 context_window: int = 3, *, state_dict=None, blob: bytes = None, seed: int = 1234, dtype="f32", device="cuda:0"):
        assert variant in weights.REC_VARIANTS, \
            f"Unknown variant: {variant}. Choose from {list(weights.REC_VARIANTS.keys())}"  # svtrv2.py:425
        if in_channels != 3:
            raise ValueError("only 3-channel input is supported (pipeline2.py:74 uses in_channels=3)")
        self.variant = variant
        self.tokenizer = Tokenizer(charset)
        if self.tokenizer.num_classes != weights.NUM_CLASSES:
            raise ValueError(f"charset gives {self.tokenizer.num_classes} classes; kernels are built for {weights.NUM_CLASSES}")
        self.blank_id = self.tokenizer.blank_id
        self.dims = list(weights.REC_VARIANTS[variant]["dims"])
        self.device = torch.device(device)
        self.dtype = _lib.dtype_code(dtype)
        self._handle = None
        self._ws = {}
        self.training = False
        self._seed = seed
        if blob is not None:        # already folded + packed (weights.pack_blob): what rank 0 broadcasts to the other ranks
            self.load_blob(blob)
        else:
            self.load_state_dict(state_dict if state_dict is not None else weights.make_rec_state_dict(variant, seed))

    # ---- nn.Module-like surface used by the pipeline (pipeline2.py:72-82)
    def load_state_dict(self, state_dict, strict: bool = True):
        """nn.Module.load_state_dict semantics for the inference keys: a missing tensor raises RuntimeError when ``strict``; the
        training-only ``sgm.*`` keys (svtrv2.py:252-385) and checkpoint wrappers are ignored."""
        if not strict:
            # nn.Module semantics: tensors the given dict lacks keep their current values.  A model built from a packed blob (the
            # ranks behind the weight broadcast, bench.py) retains no unfolded state_dict to merge into: refuse rather than fill the
            # missing tensors with seeded random values
            if getattr(self, "_state", None) is None:
                raise RuntimeError("load_state_dict(strict=False) on a model built from a packed blob: no state_dict is retained to "
                                   "keep the missing tensors from; build the model from a state_dict, or load a complete one")
            base = dict(self._state)
            base.update(weights.unwrap_checkpoint(state_dict))
            state_dict = base
        try:
            folded = weights.fold_rec(state_dict, self.variant)
        except KeyError as e:
            raise RuntimeError(f"Error(s) in loading state_dict for SVTRv2: missing key {e}") from None
        self._state = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in weights.unwrap_checkpoint(state_dict).items()}
        return self.load_blob(weights.pack_blob(folded), _from_state=True)

    def state_dict(self):
        if getattr(self, "_state", None) is None:
            raise RuntimeError("model was built from a packed blob; no state_dict is retained")
        return dict(self._state)

    def load_blob(self, blob: bytes, _from_state: bool = False):
        if not _from_state:
            self._state = None      # a packed blob carries no unfolded state_dict (state_dict() / strict=False then raise)
        cfg = _lib.RecCfg()
        v = weights.REC_VARIANTS[self.variant]
        cfg.dtype = self.dtype
        cfg.dims[:] = v["dims"]
        cfg.num_blocks[:] = v["num_blocks"]
        cfg.num_local[:] = v["num_local"]
        cfg.num_classes = self.tokenizer.num_classes
        cfg.blank_id = self.blank_id
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.ocrvi_rec_create(self._dev_index(), blob, len(blob), C.byref(cfg), C.byref(h)))
        self._free()
        self._handle = h
        self._blob = blob
        return self

    def to(self, device):
        """Moves the model like nn.Module.to: a different device re-creates the handle there from the retained weights."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ValueError("libocrvi has no CPU path: the product runs on an MI355X (use the reference module for CPU)")
        if dev.index is None:     # model.to('cuda') (the reference's pattern, pipeline2.py:53) = the current device
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev != torch.device("cuda", self._dev_index()):
            if getattr(self, "_blob", None) is None:
                raise RuntimeError("no weights retained to move")
            self.device = dev
            self._ws = {}
            self.load_blob(self._blob)
        return self

    def eval(self):
        return self

    def _dev_index(self) -> int:
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _free(self):
        if getattr(self, "_handle", None):
            _lib.load().ocrvi_rec_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _workspace(self, B, H, W) -> torch.Tensor:
        key = (B, H, W)
        ws = self._ws.get(key)
        if ws is None:
            n = C.c_size_t()
            _lib.check(_lib.load().ocrvi_rec_workspace_bytes(self._handle, B, H, W, C.byref(n)))
            self._ws.clear()  # keep one shape's scratch resident
            ws = torch.empty(n.value, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _run(self, x: torch.Tensor, want_log_probs: bool, want_ids: bool):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected (B,3,H,W) input, got {tuple(x.shape)}")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        T = W // 4
        ws = self._workspace(B, H, W)
        lp = torch.empty((T, B, self.tokenizer.num_classes), dtype=torch.float32, device=self.device) if want_log_probs else None
        am = ids = lens = None
        if want_ids:
            am = torch.empty((B, T), dtype=torch.int32, device=self.device)
            ids = torch.empty((B, T), dtype=torch.int32, device=self.device)
            lens = torch.empty((B,), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.load().ocrvi_rec_forward(self._handle, x.data_ptr(), B, H, W, _lib.ptr(lp), _lib.ptr(am), _lib.ptr(ids),
                                                 _lib.ptr(lens), ws.data_ptr(), ws.numel(), stream))
        return lp, am, ids, lens

    def forward(self, x: torch.Tensor, targets=None) -> torch.Tensor:
        """(B,3,H,W) -> log_probs (T=W/4, B, num_classes) float32 on the device (svtrv2.py:503-536)."""
        if targets is not None:
            raise NotImplementedError("the SGM training branch (svtrv2.py:519-522) is out of scope of the inference engine")
        return self._run(x, True, False)[0]

    __call__ = forward

    def debug_features(self, x: torch.Tensor):
        """Test hook: (backbone_norm [B,N,D], frm [B,T,D]) after a forward on x."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        self._run(x, False, False)
        d = self.dims[2]
        bn = torch.empty((B, (H // 16) * (W // 4), d), dtype=torch.float32, device=self.device)
        frm = torch.empty((B, W // 4, d), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, H, W)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.load().ocrvi_rec_debug_features(self._handle, B, H, W, bn.data_ptr(), frm.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        return bn, frm

    def check_range(self) -> None:
        """f16x2 mode only (a no-op otherwise): synchronise the current stream and raise OverflowError if an activation left fp16's
        exponent range (|x| >= 65520) during a forward.  ``forward`` never synchronises; the decode paths call this where they already
        wait for the ids (``.tolist()``, svtrv2.py:562)."""
        if self.dtype == _lib.OCRVI_F16X2:
            torch.cuda.current_stream(self.device).synchronize()
            _lib.check(_lib.load().ocrvi_rec_status(self._handle))

    def reset_range(self) -> None:
        """Clear the device's (sticky) f16x2 range flag, e.g. after handling an OverflowError."""
        _lib.check(_lib.load().ocrvi_range_reset(self._dev_index(), torch.cuda.current_stream(self.device).cuda_stream))

    def _ids_to_text(self, ids: torch.Tensor, lens: torch.Tensor) -> List[str]:
        ids_h, lens_h = ids.cpu().tolist(), lens.cpu().tolist()
        self.check_range()      # (the copies above synchronised the stream already)
        return self.tokenizer.decode([row[:n] for row, n in zip(ids_h, lens_h)])

    def decode_probs(self, log_probs: torch.Tensor) -> List[str]:
        """Greedy CTC decode of (T,B,C) log-probs (svtrv2.py:545-569): argmax, collapse repeats, drop blank on the
        device; id -> text (dropping pad id 1, tokenizer.py:73) on the host."""
        lp = log_probs.to(device=self.device, dtype=torch.float32).contiguous()
        T, B, Cn = lp.shape
        am = torch.empty((B, T), dtype=torch.int32, device=self.device)
        ids = torch.empty((B, T), dtype=torch.int32, device=self.device)
        lens = torch.empty((B,), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.load().ocrvi_ctc_greedy(self._dev_index(), lp.data_ptr(), T, B, Cn, self.blank_id, am.data_ptr(), ids.data_ptr(),
                                                lens.data_ptr(), stream))
        return self._ids_to_text(ids, lens)

    def decode_greedy(self, images: torch.Tensor) -> List[str]:
        """forward + decode in one device pass (svtrv2.py:538-543)."""
        _, _, ids, lens = self._run(images, False, True)
        return self._ids_to_text(ids, lens)
