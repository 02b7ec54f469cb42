"""DBNet++ detector on libocrvi -- host-side mirror of the reference module's inference API
(model/det/dbnet.py:6-17): same constructor arguments and the same five-map output dict (head.py:42-48)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib, weights


class DBNetPP:
    def __init__(self, backbone: str = "resnet50", pretrained: bool = False, in_channels: int = 3, inner_channels: int = 256,
                 k: float = 50, dcn: bool = True, *, state_dict=None, blob: bytes = None, seed: int = 1234, dtype="f32",
                 device="cuda:0"):
        if backbone != "resnet50":
            # the reference also offers resnet18 (backbone.py:12-15); the pipeline only uses resnet50 (pipeline2.py:45)
            raise NotImplementedError(f"Backbone {backbone} not implemented")
        if pretrained:
            raise RuntimeError("pretrained=True downloads torchvision ImageNet weights (backbone.py:17); there is no network -- "
                               "pass state_dict= (inference callers use pretrained=False, pipeline2.py:45)")
        if in_channels != 3 or inner_channels != 256 or not dcn:
            raise ValueError("only the pipeline configuration (in_channels=3, inner_channels=256, dcn=True) is built")
        self.k = float(k)
        self.device = torch.device(device)
        self.dtype = _lib.dtype_code(dtype)
        self._handle = None
        self._ws = {}
        self.training = False
        self._seed = seed
        if blob is not None:        # already folded + packed (weights.pack_blob): what rank 0 broadcasts to the other ranks
            self.load_blob(blob)
        else:
            self.load_state_dict(state_dict if state_dict is not None else weights.make_det_state_dict(seed))

    def load_state_dict(self, state_dict, strict: bool = True):
        """nn.Module.load_state_dict semantics for the keys the inference graph uses: with ``strict`` (default) a missing tensor
        raises KeyError-as-RuntimeError like torch does; unexpected keys (the reference's duplicate ``backbone.layerN`` aliases, the
        unused ``fc``, optimizer state in a wrapped checkpoint) are ignored, as the reference's loader effectively does."""
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
            folded = weights.fold_det(state_dict)
        except KeyError as e:
            raise RuntimeError(f"Error(s) in loading state_dict for DBNetPP: missing key {e}") from None
        self._state = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in weights.unwrap_checkpoint(state_dict).items()}
        return self.load_blob(weights.pack_blob(folded), _from_state=True)

    def state_dict(self):
        """The state_dict this model was loaded from (reference key schema); None-safe copy."""
        if getattr(self, "_state", None) is None:
            raise RuntimeError("model was built from a packed blob; no state_dict is retained")
        return dict(self._state)

    def load_blob(self, blob: bytes, _from_state: bool = False):
        if not _from_state:
            self._state = None      # a packed blob carries no unfolded state_dict (state_dict() / strict=False then raise)
        cfg = _lib.DetCfg()
        cfg.dtype = self.dtype
        cfg.k = self.k
        cfg.max_batch = 0
        h = C.c_void_p()
        _lib.check(_lib.load().ocrvi_det_create(self._dev_index(), blob, len(blob), C.byref(cfg), C.byref(h)))
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
            _lib.load().ocrvi_det_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _workspace(self, N, H, W) -> torch.Tensor:
        key = (N, H, W)
        ws = self._ws.get(key)
        if ws is None:
            n = C.c_size_t()
            _lib.check(_lib.load().ocrvi_det_workspace_bytes(self._handle, N, H, W, C.byref(n)))
            self._ws.clear()
            ws = torch.empty(n.value, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def forward(self, x: torch.Tensor, binary_only: bool = False) -> Dict[str, torch.Tensor]:
        """(N,3,H,W), H,W % 32 == 0 -> {'binary','thresh','thresh_binary','bin_logits','thresh_logits'}, each (N,1,H,W) float32
        on the device (dbnet.py:13-17, head.py:42-48).  ``binary_only`` skips writing the four maps the pipeline never reads
        (pipeline2.py:318)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected (N,3,H,W) input, got {tuple(x.shape)}")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        N, _, H, W = x.shape
        ws = self._workspace(N, H, W)
        names = ["binary"] if binary_only else ["binary", "thresh", "thresh_binary", "bin_logits", "thresh_logits"]
        out = {n: torch.empty((N, 1, H, W), dtype=torch.float32, device=self.device) for n in names}
        stream = torch.cuda.current_stream(self.device).cuda_stream
        p = [_lib.ptr(out.get(n)) for n in ("binary", "thresh", "thresh_binary", "bin_logits", "thresh_logits")]
        _lib.check(_lib.load().ocrvi_det_forward(self._handle, x.data_ptr(), N, H, W, *p, ws.data_ptr(), ws.numel(), stream))
        return out

    __call__ = forward

    def check_range(self) -> None:
        """f16x2 mode only (a no-op otherwise): synchronise the current stream and raise OverflowError if an activation left fp16's
        exponent range (|x| >= 65520) during a forward -- the one way this mode can fail to stand in for the reference's fp32 path
        (pipeline2.py:312-318).  ``forward`` itself never synchronises; the pipeline helpers call this where they already wait for the
        map (``.cpu()``, pipeline2.py:320)."""
        if self.dtype == _lib.OCRVI_F16X2:
            torch.cuda.current_stream(self.device).synchronize()
            _lib.check(_lib.load().ocrvi_det_status(self._handle))

    def reset_range(self) -> None:
        """Clear the device's (sticky) f16x2 range flag, e.g. after handling an OverflowError."""
        _lib.check(_lib.load().ocrvi_range_reset(self._dev_index(), torch.cuda.current_stream(self.device).cuda_stream))

    def debug_features(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Test hook: c2..c5 and the fused neck feature as float32 NCHW, after a forward on x."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        N, _, H, W = x.shape
        self.forward(x, binary_only=True)
        shapes = {"c2": (256, 4), "c3": (512, 8), "c4": (1024, 16), "c5": (2048, 32), "fused": (256, 4)}
        out = {k: torch.empty((N, c, H // s, W // s), dtype=torch.float32, device=self.device) for k, (c, s) in shapes.items()}
        ws = self._workspace(N, H, W)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.load().ocrvi_det_debug_features(self._handle, N, H, W, *[out[k].data_ptr() for k in ("c2", "c3", "c4", "c5", "fused")],
                                                        ws.data_ptr(), ws.numel(), stream))
        return out
