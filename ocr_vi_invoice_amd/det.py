"""DBNet++ detector on libocrvi -- host-side mirror of the reference module's inference API
(model/det/dbnet.py:6-17): same constructor arguments and the same five-map output dict (head.py:42-48)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib, weights


class DBNetPP:
    def __init__(self, backbone: str = "resnet50", pretrained: bool = False, in_channels: int = 3, inner_channels: int = 256,
                 k: float = 50, dcn: bool = True, *, state_dict=None, seed: int = 1234, dtype="bf16", device="cuda:0"):
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
        self.load_state_dict(state_dict if state_dict is not None else weights.make_det_state_dict(seed))

    def load_state_dict(self, state_dict, strict: bool = True):
        blob = weights.pack_blob(weights.fold_det(state_dict))
        cfg = _lib.DetCfg()
        cfg.dtype = self.dtype
        cfg.k = self.k
        cfg.max_batch = 0
        h = C.c_void_p()
        _lib.check(_lib.load().ocrvi_det_create(self._dev_index(), blob, len(blob), C.byref(cfg), C.byref(h)))
        self._free()
        self._handle = h
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise ValueError("the handle is bound to its device at construction; pass device= to the constructor")
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
