"""ctypes binding of libocrvi.so (include/ocrvi.h).  There is no CPU fallback: if the HIP library is
missing or fails to load, everything that needs it raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# (OCRVI_LIB: development aid -- an instrumented build of the same library, e.g. the ring-GEMM phase-stamp build tools/ring_prof.py uses)
LIB_PATH = os.environ.get("OCRVI_LIB") or os.path.join(_HERE, "lib", "libocrvi.so")

OCRVI_F32, OCRVI_BF16, OCRVI_F16, OCRVI_F16X2 = 0, 1, 2, 3
DTYPES = {"f32": OCRVI_F32, "fp32": OCRVI_F32, "float32": OCRVI_F32, "bf16": OCRVI_BF16, "bfloat16": OCRVI_BF16,
          "f16": OCRVI_F16, "fp16": OCRVI_F16, "float16": OCRVI_F16,
          # fp32-equivalent arithmetic on the 16-bit matrix pipe: every operand kept as two fp16 halves (include/ocrvi.h)
          "f16x2": OCRVI_F16X2}
ABI_VERSION = 2

EXPORTS = [
    "ocrvi_last_error", "ocrvi_abi_version",
    "ocrvi_det_create", "ocrvi_det_destroy", "ocrvi_det_workspace_bytes", "ocrvi_det_forward", "ocrvi_det_debug_features",
    "ocrvi_rec_create", "ocrvi_rec_destroy", "ocrvi_rec_workspace_bytes", "ocrvi_rec_forward", "ocrvi_rec_debug_features",
    "ocrvi_ctc_greedy", "ocrvi_normalize_u8", "ocrvi_resize_u8", "ocrvi_crop_resize_normalize", "ocrvi_db_postprocess", "ocrvi_db_boxes_batch",
    "ocrvi_unclip_polygon", "ocrvi_db_components_workspace_bytes", "ocrvi_db_components", "ocrvi_db_boxes_batch_sparse",
    "ocrvi_test_deform_conv", "ocrvi_test_offset_conv", "ocrvi_test_conv", "ocrvi_test_gemm", "ocrvi_test_attention", "ocrvi_test_mlp", "ocrvi_test_stem_pool", "ocrvi_test_pack_f16x2",
    "ocrvi_prof_enable", "ocrvi_prof_reset", "ocrvi_prof_report",
    "ocrvi_det_status", "ocrvi_rec_status", "ocrvi_range_reset", "ocrvi_range_flag",
]


class DetCfg(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("k", C.c_float), ("max_batch", C.c_int32), ("reserved", C.c_int32 * 5)]


class RecCfg(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("dims", C.c_int32 * 3), ("num_blocks", C.c_int32 * 3), ("num_local", C.c_int32 * 3),
                ("num_classes", C.c_int32), ("blank_id", C.c_int32), ("reserved", C.c_int32 * 4)]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libocrvi.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ocr_vi_invoice_amd/csrc`).  There is no CPU fallback for the product path.")
    # torch must own the HIP runtime: importing it first makes libocrvi's libamdhip64 dependency resolve to the copy torch
    # already loaded, so streams, device pointers and the device context are shared (a second runtime sees no device).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32p, i32p, sz = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t
    lib.ocrvi_last_error.restype = C.c_char_p
    lib.ocrvi_last_error.argtypes = []
    lib.ocrvi_abi_version.restype = i32
    lib.ocrvi_det_create.argtypes = [i32, vp, sz, C.POINTER(DetCfg), C.POINTER(vp)]
    lib.ocrvi_det_destroy.argtypes = [vp]
    lib.ocrvi_det_destroy.restype = None
    lib.ocrvi_det_workspace_bytes.argtypes = [vp, i32, i32, i32, C.POINTER(sz)]
    lib.ocrvi_det_forward.argtypes = [vp, f32p, i32, i32, i32, f32p, f32p, f32p, f32p, f32p, vp, sz, vp]
    lib.ocrvi_det_debug_features.argtypes = [vp, i32, i32, i32, f32p, f32p, f32p, f32p, f32p, vp, sz, vp]
    lib.ocrvi_rec_create.argtypes = [i32, vp, sz, C.POINTER(RecCfg), C.POINTER(vp)]
    lib.ocrvi_rec_destroy.argtypes = [vp]
    lib.ocrvi_rec_destroy.restype = None
    lib.ocrvi_rec_workspace_bytes.argtypes = [vp, i32, i32, i32, C.POINTER(sz)]
    lib.ocrvi_rec_forward.argtypes = [vp, f32p, i32, i32, i32, f32p, i32p, i32p, i32p, vp, sz, vp]
    lib.ocrvi_rec_debug_features.argtypes = [vp, i32, i32, i32, f32p, f32p, vp, sz, vp]
    lib.ocrvi_ctc_greedy.argtypes = [i32, f32p, i32, i32, i32, i32, i32p, i32p, i32p, vp]
    lib.ocrvi_normalize_u8.argtypes = [i32, vp, i32, i32, i32, f32p, vp]
    lib.ocrvi_resize_u8.argtypes = [i32, vp, i32, i32, vp, i32, i32, vp]
    lib.ocrvi_crop_resize_normalize.argtypes = [i32, vp, i32, i32, i32, i32p, i32, i32, i32, f32p, vp]
    lib.ocrvi_db_postprocess.argtypes = [vp, i32, i32, C.c_float, C.c_float, i32, C.c_float, C.c_float, vp, i32, vp, vp, i32, C.POINTER(i32)]
    lib.ocrvi_db_components_workspace_bytes.argtypes = [i32, i32, i32]
    lib.ocrvi_db_components_workspace_bytes.restype = sz
    lib.ocrvi_db_components.argtypes = [i32, vp, i32, i32, i32, C.c_float, vp, vp, vp, i32, vp, vp, C.c_longlong, vp, vp]
    lib.ocrvi_db_boxes_batch_sparse.argtypes = [vp, vp, vp, i32, vp, vp, C.c_longlong, i32, i32, i32, C.c_float, i32, C.c_float, C.c_float,
                                                C.c_double, C.c_double, i32, i32, i32, vp, vp, i32, vp, i32, vp]
    lib.ocrvi_unclip_polygon.argtypes = [vp, i32, C.c_double, vp, i32, C.POINTER(i32)]
    lib.ocrvi_db_boxes_batch.argtypes = [vp, i32, i32, i32, C.c_float, C.c_float, i32, C.c_float, C.c_float, C.c_double, C.c_double, i32, i32, i32,
                                         vp, vp, i32, vp, i32]
    lib.ocrvi_test_deform_conv.argtypes = [i32, i32, f32p, f32p, f32p, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32p, i32,
                                           C.POINTER(C.c_float)]
    lib.ocrvi_test_offset_conv.argtypes = [i32, i32, f32p, vp, vp, i32, i32, i32, i32, i32, f32p, i32, C.POINTER(C.c_float)]
    lib.ocrvi_test_conv.argtypes = [i32, i32, f32p, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32p, i32,
                                    C.POINTER(C.c_float)]
    lib.ocrvi_test_gemm.argtypes = [i32, i32, f32p, vp, vp, f32p, i32, i32, i32, i32, i32, i32, f32p, i32, C.POINTER(C.c_float)]
    lib.ocrvi_test_stem_pool.argtypes = [i32, i32, f32p, vp, vp, i32, i32, i32, i32, f32p, i32, C.POINTER(C.c_float)]
    lib.ocrvi_test_attention.argtypes = [i32, i32, f32p, i32, i32, i32, f32p, i32, C.POINTER(C.c_float)]
    lib.ocrvi_test_mlp.argtypes = [i32, i32, f32p, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32p, i32, C.POINTER(C.c_float)]
    lib.ocrvi_test_pack_f16x2.argtypes = [f32p, sz, vp, C.POINTER(C.c_float)]
    lib.ocrvi_det_status.argtypes = [vp]
    lib.ocrvi_rec_status.argtypes = [vp]
    lib.ocrvi_range_reset.argtypes = [i32, vp]
    lib.ocrvi_range_flag.argtypes = [i32, C.POINTER(i32)]
    lib.ocrvi_prof_enable.argtypes = [i32]
    lib.ocrvi_prof_reset.argtypes = []
    lib.ocrvi_prof_report.argtypes = [C.c_char_p, sz]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("ocrvi_last_error", "ocrvi_det_destroy", "ocrvi_rec_destroy"):
            fn.restype = i32
    if lib.ocrvi_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libocrvi ABI {lib.ocrvi_abi_version()} != binding {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def last_error() -> str:
    return load().ocrvi_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """0 ok; OCRVI_EINVAL -> ValueError (shape/argument), OCRVI_ENOMEM -> MemoryError, OCRVI_ERANGE -> OverflowError (an f16x2 activation
    left fp16's exponent range), anything else -> RuntimeError."""
    if rc == 0:
        return
    msg = last_error()
    if rc == -1:
        raise ValueError(msg)
    if rc == -3:
        raise MemoryError(msg)
    if rc == -5:
        raise OverflowError(msg)
    raise RuntimeError(f"libocrvi error {rc}: {msg}")


def dtype_code(dtype) -> int:
    if isinstance(dtype, int):
        return dtype
    try:
        return DTYPES[str(dtype).replace("torch.", "")]
    except KeyError:
        raise ValueError(f"unsupported compute dtype {dtype!r}; choose from f32, f16x2, bf16, f16") from None


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def prof_report() -> dict:
    """Aggregated per-kernel-tag timings recorded since the last reset (see ocrvi_prof_report)."""
    import json
    buf = C.create_string_buffer(1 << 20)
    check(load().ocrvi_prof_report(buf, len(buf)))
    return json.loads(buf.value.decode())
