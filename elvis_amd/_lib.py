"""ctypes binding of libelvis_amd.so (the C ABI declared in include/elvis_amd.h).

The product path has NO CPU fallback: if the shared object is missing or fails to load,
`lib()` raises RuntimeError.  Error codes from the C side are mapped to the reference's
exception conventions: ELVIS_E_INVALID -> ValueError, everything else ->
RuntimeError carrying the device label.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# ELVIS_AMD_LIB: alternative build of the same library (kernel A/B experiments, tools/build_variant.py)
LIBPATH = os.environ.get("ELVIS_AMD_LIB") or os.path.join(HERE, "lib", "libelvis_amd.so")

F32, F16 = 0, 1
ROUND_CV2, ROUND_HALF_UP = 0, 1

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None

vp, i32, f32, i64 = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "dtype", "n", "h", "w", "cin", "cin_pitch", "cin2", "cin2_pitch", "cout", "cout_pitch",
        "ksize", "stride", "pad_before", "upsample", "ho", "wo", "act", "prologue", "subpixel")]


# name -> argtypes (every symbol include/elvis_amd.h declares; checked by tests/test_cabi.py)
SIGNATURES = {
    "elvis_abi_version": [],
    "elvis_last_error": [],
    "elvis_recompose_u8": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "elvis_area_downscale_u8": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "elvis_blend_u8": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "elvis_select_levels_u8": [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "elvis_tile_accumulate_f32": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "elvis_tile_normalize_u8": [vp, vp, vp, i32, i32, i32, vp],
    "elvis_sse_u8": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "elvis_block_ssim_u8": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "elvis_u8_to_float": [vp, vp, i32, i32, i32, i32, i32, f32, f32, i32, i32, vp],
    "elvis_float_to_u8": [vp, i32, vp, vp, i32, i32, i32, i32, f32, f32, i32, i32, vp],
    "elvis_conv_packed_weight_bytes": [C.POINTER(ConvDesc)],
    "elvis_conv_pack_weights": [C.POINTER(ConvDesc), vp, vp, vp],
    "elvis_conv2d": [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp],
    "elvis_conv_stats_tiles": [C.POINTER(ConvDesc)],
    "elvis_conv_kernel_name": [C.POINTER(ConvDesc), C.c_char_p, C.c_size_t],
    "elvis_conv_x3_eligible": [C.POINTER(ConvDesc)],
    "elvis_conv_debug_set": [C.c_char_p, i32],
    "elvis_gn_partials_to_sums": [vp, i32, i32, i32, vp, i32, i32, vp],
    "elvis_groupnorm_workspace_floats": [i32, i32, i32, i32],
    "elvis_groupnorm_sums": [vp, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp],
    "elvis_groupnorm_affine": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "elvis_affine_act": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp],
    "elvis_layernorm": [vp, vp, i32, i64, i32, i32, i32, vp, vp, f32, vp],
    "elvis_swin_packed_bytes": [i32, i32, i32],
    "elvis_swin_pack_weights": [vp, vp, vp, i32, i32, i32, vp],
    "elvis_swin_mlp": [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, vp],
    "elvis_swin_ln_linear": [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, vp],
    "elvis_swin_pack_proj_mlp": [vp, vp, vp, vp, i32, i32, vp],
    "elvis_swin_proj_mlp": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, f32, vp],
    "elvis_window_attention": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, f32, vp],
    "elvis_bicubic_upsample": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "elvis_vq_nearest": [vp, vp, vp, i32, i64, i32, i32, i32, vp, i32, vp],
    "elvis_pad_reflect_axpy": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, f32, vp],
    "elvis_crop_copy": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "elvis_convert_act": [vp, i32, vp, i32, i64, i32, vp],
    "elvis_degrade_downsample_u8": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "elvis_degrade_gaussian_u8": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, f32, vp],
    "elvis_degrade_dct_u8": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "elvis_dcnv2": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "elvis_temporal_stack": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "elvis_plane_merge": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
}


def lib() -> C.CDLL:
    """Load (once) and return the shared library; fail loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIBPATH):
            raise RuntimeError(
                f"libelvis_amd.so not found at {LIBPATH}. Build it with `python -m elvis_amd._build` "
                "(needs hipcc). There is no CPU fallback for the restoration path.")
        # torch must be imported first so that libamdhip64.so.7 resolves to the HIP runtime torch
        # already loaded (same SONAME) - streams and device pointers are then shared.
        import torch  # noqa: F401
        try:
            handle = C.CDLL(LIBPATH)
        except OSError as exc:
            raise RuntimeError(f"failed to load {LIBPATH}: {exc}") from exc
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        handle.elvis_last_error.restype = C.c_char_p
        handle.elvis_conv_packed_weight_bytes.restype = C.c_size_t
        handle.elvis_groupnorm_workspace_floats.restype = C.c_size_t
        handle.elvis_swin_packed_bytes.restype = C.c_size_t
        if handle.elvis_abi_version() != 1:
            raise RuntimeError("libelvis_amd.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc: int, device=None) -> None:
    if rc == 0:
        return
    msg = lib().elvis_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    label = str(device) if device is not None else "unknown device"
    raise RuntimeError(f"elvis_amd kernel failed on {label}: {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_handle(device=None) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(device) -> None:
    import torch
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"elvis_amd runs on MI355X only (got device '{dev}'); there is no CPU path in the product. "
            "Use the reference implementation on CPU.")
    if not torch.cuda.is_available():
        raise RuntimeError("no ROCm GPU visible to torch; elvis_amd cannot run")


def dtype_code(torch_dtype) -> int:
    import torch
    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.float16:
        return F16
    raise ValueError(f"unsupported dtype {torch_dtype}")
