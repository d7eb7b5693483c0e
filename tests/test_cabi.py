"""The C-ABI library loads on a CPU-only host and exports every symbol include/elvis_amd.h
declares, with the arities the ctypes table binds (no compute calls without a GPU)."""
import os
import re

from elvis_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "elvis_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\s*\*)\s+(elvis_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        decls[name] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return decls


def test_header_symbols_exported_and_bound(built_lib):
    decls = _header_decls()
    assert len(decls) >= 20
    handle = _lib.lib()
    for name, nargs in decls.items():
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} not bound in elvis_amd/_lib.py"
        assert len(_lib.SIGNATURES[name]) == nargs, f"{name}: header has {nargs} args, binding {len(_lib.SIGNATURES[name])}"
    for name in _lib.SIGNATURES:
        assert name in decls, f"{name} bound but not declared in include/elvis_amd.h"
    assert handle.elvis_abi_version() == 1


def test_argument_validation_without_gpu(built_lib):
    """Entry points validate shapes before touching the device: ELVIS_E_INVALID (-1) + message."""
    import ctypes as C
    h = _lib.lib()
    rc = h.elvis_area_downscale_u8(1, 2, 1, 10, 16, 3, 4, 0, None)
    assert rc == -1 and b"not divisible" in h.elvis_last_error()
    rc = h.elvis_recompose_u8(None, None, None, None, None, 1, 8, 8, 3, 8, 1, 1, 0, 0, None)
    assert rc == -1 and b"null" in h.elvis_last_error()
    d = _lib.ConvDesc()
    assert h.elvis_conv_packed_weight_bytes(C.byref(d)) == 0
    d.dtype, d.cin, d.cin_pitch, d.cout, d.cout_pitch, d.ksize = 1, 128, 128, 128, 128, 3
    assert h.elvis_conv_packed_weight_bytes(C.byref(d)) == 9 * 4 * 128 * 64
    d.ksize = 5
    d.n = d.h = d.w = d.ho = d.wo = 1
    d.stride = 1
    assert h.elvis_conv2d(C.byref(d), 16, None, 16, None, None, 0, None, None, 16, None, None) == -1
    assert h.elvis_conv_stats_tiles(C.byref(d)) == 0
    d.ksize, d.pad_before, d.h, d.w, d.ho, d.wo, d.n = 3, 1, 1080, 1920, 1080, 1920, 1
    # f16 / 128-channel tile: the 256-thread two-workgroups-per-CU kernel, 8 x 32 pixel tiles
    assert h.elvis_conv_stats_tiles(C.byref(d)) == 135 * 60
    buf = C.create_string_buffer(96)
    assert h.elvis_conv_kernel_name(C.byref(d), buf, len(buf)) == 0
    assert buf.value == b"conv3x3_halo_kernel<half,128,256,8,false,3,false>"
    d.prologue, d.act = 1, 2
    assert h.elvis_conv_stats_tiles(C.byref(d)) == 135 * 60
    h.elvis_conv_kernel_name(C.byref(d), buf, len(buf))
    assert buf.value == b"conv3x3_halo_kernel<half,128,256,8,true,3,true>"
    d.dtype, d.prologue, d.act = 0, 0, 0                          # f32: 512-thread kernel, 16 x 32 tiles
    assert h.elvis_conv_stats_tiles(C.byref(d)) == 68 * 60
    d.prologue = 1
    assert h.elvis_conv_stats_tiles(C.byref(d)) == 90 * 60      # 12 x 32 tiles with the fused prologue
    d.ksize, d.stride, d.pad_before = 3, 2, 0
    d.ho, d.wo = 540, 960
    h.elvis_conv_kernel_name(C.byref(d), buf, len(buf))
    assert buf.value == b"conv_igemm_kernel<float,4,4,2,2>"
