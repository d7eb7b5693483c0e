"""Tensor-level wrappers over the C ABI (device memory and streams come from torch - plumbing
only; every arithmetic op below is a hand-written HIP kernel in libelvis_amd.so).

Float activations are NHWC torch tensors of shape [n, h, w, pitch] (pitch = channel stride, a
multiple of 8, pad channels are kept at zero) accompanied by the logical channel count.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from ._lib import ConvDesc, check, lib, ptr


_NO_S2D = bool(os.environ.get("ELVIS_NO_S2D"))   # A/B switch, read once at import


def _s(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def pitch_for(c: int) -> int:
    return (c + 7) // 8 * 8


@dataclass
class Act:
    """NHWC activation: t[n,h,w,pitch], logical channels c.  `stats` (optional) holds the per-tile
    GroupNorm partial sums f32[n*tiles, c, 2] the producing conv wrote for this tensor."""
    t: torch.Tensor
    c: int
    stats: Optional[torch.Tensor] = None

    @property
    def n(self): return self.t.shape[0]
    @property
    def h(self): return self.t.shape[1]
    @property
    def w(self): return self.t.shape[2]
    @property
    def pitch(self): return self.t.shape[3]
    @property
    def dtype_code(self): return L.dtype_code(self.t.dtype)


def new_act(n, h, w, c, dtype, device, zero=None) -> Act:
    p = pitch_for(c)
    need_zero = (p != c) if zero is None else zero
    f = torch.zeros if need_zero else torch.empty
    return Act(f((n, h, w, p), dtype=dtype, device=device), c)


# ----------------------------------------------------------------------------- u8 glue
def _chk_u8(*ts):
    for t in ts:
        if t.dtype != torch.uint8 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("expected contiguous CUDA uint8 tensors")


def recompose_u8(a, b, map_i32, block, thr, out=None, map_out=None, clamp_to=0):
    """out = (map <= thr) ? a : b per block.  a,b: [n,h,w,c] u8; map: [n,by,bx] int32."""
    _chk_u8(a, b)
    if a.shape != b.shape:
        raise ValueError(f"recompose: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    n, h, w, c = a.shape
    if map_i32.dtype != torch.int32 or map_i32.dim() != 3 or map_i32.shape[0] != n:
        raise ValueError("recompose: map must be int32 [n,by,bx]")
    map_i32 = map_i32.contiguous()
    by, bx = map_i32.shape[1:]
    if out is None:
        out = torch.empty_like(a)
    check(lib().elvis_recompose_u8(ptr(a), ptr(b), ptr(map_i32), ptr(out), ptr(map_out), n, h, w, c, block, by, bx,
                                   int(thr), int(clamp_to), _s(a)), a.device)
    return out


def area_downscale_u8(src, factor, rounding=L.ROUND_CV2, out=None):
    _chk_u8(src)
    n, h, w, c = src.shape
    if h % factor or w % factor:
        raise ValueError("area_downscale_u8 needs H,W divisible by the factor")
    if out is None:
        out = torch.empty((n, h // factor, w // factor, c), dtype=torch.uint8, device=src.device)
    check(lib().elvis_area_downscale_u8(ptr(src), ptr(out), n, h, w, c, factor, rounding, _s(src)), src.device)
    return out


def blend_u8(orig, rest, map_i32, block, alpha, out=None):
    _chk_u8(orig, rest)
    n, h, w, c = orig.shape
    map_i32 = map_i32.contiguous()
    by, bx = map_i32.shape[1:]
    if out is None:
        out = torch.empty_like(orig)
    check(lib().elvis_blend_u8(ptr(orig), ptr(rest), ptr(map_i32), ptr(out), n, h, w, c, block, by, bx, float(alpha),
                               _s(orig)), orig.device)
    return out


def select_levels_u8(versions, slot_of_level, map_i32, block, out=None):
    """versions: list of [n,h,w,c] u8 tensors; slot_of_level: int32 tensor level->index (-1 = none)."""
    _chk_u8(*versions)
    n, h, w, c = versions[0].shape
    dev = versions[0].device
    ptrs = torch.tensor([v.data_ptr() for v in versions], dtype=torch.int64, device=dev)
    map_i32 = map_i32.contiguous()
    by, bx = map_i32.shape[1:]
    if out is None:
        out = torch.empty_like(versions[0])
    check(lib().elvis_select_levels_u8(ptr(ptrs), ptr(slot_of_level), slot_of_level.numel(), ptr(map_i32), ptr(out),
                                       n, h, w, c, block, by, bx, _s(out)), dev)
    return out


def tile_accumulate(acc, wsum, tile, wy, wx, wx2, y0, x0, temporal_weight):
    h, w, c = acc.shape
    th, tw = tile.shape[:2]
    check(lib().elvis_tile_accumulate_f32(ptr(acc), ptr(wsum), ptr(tile), ptr(wy), ptr(wx), ptr(wx2), h, w, y0, x0, th, tw, c,
                                          float(temporal_weight), _s(acc)), acc.device)


def tile_normalize(acc, wsum, out=None):
    h, w, c = acc.shape
    if out is None:
        out = torch.empty((h, w, c), dtype=torch.uint8, device=acc.device)
    check(lib().elvis_tile_normalize_u8(ptr(acc), ptr(wsum), ptr(out), h, w, c, _s(acc)), acc.device)
    return out


def sse_u8(a, b, mask=None):
    """Per-frame (sum of squared differences, element count) as int64 tensors."""
    _chk_u8(a, b)
    n, h, w, c = a.shape
    sse = torch.zeros(n, dtype=torch.int64, device=a.device)
    cnt = torch.zeros(n, dtype=torch.int64, device=a.device)
    check(lib().elvis_sse_u8(ptr(a), ptr(b), ptr(mask), ptr(sse), ptr(cnt), n, h, w, c, _s(a)), a.device)
    return sse, cnt


# ----------------------------------------------------------------------------- conversions
def u8_to_float(src, dtype, scale, bias, swap_rb=False, div255=False, pitch=8) -> Act:
    _chk_u8(src)
    n, h, w, c = src.shape
    if c != 3:
        raise ValueError("u8_to_float expects 3-channel frames")
    dst = torch.empty((n, h, w, pitch), dtype=dtype, device=src.device)
    check(lib().elvis_u8_to_float(ptr(src), ptr(dst), L.dtype_code(dtype), n, h, w, pitch, float(scale), float(bias),
                                  int(swap_rb), int(div255), _s(src)), src.device)
    return Act(dst, 3)


def float_to_u8(x: Act, scale, bias, mode=0, swap_rb=False, want_f32=False):
    n, h, w = x.n, x.h, x.w
    dst = torch.empty((n, h, w, 3), dtype=torch.uint8, device=x.t.device)
    f32 = torch.empty((n, h, w, 3), dtype=torch.float32, device=x.t.device) if want_f32 else None
    check(lib().elvis_float_to_u8(ptr(x.t), x.dtype_code, ptr(dst), ptr(f32), n, h, w, x.pitch, float(scale),
                                  float(bias), mode, int(swap_rb), _s(x.t)), x.t.device)
    return (dst, f32) if want_f32 else dst


# ----------------------------------------------------------------------------- model kernels
F32X3_CODE = 2   # include/elvis_amd.h ELVIS_F32X3


def x3_mfma_factor(kernel_name: str) -> int:
    """f16 MFMAs a kernel executes per algorithmic one: 3 for the planar compensated form (conv_x3p.inc: hi*hi + hi*lo +
    lo*hi on full-K fragments), 4 for the interleaved form (conv_kernels.inc mma_tile_x), 1 otherwise."""
    return 3 if "_x3p_" in kernel_name else 4 if "_x3_" in kernel_name else 1
_X3_DEFAULT = False


class x3_default:
    """`with ops.x3_default(True):` - fp32 PackedConv layers built inside run their products on the f16 matrix pipe with
    the rounding error compensated (ELVIS_F32X3) unless they say otherwise."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global _X3_DEFAULT
        self.prev, _X3_DEFAULT = _X3_DEFAULT, self.on

    def __exit__(self, *exc):
        global _X3_DEFAULT
        _X3_DEFAULT = self.prev


class PackedConv:
    """A conv/linear layer with weights packed for the implicit-GEMM kernel."""

    def __init__(self, weight_oihw: torch.Tensor, bias: Optional[torch.Tensor], dtype, device, cin: int,
                 cin2: int = 0, x3: Optional[bool] = None):
        cout, ctot, kh, kw = weight_oihw.shape
        assert kh == kw and ctot == cin + cin2
        self.cin, self.cin2, self.cout, self.ksize = cin, cin2, cout, kh
        self.dtype, self.device = dtype, device
        # A ragged output count (189, 3, 1 ...) is padded to whole 4-channel groups with zero weights and zero bias: the extra
        # channels ARE pad channels of the output activation (zero either way), and the kernels then run their branch-free
        # epilogue instead of the ragged one (per-element bias loads and 2-byte stores: ~40 k cycles per tile - the DCT slot's
        # 32 -> 189 conv went from 3.7 to 2.0 ms).  `cout` stays the logical count; `cout_k` is what the kernels see.
        self.cout_k = (cout + 3) // 4 * 4
        if self.cout_k != cout:
            weight_oihw = torch.cat([weight_oihw, weight_oihw.new_zeros((self.cout_k - cout, ctot, kh, kw))], 0)
            if bias is not None:
                bias = torch.cat([bias, bias.new_zeros(self.cout_k - cout)], 0)
        # fp32 tensors, products on the f16 matrix pipe with the rounding error compensated (ELVIS_F32X3, conv.hip
        # mma_tile_x): ~1e-6 relative instead of f16's 5e-4, at about twice the fp32 MFMA's speed
        self.x3 = bool(_X3_DEFAULT if x3 is None else x3) and dtype == torch.float32
        d = ConvDesc()
        d.dtype = L.dtype_code(dtype)
        d.n = d.h = d.w = d.ho = d.wo = 1
        d.cin, d.cin_pitch, d.cin2, d.cin2_pitch = cin, pitch_for(cin), cin2, pitch_for(cin2) if cin2 else 0
        d.cout, d.cout_pitch = self.cout_k, pitch_for(cout)
        d.ksize, d.stride = kh, 1
        if kh == 2:   # one parity of a sub-pixel upsample conv (PackedUpConv): h x w -> 2h x 2w
            d.subpixel, d.ho, d.wo = 1, 2, 2
        nbytes = lib().elvis_conv_packed_weight_bytes(C.byref(d))
        w_dev = weight_oihw.to(device=device, dtype=torch.float32).contiguous()
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
        check(lib().elvis_conv_pack_weights(C.byref(d), ptr(w_dev), ptr(self.packed), _s(self.packed)), device)
        self.packed_x3 = None
        if self.x3:   # second packing, (hi, lo) half pairs: read by the compensated kernels only (weights_for picks per call)
            d.dtype = F32X3_CODE
            nbytes = lib().elvis_conv_packed_weight_bytes(C.byref(d))   # its own size: the planar 3x3 format pads K to 32
            self.packed_x3 = torch.empty(nbytes, dtype=torch.uint8, device=device)
            check(lib().elvis_conv_pack_weights(C.byref(d), ptr(w_dev), ptr(self.packed_x3), _s(self.packed_x3)), device)
        torch.cuda.current_stream(device).synchronize()  # w_dev may be freed after return
        self.bias = None if bias is None else bias.to(device=device, dtype=torch.float32).contiguous()

    def weights_for(self, d) -> torch.Tensor:
        """Packed weights for descriptor `d`; sets d.dtype to ELVIS_F32X3 when this layer is compensated AND the shape
        has a compensated kernel, otherwise leaves the tensor's dtype code (exact fp32 MFMA / f16)."""
        if self.packed_x3 is not None:
            keep, d.dtype = d.dtype, F32X3_CODE
            if lib().elvis_conv_x3_eligible(C.byref(d)):
                return self.packed_x3
            d.dtype = keep
        return self.packed

    def __call__(self, x: Act, x2: Optional[Act] = None, *, stride=1, pad=None, upsample=False, act=0,
                 residual: Optional[Act] = None, prologue=None, out: Optional[Act] = None, ho=None, wo=None,
                 want_stats: bool = False) -> Act:
        if x.c != self.cin or (x2.c if x2 is not None else 0) != self.cin2:
            raise ValueError(f"conv: expected inputs with {self.cin}+{self.cin2} channels, got {x.c}+{x2.c if x2 else 0}")
        for t in (x, x2, residual):
            if t is not None and t.t.dtype != self.dtype:
                raise ValueError(f"conv: weights are packed for {self.dtype}, got a {t.t.dtype} tensor")
        d = ConvDesc()
        d.dtype = x.dtype_code
        d.n, d.h, d.w = x.n, x.h, x.w
        d.cin, d.cin_pitch = x.c, x.pitch
        d.cin2, d.cin2_pitch = (x2.c, x2.pitch) if x2 is not None else (0, 0)
        d.ksize, d.stride, d.upsample, d.act = self.ksize, stride, int(upsample), act
        d.pad_before = (self.ksize // 2) if pad is None else pad
        lh, lw = (x.h * 2, x.w * 2) if upsample else (x.h, x.w)
        if ho is None:
            ho = (lh + 2 * d.pad_before - self.ksize) // stride + 1
            wo = (lw + 2 * d.pad_before - self.ksize) // stride + 1
        d.ho, d.wo = ho, wo
        if out is None:   # (the conv kernels write the pad channels themselves: no zero-fill pass)
            out = new_act(x.n, ho, wo, self.cout, x.t.dtype, x.t.device, zero=False)
        d.cout, d.cout_pitch = self.cout_k, out.pitch
        pa = pb = None
        if prologue is not None:
            pa, pb = prologue
            d.prologue = 1
        stats = None
        packed = self.weights_for(d)
        tiles = lib().elvis_conv_stats_tiles(C.byref(d))
        if want_stats and tiles > 0:
            stats = torch.empty((tiles, self.cout_k, 2), dtype=torch.float32, device=x.t.device)
        out.stats = stats
        prof = CONV_PROFILER
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(lib().elvis_conv2d(C.byref(d), ptr(x.t), ptr(x2.t) if x2 is not None else 0, ptr(packed),
                                 ptr(self.bias), ptr(residual.t) if residual is not None else 0,
                                 residual.pitch if residual is not None else 0, ptr(pa), ptr(pb), ptr(out.t),
                                 ptr(stats), _s(x.t)), x.t.device)
        if prof is not None:
            e1.record()
            flops = 2.0 * self.ksize * self.ksize * (self.cin + self.cin2) * self.cout * x.n * ho * wo
            name = conv_kernel_name(d)
            if name.startswith("conv3x3_ws_kernel") and residual is None and KERNEL_PROFILER is not None:
                # the weight-stationary kernel serves narrow layers, which are HBM-bound (csrc/conv_ws.inc): priced against
                # the bytes it has to move - every input channel read once, every output channel written once
                nbytes = float(self.cin + self.cin2 + self.cout) * x.t.element_size() * x.n * ho * wo
                KERNEL_PROFILER.append((name, "hbm", nbytes, e0, e1))
            else:
                prof.append((name, flops, e0, e1))
            if CONV_SHAPES is not None:
                CONV_SHAPES.append((x.n, x.h, x.w, self.cin + self.cin2, self.cout, self.ksize, stride,
                                    prologue is not None, residual is not None))
        if stats is not None and self.cout_k != self.cout:   # padded count: the consumers read [tiles, cout, 2] rows
            out.stats = stats[:, :self.cout].contiguous()
        return out


# When set to a list, every conv launch appends (kernel_name, algorithmic_flops, start_evt, end_evt);
# the events are recorded on the stream the kernel is launched on (torch's current stream).
CONV_PROFILER = None
CONV_SHAPES = None    # optional parallel list of (n,h,w,cin,cout,ksize,stride,prologue,residual) per launch
# Same idea for the non-conv hot kernels (window attention, DCNv2): (kernel, bound "mfma"|"hbm", algorithmic
# FLOPs or bytes, start_evt, end_evt)
KERNEL_PROFILER = None


class _Timed:
    def __init__(self, name, bound, work):
        self.rec = (name, bound, work) if KERNEL_PROFILER is not None else None

    def __enter__(self):
        if self.rec:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.rec and KERNEL_PROFILER is not None:
            self.e1.record()
            KERNEL_PROFILER.append(self.rec + (self.e0, self.e1))


def conv_kernel_name(d) -> str:
    """Name of the kernel instantiation `elvis_conv2d` dispatches descriptor `d` to - the template
    name rocprofv3's kernel trace shows (queried from the library: conv.hip choose_tile / halo_two)."""
    buf = C.create_string_buffer(128)
    check(lib().elvis_conv_kernel_name(C.byref(d), buf, len(buf)), None)
    return buf.value.decode()


class PackedUpConv:
    """`conv3x3(nearest_upsample_2x(x))` as four sub-pixel 2x2 convs on the low-res grid.

    Output pixel (2y+a, 2x+b) sees, through the upsampled 3x3 window, only a 2x2 block of low-res
    pixels: rows {y-1, y} for a=0 ({y, y+1} for a=1) and likewise for columns, with the 3x3 weights
    that land on the same low-res pixel summed (rows 1+2 for a=0, rows 0+1 for a=1).  Exact in real
    arithmetic; 16 instead of 36 taps per low-res pixel (2.25x fewer FLOPs).  Weights are summed in
    fp32 at load time."""

    ROWSETS = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}

    def __init__(self, weight_oihw: torch.Tensor, bias: Optional[torch.Tensor], dtype, device, cin: int):
        cout, ctot, kh, kw = weight_oihw.shape
        assert kh == 3 and kw == 3 and ctot == cin
        self.cin, self.cout = cin, cout
        w = weight_oihw.float()
        self.par = []
        for a in (0, 1):
            for b in (0, 1):
                w2 = torch.zeros(cout, cin, 2, 2)
                for dyi, rows in enumerate(self.ROWSETS[a]):
                    for dxi, cols in enumerate(self.ROWSETS[b]):
                        w2[:, :, dyi, dxi] = sum(w[:, :, r, c] for r in rows for c in cols)
                self.par.append(PackedConv(w2, bias, dtype, device, cin))

    def __call__(self, x: Act, want_stats: bool = False, act: int = 0) -> Act:
        if x.t.dtype != self.par[0].dtype:
            raise ValueError(f"conv: weights are packed for {self.par[0].dtype}, got a {x.t.dtype} tensor")
        n, h, w = x.n, x.h, x.w
        out = new_act(n, 2 * h, 2 * w, self.cout, x.t.dtype, x.t.device, zero=False)   # the four parities write every pixel, pads included
        stats = None
        for k, conv in enumerate(self.par):
            d = ConvDesc()
            d.dtype = x.dtype_code
            d.n, d.h, d.w = n, h, w
            d.cin, d.cin_pitch = x.c, x.pitch
            d.ksize, d.stride, d.subpixel, d.act = 2, 1, 1 + k, act
            d.ho, d.wo = 2 * h, 2 * w
            d.cout, d.cout_pitch = conv.cout_k, out.pitch
            packed = conv.weights_for(d)
            tiles = lib().elvis_conv_stats_tiles(C.byref(d))
            if want_stats and stats is None:
                stats = torch.empty((4 * tiles, self.cout, 2), dtype=torch.float32, device=x.t.device)
            sp = stats[k * tiles:(k + 1) * tiles] if stats is not None else None
            prof = CONV_PROFILER
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib().elvis_conv2d(C.byref(d), ptr(x.t), 0, ptr(packed), ptr(conv.bias), 0, 0, 0, 0, ptr(out.t),
                                     ptr(sp), _s(x.t)), x.t.device)
            if prof is not None:
                e1.record()
                prof.append((conv_kernel_name(d),
                             2.0 * 4 * self.cin * self.cout * n * h * w, e0, e1))
                if CONV_SHAPES is not None:
                    CONV_SHAPES.append((n, h, w, self.cin, self.cout, 2, 1, False, False))
        if stats is not None:
            # per image the reduce expects that image's tiles contiguous: [parity][n*tiles] -> [n][4*tiles/n]
            t_img = tiles // n
            stats = stats.view(4, n, t_img, self.cout, 2).permute(1, 0, 2, 3, 4).reshape(4 * tiles, self.cout, 2).contiguous() if n > 1 else stats
        out.stats = stats
        return out


class PackedDownConv:
    """`conv3x3(pad(x, (0,1,0,1)), stride=2)` - the autoencoder's Downsample - in
    space-to-depth form: the four (row, column) phases of the full-res input are read as 4C channels of
    an ho x wo image and a 2x2 conv with the re-indexed 3x3 taps runs on the halo-tile kernel
    (W[2ry+py][2rx+px] at channel (2py+px)*C + c, tap (ry, rx); 7 of the 16 phase/tap blocks are
    zero).  Same result as the strided conv in real arithmetic; 1.78x the MFMA work of the direct form,
    at ~3.5x its rate on the generic strided kernel.  f16 (C % 32 == 0) or fp32 tensors on the compensated
    f16 MFMA (ELVIS_F32X3, C % 16 == 0); cout >= 64."""

    S2D = 5   # ELVIS_CONV_S2D

    @staticmethod
    def supported(dtype, cin: int, cout: int, x3: bool = False) -> bool:
        """f16 with whole 32-channel chunks, or fp32 tensors on the compensated f16 MFMA (16-channel chunks)."""
        return cout >= 64 and ((dtype == torch.float16 and cin % 32 == 0) or (dtype == torch.float32 and x3 and cin % 16 == 0))

    @staticmethod
    def supported_pad1(dtype, cin: int, cout: int) -> bool:
        """`conv3x3(x, stride=2, padding=1)` (the Blur / DCT slots' down convs) in the same form: f16, 32 or more outputs."""
        return dtype == torch.float16 and cin % 32 == 0 and cout >= 32 and not _NO_S2D

    def __init__(self, weight_oihw: torch.Tensor, bias: Optional[torch.Tensor], dtype, device, cin: int, pad1: bool = False):
        cout, ctot, kh, kw = weight_oihw.shape
        assert kh == 3 and kw == 3 and ctot == cin
        assert self.supported_pad1(dtype, cin, cout) if pad1 else self.supported(dtype, cin, cout, x3=True)
        self.cin, self.cout, self.device, self.dtype = cin, cout, device, dtype
        self.pad1 = bool(pad1)   # taps at phase rows y-1, y (pad 1) instead of y, y+1 (pad (0,1,0,1))
        self.code = L.F16 if dtype == torch.float16 else F32X3_CODE
        w = weight_oihw.float()
        w4 = torch.zeros(cout, 4 * cin, 2, 2)
        for dy in range(3):
            for dx in range(3):
                ry, py, rx, px = ((dy + 1) // 2, (dy + 1) % 2, (dx + 1) // 2, (dx + 1) % 2) if pad1 else (dy // 2, dy % 2, dx // 2, dx % 2)
                ph = 2 * py + px
                w4[:, ph * cin:(ph + 1) * cin, ry, rx] = w[:, :, dy, dx]
        d = self._desc(1, 2, 2, pitch_for(cin), pitch_for(cout))
        nbytes = lib().elvis_conv_packed_weight_bytes(C.byref(d))
        w_dev = w4.to(device=device).contiguous()
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
        check(lib().elvis_conv_pack_weights(C.byref(d), ptr(w_dev), ptr(self.packed), _s(self.packed)), device)
        torch.cuda.current_stream(device).synchronize()
        self.bias = None if bias is None else bias.to(device=device, dtype=torch.float32).contiguous()

    def _desc(self, n, h, w, cin_pitch, cout_pitch):
        d = ConvDesc()
        d.dtype = self.code
        d.n, d.h, d.w, d.ho, d.wo = n, h, w, h // 2, w // 2
        d.cin, d.cin_pitch = 4 * self.cin, cin_pitch
        d.cout, d.cout_pitch = self.cout, cout_pitch
        d.ksize, d.stride, d.subpixel = 2, 1, self.S2D
        d.pad_before = 1 if self.pad1 else 0
        return d

    def __call__(self, x: Act, want_stats: bool = False, act: int = 0) -> Act:
        if x.c != self.cin or x.h % 2 or x.w % 2 or x.t.dtype != self.dtype:
            raise ValueError(f"downsample conv: expected {self.dtype}, {self.cin} channels and even H, W; got {x.t.dtype}, {x.c}, {x.h}x{x.w}")
        out = new_act(x.n, x.h // 2, x.w // 2, self.cout, x.t.dtype, x.t.device, zero=False)
        d = self._desc(x.n, x.h, x.w, x.pitch, out.pitch)
        d.act = act
        tiles = lib().elvis_conv_stats_tiles(C.byref(d))
        stats = torch.empty((tiles, self.cout, 2), dtype=torch.float32, device=x.t.device) if want_stats and tiles > 0 else None
        prof = CONV_PROFILER
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(lib().elvis_conv2d(C.byref(d), ptr(x.t), 0, ptr(self.packed), ptr(self.bias), 0, 0, 0, 0, ptr(out.t),
                                 ptr(stats), _s(x.t)), x.t.device)
        if prof is not None:
            e1.record()
            prof.append((conv_kernel_name(d), 2.0 * 9 * self.cin * self.cout * x.n * out.h * out.w, e0, e1))   # algorithmic FLOPs of the 3x3/s2 conv
            if CONV_SHAPES is not None:
                CONV_SHAPES.append((x.n, x.h, x.w, self.cin, self.cout, 3, 2, False, False))
        out.stats = stats
        return out


def groupnorm_affine(xs, gamma, beta, groups, eps, scale=None, shift=None):
    """GroupNorm statistics over the (virtual) channel concat of `xs` -> per-(n,c) affine (pa, pb)
    such that GN(x)*(1+scale)+shift == x*pa + pb."""
    x0 = xs[0]
    n, hw = x0.n, x0.h * x0.w
    ctot = sum(x.c for x in xs)
    dev = x0.t.device
    sums = torch.empty((n, ctot, 2), dtype=torch.float64, device=dev)   # every entry is written below
    off = 0
    for x in xs:
        if x.stats is not None:   # fused: the producing conv already wrote per-tile partial sums
            check(lib().elvis_gn_partials_to_sums(ptr(x.stats), x.stats.shape[0] // n, n, x.c, ptr(sums), ctot, off,
                                                  _s(x.t)), dev)
        else:
            nws = lib().elvis_groupnorm_workspace_floats(x.dtype_code, n, hw, x.c)
            ws = torch.empty(nws, dtype=torch.float32, device=dev)
            check(lib().elvis_groupnorm_sums(ptr(x.t), x.dtype_code, n, hw, x.c, x.pitch, ptr(sums), ctot, off,
                                             ptr(ws), _s(x.t)), dev)
        off += x.c
    pa = torch.empty((n, ctot), dtype=torch.float32, device=dev)
    pb = torch.empty((n, ctot), dtype=torch.float32, device=dev)
    check(lib().elvis_groupnorm_affine(ptr(sums), ptr(gamma), ptr(beta), ptr(scale), ptr(shift), ptr(pa), ptr(pb), n,
                                       hw, ctot, groups, float(eps), _s(x0.t)), dev)
    return pa, pb


def affine_act(x: Act, pa, pb, act=2, out: Optional[Act] = None) -> Act:
    if out is None:
        out = new_act(x.n, x.h, x.w, x.c, x.t.dtype, x.t.device, zero=False)
    check(lib().elvis_affine_act(ptr(x.t), ptr(out.t), x.dtype_code, x.n, x.h * x.w, x.c, x.pitch, out.pitch, ptr(pa),
                                 ptr(pb), act, _s(x.t)), x.t.device)
    return out


def layernorm(x: Act, gamma, beta, eps=1e-5, out: Optional[Act] = None) -> Act:
    if out is None:
        out = new_act(x.n, x.h, x.w, x.c, x.t.dtype, x.t.device, zero=False)
    check(lib().elvis_layernorm(ptr(x.t), ptr(out.t), x.dtype_code, x.n * x.h * x.w, x.c, x.pitch, out.pitch,
                                ptr(gamma), ptr(beta), float(eps), _s(x.t)), x.t.device)
    return out


def window_attention(qkv: Act, heads, head_dim, ws, shift, bias_table, scale) -> Act:
    out = new_act(qkv.n, qkv.h, qkv.w, heads * head_dim, qkv.t.dtype, qkv.t.device, zero=False)
    # 4*64^2*32 FLOP per window-head over ~16 KB of q, k, v and output = 32 FLOP/B against a machine balance of ~312:
    # HBM-bound.  Algorithmic bytes = the qkv tensor read once + the output written once.
    with _Timed("window_attention", "hbm", 4.0 * heads * head_dim * qkv.t.element_size() * qkv.n * qkv.h * qkv.w):
        check(lib().elvis_window_attention(ptr(qkv.t), ptr(out.t), qkv.dtype_code, qkv.n, qkv.h, qkv.w, heads, head_dim,
                                           ws, shift, qkv.pitch, out.pitch, ptr(bias_table), float(scale), _s(qkv.t)),
              qkv.t.device)
    return out


SWIN_FUSED_CHANNELS = (64, 128, 192, 256)


class SwinFused:
    """LayerNorm + one or two token-wise linear layers in ONE kernel (csrc/swin.hip), f16 tensors:

      SwinFused(norm_w, norm_b, fc1_w, fc1_b, fc2_w, fc2_b)  ->  __call__(x) = x + fc2(GELU(fc1(LN(x))))   (the Swin MLP)
      SwinFused(norm_w, norm_b, w, b)                        ->  __call__(x) = w . LN(x) + b               (LN + qkv)
      SwinFused(..., fc2_w, fc2_b, proj_w=, proj_b=)         ->  __call__(a, y): y' = y + proj(a); y' + fc2(GELU(fc1(LN(y'))))
                                                                 (the attention output projection folded in front of the MLP)

    `supported(c, n1)` says whether a shape has a fused kernel (channels 64 / 128 / 192 / 256, outputs a multiple of 64);
    callers keep the separate layernorm + 1x1-conv kernels for everything else (fp32 / compensated modes too)."""

    @staticmethod
    def supported(dtype, c: int, n1: int) -> bool:
        return dtype == torch.float16 and c in SWIN_FUSED_CHANNELS and n1 % 64 == 0

    @staticmethod
    def proj_pays(c: int) -> bool:
        """Folding the projection in front of the MLP, against projection + MLP as two kernels (tools/swin_bench.py, us per
        launch on the largest level): 559 vs ~890 at C = 64 (1080p), 434 vs ~570 at 128, 328 vs ~380 at 256.  At C = 192 the
        combined kernel sits at 256 VGPRs (96 spilled around its LayerNorm) and cannot take the staggered MLP loop: over the
        SinSR step's 216 launches it averages 990 us against 769 (MLP, lockstep; ~720 staggered) + 195 (projection) - not there."""
        return c != 192

    def __init__(self, norm_w, norm_b, w1, b1, w2=None, b2=None, *, proj_w=None, proj_b=None, device, eps: float = 1e-5):
        n1, c = w1.shape[0], w1.shape[1]
        if not self.supported(torch.float16, c, n1):
            raise ValueError(f"no fused Swin kernel for {c} channels / {n1} outputs")
        self.c, self.n1, self.mlp, self.eps, self.device = c, n1, w2 is not None, float(eps), device
        self.proj = proj_w is not None
        if self.proj and not self.mlp:
            raise ValueError("the projection is fused in front of an MLP only")
        self.mode = 2 if self.proj else int(self.mlp)
        f32 = dict(device=device, dtype=torch.float32)
        w1d = w1.reshape(n1, c).to(**f32).contiguous()
        w2d = w2.reshape(c, n1).to(**f32).contiguous() if self.mlp else None
        self.packed = torch.empty(lib().elvis_swin_packed_bytes(c, n1, self.mode), dtype=torch.uint8, device=device)
        if self.proj:
            wpd = proj_w.reshape(c, c).to(**f32).contiguous()
            check(lib().elvis_swin_pack_proj_mlp(ptr(wpd), ptr(w1d), ptr(w2d), ptr(self.packed), c, n1, _s(self.packed)), device)
            self.bp = proj_b.to(**f32).contiguous()
        else:
            check(lib().elvis_swin_pack_weights(ptr(w1d), ptr(w2d), ptr(self.packed), c, n1, int(self.mlp), _s(self.packed)), device)
        torch.cuda.current_stream(device).synchronize()   # the fp32 copies may be freed after return
        self.b1 = b1.to(**f32).contiguous()
        self.b2 = b2.to(**f32).contiguous() if self.mlp else None
        self.gamma, self.beta = norm_w.to(**f32).contiguous(), norm_b.to(**f32).contiguous()

    def __call__(self, x: Act, y: Optional[Act] = None) -> Act:
        if x.c != self.c or x.t.dtype != torch.float16:
            raise ValueError(f"fused Swin block: expected f16 with {self.c} channels, got {x.t.dtype} with {x.c}")
        if self.proj != (y is not None) or (y is not None and (y.c != self.c or y.t.dtype != torch.float16 or y.t.shape[:3] != x.t.shape[:3])):
            raise ValueError("fused Swin block: the projection form takes (attention output, residual stream) of one shape")
        tokens = x.n * x.h * x.w
        cout = self.c if self.mlp else self.n1
        out = new_act(x.n, x.h, x.w, cout, torch.float16, x.t.device, zero=False)
        if self.proj:
            with _Timed("swin_proj_mlp", "mfma", (4.0 * self.c * self.n1 + 2.0 * self.c * self.c) * tokens):
                check(lib().elvis_swin_proj_mlp(ptr(x.t), ptr(y.t), ptr(out.t), ptr(self.packed), ptr(self.bp), ptr(self.b1), ptr(self.b2),
                                                ptr(self.gamma), ptr(self.beta), tokens, self.c, self.n1, x.pitch, y.pitch, out.pitch,
                                                self.eps, _s(x.t)), x.t.device)
        elif self.mlp:
            with _Timed("swin_mlp", "mfma", 4.0 * self.c * self.n1 * tokens):
                check(lib().elvis_swin_mlp(ptr(x.t), ptr(out.t), ptr(self.packed), ptr(self.b1), ptr(self.b2), ptr(self.gamma),
                                           ptr(self.beta), tokens, self.c, self.n1, x.pitch, out.pitch, self.eps, _s(x.t)), x.t.device)
        else:
            with _Timed("swin_ln_linear", "hbm", 2.0 * (self.c + self.n1) * tokens):
                check(lib().elvis_swin_ln_linear(ptr(x.t), ptr(out.t), ptr(self.packed), ptr(self.b1), ptr(self.gamma), ptr(self.beta),
                                                 tokens, self.c, self.n1, x.pitch, out.pitch, self.eps, _s(x.t)), x.t.device)
        return out


def bicubic_upsample(x: Act, sf: int) -> Act:
    out = new_act(x.n, x.h * sf, x.w * sf, x.c, x.t.dtype, x.t.device, zero=False)
    check(lib().elvis_bicubic_upsample(ptr(x.t), ptr(out.t), x.dtype_code, x.n, x.h, x.w, x.c, x.pitch, out.pitch, sf,
                                       _s(x.t)), x.t.device)
    return out


def vq_nearest(z: Act, codebook: torch.Tensor, want_idx=False):
    out = new_act(z.n, z.h, z.w, z.c, z.t.dtype, z.t.device, zero=False)
    idx = torch.empty((z.n, z.h, z.w), dtype=torch.int32, device=z.t.device) if want_idx else None
    check(lib().elvis_vq_nearest(ptr(z.t), ptr(out.t), ptr(idx), z.dtype_code, z.n * z.h * z.w, z.c, z.pitch,
                                 out.pitch, ptr(codebook), codebook.shape[0], _s(z.t)), z.t.device)
    return (out, idx) if want_idx else out


def pad_reflect_axpy(x: Act, hp, wp, out: Act, ch_offset=0, mul=1.0, add=None, add_mul=0.0):
    check(lib().elvis_pad_reflect_axpy(ptr(x.t), ptr(out.t), x.dtype_code, x.n, x.h, x.w, x.c, x.pitch, hp, wp,
                                       out.pitch, ch_offset, float(mul), ptr(add), float(add_mul), _s(x.t)),
          x.t.device)
    return out


def crop_copy(x: Act, h, w) -> Act:
    out = new_act(x.n, h, w, x.c, x.t.dtype, x.t.device, zero=False)
    check(lib().elvis_crop_copy(ptr(x.t), ptr(out.t), x.dtype_code, x.n, x.h, x.w, x.pitch, h, w, x.c, out.pitch,
                                _s(x.t)), x.t.device)
    return out


def convert_act(x: Act, dtype) -> Act:
    """Same tensor in another storage dtype (f16 <-> f32).  Widening keeps the producer's GroupNorm
    partial sums (the values are unchanged); narrowing drops them (the stored values moved)."""
    if x.t.dtype == dtype:
        return x
    out = torch.empty(x.t.shape, dtype=dtype, device=x.t.device)
    check(lib().elvis_convert_act(ptr(x.t), x.dtype_code, ptr(out), L.dtype_code(dtype), x.n * x.h * x.w, x.pitch,
                                  _s(x.t)), x.t.device)
    return Act(out, x.c, x.stats if dtype == torch.float32 else None)


# ----------------------------------------------------------------------------- DCT-slot kernels
def dcnv2(x: Act, om: Act, weight: torch.Tensor, bias: Optional[torch.Tensor], groups: int, cout: int,
          mask_sigmoid: bool = True, act: int = 0) -> Act:
    """Modulated deformable 3x3 conv.  om: offsets (18*G) then masks (9*G) channels; weight
    [cout, cin, 3, 3] tensor already in the activation dtype on the device."""
    out = new_act(x.n, x.h, x.w, cout, x.t.dtype, x.t.device)
    es = x.t.element_size()
    # algorithmic HBM bytes per output pixel: the 27*G offset/mask channels, the input channels (each read once:
    # the 36 bilinear corner reads per channel hit the LDS-resident input window), the output channels
    px_bytes = (27 * groups + x.c + cout) * es
    with _Timed("dcnv2", "hbm", float(px_bytes) * x.n * x.h * x.w):
        check(lib().elvis_dcnv2(ptr(x.t), ptr(om.t), ptr(weight), ptr(bias), ptr(out.t), x.dtype_code, x.n, x.h, x.w, x.c,
                                x.pitch, groups, om.pitch, int(mask_sigmoid), cout, out.pitch, act, _s(x.t)), x.t.device)
    return out


def temporal_stack(frames_u8: torch.Tensor, f0: int, nsel: int, radius: int, dtype) -> Act:
    """[F,H,W,3] u8 -> Act [(nsel*3), H, W, pitch]: the 2R+1 temporal window of every colour plane."""
    _chk_u8(frames_u8)
    nf, h, w, _ = frames_u8.shape
    t = 2 * radius + 1
    out = torch.empty((nsel * 3, h, w, pitch_for(t)), dtype=dtype, device=frames_u8.device)
    check(lib().elvis_temporal_stack(ptr(frames_u8), ptr(out), L.dtype_code(dtype), nf, f0, nsel, h, w, radius,
                                     out.shape[3], _s(out)), frames_u8.device)
    return Act(out, t)


def plane_merge(frames_u8: torch.Tensor, residual: Act, f0: int, nsel: int, out=None) -> torch.Tensor:
    _chk_u8(frames_u8)
    _, h, w, _ = frames_u8.shape
    if out is None:
        out = torch.empty((nsel, h, w, 3), dtype=torch.uint8, device=frames_u8.device)
    elif tuple(out.shape) != (nsel, h, w, 3) or not out.is_contiguous() or out.dtype != torch.uint8:
        raise ValueError("plane_merge: `out` must be a contiguous uint8 [nsel,H,W,3] tensor")
    check(lib().elvis_plane_merge(ptr(frames_u8), ptr(residual.t), ptr(out), residual.dtype_code, f0, nsel, h, w,
                                  residual.pitch, _s(out)), frames_u8.device)
    return out
