"""GPU parity tests of the individual HIP kernels (through the C ABI) against PyTorch-CPU fp32
restatements of the same op.  Tolerances: fp32 mode 1e-4 abs on O(1) activations (the path bar is
1e-3 max-abs end to end, BASELINE.json); f16 mode 2e-2 (f16 inputs/weights, fp32 accumulate)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _act(x_nchw, dtype, dev):
    from elvis_amd import ops
    n, c, h, w = x_nchw.shape
    a = ops.new_act(n, h, w, c, dtype, dev, zero=True)
    a.t[..., :c] = x_nchw.permute(0, 2, 3, 1).to(dev, dtype)
    return a


def _nchw(a):
    return a.t[..., :a.c].float().cpu().permute(0, 3, 1, 2).contiguous()


TOL = {torch.float32: 2e-4, torch.float16: 3e-2}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [
    # cin, cout, k, stride, pad, upsample, h, w
    (32, 64, 3, 1, 1, False, 17, 23),
    (3, 128, 3, 1, 1, False, 16, 16),
    (128, 3, 3, 1, 1, False, 19, 21),
    (160, 160, 3, 1, 1, False, 16, 24),
    (64, 64, 3, 2, 1, False, 16, 16),
    (64, 64, 3, 2, 0, False, 16, 18),     # AE downsample: pad (0,1,0,1)
    (64, 32, 3, 1, 1, True, 9, 11),       # nearest-2x fused
    (192, 576, 1, 1, 0, False, 8, 16),    # linear
    (48, 200, 1, 1, 0, False, 5, 7),
])
def test_conv(gpu_device, dtype, cfg):
    from elvis_amd import ops
    cin, cout, k, stride, pad, ups, h, w = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    conv = ops.PackedConv(wt, b, dtype, gpu_device, cin)
    xa = _act(x, dtype, gpu_device)
    xr = x.to(dtype).float()
    wr = wt.to(dtype).float()
    if ups:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    if stride == 2 and pad == 0:
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, b, stride=2)
        y = conv(xa, stride=2, pad=0, ho=h // 2, wo=w // 2)
    else:
        ref = F.conv2d(xr, wr, b, stride=stride, padding=pad)
        y = conv(xa, stride=stride, pad=pad, upsample=ups)
    got = _nchw(y)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_conv_concat_residual_act_prologue(gpu_device, dtype):
    from elvis_amd import ops
    g = torch.Generator().manual_seed(2)
    c1, c2, cout, h, w = 64, 32, 96, 12, 20
    x1, x2 = torch.randn(1, c1, h, w, generator=g), torch.randn(1, c2, h, w, generator=g)
    wt = torch.randn(cout, c1 + c2, 3, 3, generator=g) / math.sqrt((c1 + c2) * 9)
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(1, cout, h, w, generator=g)
    pa = torch.rand(1, c1 + c2, generator=g) + 0.5
    pb = torch.randn(1, c1 + c2, generator=g) * 0.2
    conv = ops.PackedConv(wt, b, dtype, gpu_device, c1, c2)
    a1, a2, ar = _act(x1, dtype, gpu_device), _act(x2, dtype, gpu_device), _act(res, dtype, gpu_device)
    xcat = torch.cat([x1, x2], 1).to(dtype).float()
    wr = wt.to(dtype).float()
    # concat + gelu + residual
    y = conv(a1, a2, act=1, residual=ar)
    ref = F.gelu(F.conv2d(xcat, wr, b, padding=1)) + res.to(dtype).float()
    assert (_nchw(y) - ref).abs().max().item() < TOL[dtype]
    # fused GroupNorm-affine + SiLU prologue
    y = conv(a1, a2, prologue=(pa.to(gpu_device), pb.to(gpu_device)))
    xin = F.silu(xcat * pa[:, :, None, None] + pb[:, :, None, None])
    if dtype == torch.float16:
        xin = xin.half().float()
    ref = F.conv2d(xin, wr, b, padding=1)
    assert (_nchw(y) - ref).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [
    # c1, c2, cout, h, w, upsample   (3x3 / stride 1 / pad 1 -> LDS halo-tile kernel)
    (128, 0, 128, 40, 70, False),     # several 8x32 tiles, ragged right/bottom edges
    (64, 32, 192, 19, 45, False),     # virtual concat, TCO=64 tile with 3 co tiles
    (160, 0, 160, 16, 64, False),     # cout padded 160 -> 192
    (64, 0, 128, 13, 21, True),       # fused nearest-2x upsample
    (3, 0, 64, 9, 33, False),         # cin < one K chunk
    (128, 0, 3, 21, 40, False),       # conv_out-style: 16-channel output tile, cout = 3
    (64, 0, 32, 17, 35, False),       # 32-channel output tile
    (128, 0, 128, 64, 1920, False),   # the dominant instantiation at full 1080p width: XCD remap, 60 tiles per row
    (64, 0, 64, 72, 1920, False),     # >= 128 K pixels per image: the 16-row ("tall") 64-channel tile
    # whole 32-pixel tile columns + whole cout tiles
    (128, 0, 128, 21, 64, False),     # ragged bottom edge only
    (256, 0, 256, 12, 32, False),     # two cout tiles, eight K chunks, one tile column
    (32, 0, 128, 16, 64, False),      # a single K chunk
    (64, 0, 64, 20, 96, False),       # 64-channel tile, whole tile columns
    (32, 0, 512, 40, 96, False),      # four cout tiles: the grouped block walk (decode_block, co_group = 2), 30 pixel tiles over 8 XCDs
    (32, 0, 320, 17, 64, False),      # five 64-channel cout tiles: the last group is narrower
])
def test_conv_halo_fused(gpu_device, dtype, cfg):
    """The halo kernel with everything fused: GN-affine+SiLU prologue, bias, residual, and the
    per-tile GroupNorm partial sums of its output."""
    from elvis_amd import ops
    c1, c2, cout, h, w, ups = cfg
    g = torch.Generator().manual_seed(12)
    n = 2
    x1 = torch.randn(n, c1, h, w, generator=g)
    x2 = torch.randn(n, c2, h, w, generator=g) if c2 else None
    ctot = c1 + c2
    wt = torch.randn(cout, ctot, 3, 3, generator=g) / math.sqrt(ctot * 9)
    b = torch.randn(cout, generator=g) * 0.1
    ho, wo = (2 * h, 2 * w) if ups else (h, w)
    res = torch.randn(n, cout, ho, wo, generator=g)
    pa = torch.rand(n, ctot, generator=g) + 0.5
    pb = torch.randn(n, ctot, generator=g) * 0.2
    conv = ops.PackedConv(wt, b, dtype, gpu_device, c1, c2)
    a1 = _act(x1, dtype, gpu_device)
    a2 = _act(x2, dtype, gpu_device) if c2 else None
    ar = _act(res, dtype, gpu_device)
    y = conv(a1, a2, upsample=ups, prologue=(pa.to(gpu_device), pb.to(gpu_device)), residual=ar, want_stats=True)
    assert y.stats is not None
    xcat = (torch.cat([x1, x2], 1) if c2 else x1).to(dtype).float()
    xin = F.silu(xcat * pa[:, :, None, None] + pb[:, :, None, None])
    if dtype == torch.float16:
        xin = xin.half().float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, wt.to(dtype).float(), b, padding=1) + res.to(dtype).float()
    got = _nchw(y)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL[dtype]
    # fused statistics == sums of the stored tensor
    sums = torch.zeros((n, cout, 2), dtype=torch.float64, device=gpu_device)
    from elvis_amd._lib import lib, check, ptr
    check(lib().elvis_gn_partials_to_sums(ptr(y.stats), y.stats.shape[0] // n, n, cout, ptr(sums), cout, 0,
                                          torch.cuda.current_stream().cuda_stream))
    s = sums.cpu()
    gd = got.double()
    assert torch.allclose(s[:, :, 0], gd.sum((2, 3)), rtol=1e-5, atol=1e-2)
    assert torch.allclose(s[:, :, 1], (gd * gd).sum((2, 3)), rtol=1e-5, atol=1e-2)
    # and the same conv on the generic implicit-GEMM kernel (the run-time form of ELVIS_NO_HALO=1)
    d = ops.ConvDesc()
    d.dtype, d.n, d.h, d.w, d.ho, d.wo = ops.L.dtype_code(dtype), n, h, w, ho, wo
    d.cin, d.cin_pitch, d.cin2, d.cin2_pitch = c1, a1.pitch, c2, (a2.pitch if c2 else 0)
    d.cout, d.cout_pitch, d.ksize, d.stride, d.pad_before, d.upsample, d.prologue = conv.cout_k, y.pitch, 3, 1, 1, int(ups), 1
    assert ops.conv_kernel_name(d).startswith("conv3x3_halo")
    check(lib().elvis_conv_debug_set(b"no_halo", 1))
    try:
        assert ops.conv_kernel_name(d).startswith("conv_igemm")
        y0 = conv(a1, a2, upsample=ups, prologue=(pa.to(gpu_device), pb.to(gpu_device)), residual=ar)
    finally:
        check(lib().elvis_conv_debug_set(b"no_halo", -1))
    assert (_nchw(y0) - got).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [
    # c1, c2, cout, h, w, act   (1x1 / linear through the tile-staging kernel: n*h*w >= 4096)
    (192, 0, 576, 40, 72, 0),
    (192, 0, 768, 33, 70, 1),     # GELU epilogue
    (256, 0, 128, 48, 64, 0),     # nin_shortcut
    (320, 160, 160, 40, 64, 0),   # skip_connection on a virtual concat, cout padded 160 -> 192
    (768, 0, 192, 32, 64, 0),
    (3, 0, 3, 64, 64, 0),         # quant_conv-style
])
def test_conv_1x1_tiled(gpu_device, dtype, cfg):
    from elvis_amd import ops
    c1, c2, cout, h, w, act = cfg
    g = torch.Generator().manual_seed(13)
    n = 2
    x1 = torch.randn(n, c1, h, w, generator=g)
    x2 = torch.randn(n, c2, h, w, generator=g) if c2 else None
    ctot = c1 + c2
    wt = torch.randn(cout, ctot, 1, 1, generator=g) / math.sqrt(ctot)
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g)
    conv = ops.PackedConv(wt, b, dtype, gpu_device, c1, c2)
    a1 = _act(x1, dtype, gpu_device)
    a2 = _act(x2, dtype, gpu_device) if c2 else None
    ar = _act(res, dtype, gpu_device)
    y = conv(a1, a2, act=act, residual=ar, want_stats=True)
    assert y.stats is not None
    xcat = (torch.cat([x1, x2], 1) if c2 else x1).to(dtype).float()
    ref = F.conv2d(xcat, wt.to(dtype).float(), b)
    if act == 1:
        ref = F.gelu(ref)
    ref = ref + res.to(dtype).float()
    got = _nchw(y)
    assert (got - ref).abs().max().item() < TOL[dtype]
    sums = torch.zeros((n, cout, 2), dtype=torch.float64, device=gpu_device)
    from elvis_amd._lib import lib, check, ptr
    check(lib().elvis_gn_partials_to_sums(ptr(y.stats), y.stats.shape[0] // n, n, cout, ptr(sums), cout, 0,
                                          torch.cuda.current_stream().cuda_stream))
    assert torch.allclose(sums.cpu()[:, :, 0], got.double().sum((2, 3)), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [(128, 128, 20, 37), (64, 192, 16, 32), (320, 320, 9, 40), (512, 512, 33, 35)])
def test_subpixel_upsample_conv(gpu_device, dtype, cfg):
    """nearest-2x + 3x3 conv as four 2x2 sub-pixel convs == the direct formulation."""
    from elvis_amd import ops
    cin, cout, h, w = cfg
    g = torch.Generator().manual_seed(14)
    n = 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    up = ops.PackedUpConv(wt, b, dtype, gpu_device, cin)
    xa = _act(x, dtype, gpu_device)
    y = up(xa, want_stats=True)
    ref = F.conv2d(F.interpolate(x.to(dtype).float(), scale_factor=2, mode="nearest"), wt.to(dtype).float(), b, padding=1)
    got = _nchw(y)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL[dtype] * 1.5
    sums = torch.zeros((n, cout, 2), dtype=torch.float64, device=gpu_device)
    from elvis_amd._lib import lib, check, ptr
    check(lib().elvis_gn_partials_to_sums(ptr(y.stats), y.stats.shape[0] // n, n, cout, ptr(sums), cout, 0,
                                          torch.cuda.current_stream().cuda_stream))
    assert torch.allclose(sums.cpu()[:, :, 0], got.double().sum((2, 3)), rtol=1e-5, atol=2e-2)
    assert torch.allclose(sums.cpu()[:, :, 1], (got.double() ** 2).sum((2, 3)), rtol=1e-5, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("c,groups", [(64, 32), (160, 32), (96, 8)])
def test_groupnorm_silu(gpu_device, dtype, c, groups):
    from elvis_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, c, 21, 19, generator=g) * 1.7 + 0.3
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    scale, shift = torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3
    xa = _act(x, dtype, gpu_device)
    dv = lambda t: t.to(gpu_device)
    pa, pb = ops.groupnorm_affine([xa], dv(gamma), dv(beta), groups, 1e-5, scale=dv(scale), shift=dv(shift))
    y = ops.affine_act(xa, pa, pb, act=2)
    xr = x.to(dtype).float()
    ref = F.silu(F.group_norm(xr, groups, gamma, beta, 1e-5) * (1 + scale[None, :, None, None]) + shift[None, :, None, None])
    assert (_nchw(y) - ref).abs().max().item() < (5e-5 if dtype == torch.float32 else 2e-2)


def test_groupnorm_virtual_concat(gpu_device):
    from elvis_amd import ops
    g = torch.Generator().manual_seed(4)
    x1, x2 = torch.randn(1, 64, 8, 8, generator=g), torch.randn(1, 32, 8, 8, generator=g) * 2
    gamma, beta = torch.randn(96, generator=g), torch.randn(96, generator=g)
    a1, a2 = _act(x1, torch.float32, gpu_device), _act(x2, torch.float32, gpu_device)
    pa, pb = ops.groupnorm_affine([a1, a2], gamma.to(gpu_device), beta.to(gpu_device), 32, 1e-6)
    xc = torch.cat([x1, x2], 1)
    ref = F.group_norm(xc, 32, gamma, beta, 1e-6)
    got = xc * pa.cpu()[:, :, None, None] + pb.cpu()[:, :, None, None]
    assert (got - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("c", [192, 64])
def test_layernorm(gpu_device, dtype, c):
    from elvis_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, c, 9, 13, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    xa = _act(x, dtype, gpu_device)
    y = ops.layernorm(xa, gamma.to(gpu_device), beta.to(gpu_device), 1e-5)
    ref = F.layer_norm(x.to(dtype).float().permute(0, 2, 3, 1), (c,), gamma, beta, 1e-5).permute(0, 3, 1, 2)
    assert (_nchw(y) - ref).abs().max().item() < (5e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shift", [0, 4])
def test_window_attention(gpu_device, dtype, shift):
    from elvis_amd import ops
    from elvis_amd.weights import relative_position_index
    from oracle import sinsr_ref as R
    g = torch.Generator().manual_seed(6)
    heads, hd, ws, h, w = 3, 32, 8, 16, 24
    E = heads * hd
    qkv = torch.randn(1, 3 * E, h, w, generator=g)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    qa = _act(qkv, dtype, gpu_device)
    out = ops.window_attention(qa, heads, hd, ws, shift, table.to(gpu_device), hd ** -0.5)
    # torch reference (same math as oracle.sinsr_ref.swin_block's attention core)
    t = qkv.to(dtype).float().permute(0, 2, 3, 1)
    if shift:
        t = torch.roll(t, (-shift, -shift), (1, 2))
    win = R.window_partition(t, ws)
    nw, n = win.shape[0], ws * ws
    q, k, v = win.view(nw, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    bias = table[relative_position_index(ws).view(-1)].view(n, n, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if shift:
        m = R.shift_mask(h, w, ws, shift)
        attn = (attn.view(1, m.shape[0], heads, n, n) + m[None, :, None]).view(-1, heads, n, n)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(nw, n, E)
    o = R.window_reverse(o, ws, h, w)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    ref = o.permute(0, 3, 1, 2)
    assert (_nchw(out) - ref).abs().max().item() < (5e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_bicubic(gpu_device, dtype):
    from elvis_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 3, 13, 17, generator=g) * 2 - 1
    y = ops.bicubic_upsample(_act(x, dtype, gpu_device), 4)
    ref = F.interpolate(x.to(dtype).float(), scale_factor=4, mode="bicubic", align_corners=False)
    assert (_nchw(y) - ref).abs().max().item() < (2e-5 if dtype == torch.float32 else 3e-3)


def test_vq_nearest_exact(gpu_device):
    from elvis_amd import ops
    from oracle import sinsr_ref as R
    g = torch.Generator().manual_seed(8)
    cb = torch.rand(3000, 3, generator=g) * 3 - 1.5
    z = torch.randn(1, 3, 30, 41, generator=g)
    z[0, :, 0, 0] = cb[17]          # exact hit
    zq, idx = ops.vq_nearest(_act(z, torch.float32, gpu_device), cb.to(gpu_device), want_idx=True)
    ref_q, ref_idx = R.vq_quantize({"ae.quantize.embedding.weight": cb}, z)
    assert torch.equal(idx.cpu().long(), ref_idx)
    assert torch.equal(_nchw(zq), ref_q)


def test_pad_reflect_axpy_and_crop(gpu_device):
    from elvis_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 3, 10, 13, generator=g)
    noise = torch.randn(1, 3, 16, 16, generator=g)
    out = ops.new_act(1, 16, 16, 6, torch.float32, gpu_device, zero=True)
    xa = _act(x, torch.float32, gpu_device)
    ops.pad_reflect_axpy(xa, 16, 16, out, 0, mul=0.5, add=noise.to(gpu_device), add_mul=2.0)
    ops.pad_reflect_axpy(xa, 16, 16, out, 3, mul=1.0)
    xp = F.pad(x, (0, 3, 0, 6), mode="reflect")
    ref = torch.cat([xp * 0.5 + 2.0 * noise, xp], 1)
    assert (_nchw(out) - ref).abs().max().item() < 1e-6
    assert float(out.t[..., 6:].abs().max()) == 0.0
    c = ops.crop_copy(out, 10, 13)
    assert torch.equal(_nchw(c), _nchw(out)[:, :, :10, :13])


def test_u8_float_roundtrip(gpu_device):
    from elvis_amd import ops
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.integers(0, 256, size=(2, 9, 11, 3), dtype=np.uint8)).to(gpu_device)
    a = ops.u8_to_float(img, torch.float32, 2.0, -1.0, swap_rb=True, div255=True)
    ref = (img.cpu().float().flip(-1) / 255.0) * 2.0 - 1.0
    assert torch.equal(a.t[..., :3].cpu(), ref)
    assert float(a.t[..., 3:].abs().max()) == 0.0
    back, f32 = ops.float_to_u8(a, 0.5, 0.5, mode=0, swap_rb=True, want_f32=True)
    assert torch.equal(back.cpu(), img.cpu())


@pytest.mark.parametrize("cfg", [(128, 128, 20, 36), (64, 192, 18, 70), (256, 256, 34, 64)])
def test_downsample_conv_space_to_depth(gpu_device, cfg):
    """pad (0,1,0,1) + 3x3 / stride 2 in space-to-depth form (2x2 conv over the four input phases) == the
    direct strided conv; fused GroupNorm statistics == sums of the stored tensor."""
    from elvis_amd import ops
    cin, cout, h, w = cfg
    g = torch.Generator().manual_seed(21)
    n = 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    conv = ops.PackedDownConv(wt, b, torch.float16, gpu_device, cin)
    y = conv(_act(x, torch.float16, gpu_device), want_stats=True)
    ref = F.conv2d(F.pad(x.half().float(), (0, 1, 0, 1)), wt.half().float(), b, stride=2)
    got = _nchw(y)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL[torch.float16]
    assert y.stats is not None
    sums = torch.zeros((n, cout, 2), dtype=torch.float64, device=gpu_device)
    from elvis_amd._lib import lib, check, ptr
    check(lib().elvis_gn_partials_to_sums(ptr(y.stats), y.stats.shape[0] // n, n, cout, ptr(sums), cout, 0,
                                          torch.cuda.current_stream().cuda_stream))
    assert torch.allclose(sums.cpu()[:, :, 0], got.double().sum((2, 3)), rtol=1e-5, atol=1e-2)
    # and against the generic strided kernel on the same packed-from-OIHW weights
    direct = ops.PackedConv(wt, b, torch.float16, gpu_device, cin)
    y0 = direct(_act(x, torch.float16, gpu_device), stride=2, pad=0, ho=h // 2, wo=w // 2)
    assert (_nchw(y0) - got).abs().max().item() < TOL[torch.float16]


@pytest.mark.parametrize("cfg", [
    # cin, cout, h, w, act   (3x3 / stride 2 / pad 1 in space-to-depth form: the Blur / DCT slots' down convs)
    (64, 128, 48, 64, 0),
    (128, 256, 24, 40, 0),       # ragged tile edges
    (32, 32, 40, 96, 3),         # 32-channel tile, ReLU
])
def test_downsample_conv_space_to_depth_pad1(gpu_device, cfg):
    from elvis_amd import ops
    cin, cout, h, w, act = cfg
    g = torch.Generator().manual_seed(23)
    n = 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    assert ops.PackedDownConv.supported_pad1(torch.float16, cin, cout)
    conv = ops.PackedDownConv(wt, b, torch.float16, gpu_device, cin, pad1=True)
    y = conv(_act(x, torch.float16, gpu_device), act=act)
    ref = F.conv2d(x.half().float(), wt.half().float(), b, stride=2, padding=1)
    if act == 3:
        ref = torch.relu(ref)
    got = _nchw(y)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL[torch.float16]
    # and against the generic strided kernel
    direct = ops.PackedConv(wt, b, torch.float16, gpu_device, cin)
    y0 = direct(_act(x, torch.float16, gpu_device), stride=2, act=act)
    assert (_nchw(y0) - got).abs().max().item() < TOL[torch.float16]


@pytest.mark.parametrize("cfg", [
    # c1, c2, cout, k, h, w, prologue   (shapes that reach the x3 instantiations: 64- / 128-channel tiles)
    (128, 0, 128, 3, 40, 70, True),
    (96, 32, 64, 3, 33, 65, True),       # two inputs, 64-channel tile
    (64, 0, 256, 3, 24, 40, False),
    (192, 0, 576, 1, 40, 72, False),     # linear layer
    (32, 0, 16, 3, 17, 23, True),        # narrow output tile: stays on the exact fp32 MFMA (still must be right)
    (3, 0, 128, 3, 30, 50, False),       # conv_in: 3 of a K chunk's 16 channels exist
    (40, 0, 64, 3, 21, 37, True),        # cin not a multiple of 16
    (160, 160, 160, 3, 26, 34, True),    # planar form: two 5-chunk inputs, cout padded 160 -> 192 (three 64-channel tiles)
    (256, 0, 128, 3, 25, 64, False),     # planar form: 8 K chunks, ragged last tile row
])
def test_conv_compensated_f16_matches_fp32(gpu_device, cfg):
    """ELVIS_F32X3: fp32 tensors, products on the f16 matrix pipe with the operands split hi + lo.  Bar: the same
    2e-4 as the exact fp32 kernels against an fp32 CPU conv, and within 2e-4 of the exact fp32 kernel itself
    (plain f16 operands would be off by ~1e-2 here)."""
    from elvis_amd import ops
    c1, c2, cout, k, h, w, pro = cfg
    g = torch.Generator().manual_seed(21)
    n = 2
    x1 = torch.randn(n, c1, h, w, generator=g) * 3.0
    x2 = torch.randn(n, c2, h, w, generator=g) if c2 else None
    ctot = c1 + c2
    wt = torch.randn(cout, ctot, k, k, generator=g) / math.sqrt(ctot * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g)
    pa = torch.rand(n, ctot, generator=g) + 0.5
    pb = torch.randn(n, ctot, generator=g) * 0.2
    a1, a2, ar = _act(x1, torch.float32, gpu_device), (_act(x2, torch.float32, gpu_device) if c2 else None), _act(res, torch.float32, gpu_device)
    kw = dict(residual=ar)
    if pro:
        kw["prologue"] = (pa.to(gpu_device), pb.to(gpu_device))
    x3 = ops.PackedConv(wt, b, torch.float32, gpu_device, c1, c2, x3=True)
    exact = ops.PackedConv(wt, b, torch.float32, gpu_device, c1, c2, x3=False)
    assert x3.x3 and not exact.x3
    y3, ye = _nchw(x3(a1, a2, **kw)), _nchw(exact(a1, a2, **kw))
    xin = torch.cat([x1, x2], 1) if c2 else x1
    if pro:
        xin = F.silu(xin * pa[:, :, None, None] + pb[:, :, None, None])
    ref = F.conv2d(xin, wt, b, padding=k // 2) + res
    assert (y3 - ref).abs().max().item() < 2e-4
    assert (y3 - ye).abs().max().item() < 2e-4
    f16_err = (F.conv2d(xin.half().float(), wt.half().float(), b, padding=k // 2) + res - ref).abs().max().item()
    assert f16_err > 10 * (y3 - ref).abs().max().item()      # the compensation is doing something


@pytest.mark.parametrize("cout,act", [(128, 0), (64, 1), (128, 3)])
def test_conv_compensated_planar_stats_and_activation(gpu_device, cout, act):
    """The planar compensated 3x3 kernel (conv_x3p.inc): fused GroupNorm partial statistics of the stored values, the
    epilogue activations, and the dispatch itself (the library names the kernel it runs)."""
    import ctypes as C
    from elvis_amd import ops
    from elvis_amd._lib import ConvDesc, check, lib, ptr
    g = torch.Generator().manual_seed(31)
    n, cin, h, w = 2, 96, 27, 45
    x = torch.randn(n, cin, h, w, generator=g) * 2.0
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g)
    conv = ops.PackedConv(wt, b, torch.float32, gpu_device, cin, x3=True)
    d = ConvDesc()
    d.dtype, d.n, d.h, d.w, d.ho, d.wo = ops.F32X3_CODE, n, h, w, h, w
    d.cin, d.cin_pitch, d.cout, d.cout_pitch, d.ksize, d.stride, d.pad_before, d.act = cin, cin, cout, cout, 3, 1, 1, act
    assert ops.conv_kernel_name(d).startswith("conv3x3_x3p_kernel<%d," % cout)
    y = conv(_act(x, torch.float32, gpu_device), act=act, residual=_act(res, torch.float32, gpu_device), want_stats=True)
    pre = F.conv2d(x, wt, b, padding=1)
    ref = {0: pre, 1: F.gelu(pre), 3: F.relu(pre)}[act] + res
    got = _nchw(y)
    assert (got - ref).abs().max().item() < 2e-4
    assert y.stats is not None
    sums = torch.zeros((n, cout, 2), dtype=torch.float64, device=gpu_device)
    check(lib().elvis_gn_partials_to_sums(ptr(y.stats), y.stats.shape[0] // n, n, cout, ptr(sums), cout, 0,
                                          torch.cuda.current_stream().cuda_stream))
    assert torch.allclose(sums.cpu()[:, :, 0], got.double().sum((2, 3)), rtol=1e-6, atol=1e-3)
    assert torch.allclose(sums.cpu()[:, :, 1], (got.double() ** 2).sum((2, 3)), rtol=1e-6, atol=1e-3)


def test_conv_compensated_operand_range(gpu_device):
    """The documented operand limit of ELVIS_F32X3 (elvis_amd.h): the hi part of the hi + lo split is an f16, so an
    activation or weight with |v| >= 65504 makes the products non-finite.  In range (here |x| up to 6e4) the result is
    fp32-grade; out of range it is NOT silently wrong - it is inf / NaN."""
    from elvis_amd import ops
    g = torch.Generator().manual_seed(33)
    cin, cout, h, w = 64, 64, 16, 32
    x = torch.randn(1, cin, h, w, generator=g)
    x[0, 3, 5, 7] = 6.0e4
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    conv = ops.PackedConv(wt, None, torch.float32, gpu_device, cin, x3=True)
    y = _nchw(conv(_act(x, torch.float32, gpu_device)))
    ref = F.conv2d(x.double(), wt.double(), padding=1).float()
    # (the lo part of a small weight is an f16 subnormal - 6e-8 absolute - which a 6e4 activation scales up to ~2e-3)
    assert torch.isfinite(y).all() and ((y - ref).abs() / (ref.abs() + 1.0)).max().item() < 1e-3
    x[0, 3, 5, 7] = 1.0e5
    y = _nchw(conv(_act(x, torch.float32, gpu_device)))
    assert not torch.isfinite(y[0, :, 4:7, 6:9]).all()


def test_upconv_compensated_f16(gpu_device):
    """The sub-pixel upsample conv (four 2x2 parity convs) on the compensated path."""
    from elvis_amd import ops
    g = torch.Generator().manual_seed(22)
    cin, cout, h, w = 128, 128, 20, 36
    x = torch.randn(2, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    with ops.x3_default(True):
        up = ops.PackedUpConv(wt, b, torch.float32, gpu_device, cin)
    assert all(c.x3 for c in up.par)
    y = _nchw(up(_act(x, torch.float32, gpu_device)))
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1)
    assert (y - ref).abs().max().item() < 2e-4


def test_conv_compensated_f16_fused_upsample(gpu_device):
    """nearest-2x fused into the 3x3 conv (the non-sub-pixel form), compensated path."""
    from elvis_amd import ops
    g = torch.Generator().manual_seed(23)
    cin, cout, h, w = 64, 128, 13, 21
    x = torch.randn(2, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    conv = ops.PackedConv(wt, b, torch.float32, gpu_device, cin, x3=True)
    y = _nchw(conv(_act(x, torch.float32, gpu_device), upsample=True))
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1)
    assert (y - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("c,ratio,tokens_hw", [(64, 4, (3, 19, 23)), (128, 2, (1, 40, 33)), (192, 4, (2, 16, 40)), (256, 2, (1, 24, 24))])
def test_swin_fused_mlp_and_ln_linear(gpu_device, c, ratio, tokens_hw):
    """csrc/swin.hip: out = x + fc2(GELU(fc1(LN(x)))) and out = W.LN(x) + b in one kernel each, against an fp32 CPU
    restatement (f16 storage of x, the normalised tokens, the weights and the hidden activations emulated) and against
    the unfused kernel chain (layernorm + 1x1 convs).  Token counts that do not fill the last workgroup / wave."""
    from elvis_amd import ops
    g = torch.Generator().manual_seed(41)
    n, h, w = tokens_hw
    hid = ratio * c
    x = torch.randn(n, c, h, w, generator=g) * 1.5 + torch.randn(n, 1, h, w, generator=g)
    nw, nb = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    w1, b1 = torch.randn(hid, c, generator=g) / math.sqrt(c), torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(c, hid, generator=g) / math.sqrt(hid), torch.randn(c, generator=g) * 0.1
    wq, bq = torch.randn(3 * c, c, generator=g) / math.sqrt(c), torch.randn(3 * c, generator=g) * 0.1
    xa = _act(x, torch.float16, gpu_device)
    q = lambda t: t.half().float()
    xt = q(x).permute(0, 2, 3, 1)                                        # tokens last-dim channels
    ln = q(F.layer_norm(xt, (c,), nw, nb, 1e-5))
    ref_mlp = xt + F.linear(q(F.gelu(F.linear(ln, q(w1), b1))), q(w2), b2)
    ref_qkv = F.linear(ln, q(wq), bq)

    mlp = ops.SwinFused(nw, nb, w1, b1, w2, b2, device=gpu_device)
    got = mlp(xa).t[..., :c].float().cpu()
    assert (got - ref_mlp).abs().max().item() < 2e-2
    lin = ops.SwinFused(nw, nb, wq, bq, device=gpu_device)
    gq = lin(xa).t[..., :3 * c].float().cpu()
    assert (gq - ref_qkv).abs().max().item() < 2e-2
    # the unfused chain of product kernels
    t = ops.layernorm(xa, nw.to(gpu_device), nb.to(gpu_device))
    fc1 = ops.PackedConv(w1[:, :, None, None], b1, torch.float16, gpu_device, c)
    fc2 = ops.PackedConv(w2[:, :, None, None], b2, torch.float16, gpu_device, hid)
    chain = fc2(fc1(t, act=1), residual=xa).t[..., :c].float().cpu()
    assert (got - chain).abs().max().item() < 2e-2
    qk = ops.PackedConv(wq[:, :, None, None], bq, torch.float16, gpu_device, c)(t).t[..., :3 * c].float().cpu()
    assert (gq - qk).abs().max().item() < 2e-2
    # tighter, in rms: the fused kernels are as close to the fp32 restatement as the chain is
    rms = lambda a, b: (a - b).pow(2).mean().sqrt().item()
    assert rms(got, ref_mlp) < 1.5 * rms(chain, ref_mlp) + 1e-4 and rms(gq, ref_qkv) < 1.5 * rms(qk, ref_qkv) + 1e-4
    with pytest.raises(ValueError):
        ops.SwinFused(nw[:40], nb[:40], w1[:, :40], b1, device=gpu_device)
    # the projection folded in front of the MLP: y' = y + proj(a); out = y' + fc2(GELU(fc1(LN(y'))))
    a = torch.randn(n, c, h, w, generator=g)
    wp, bp = torch.randn(c, c, generator=g) / math.sqrt(c), torch.randn(c, generator=g) * 0.1
    aa = _act(a, torch.float16, gpu_device)
    y1 = xt + F.linear(q(a).permute(0, 2, 3, 1), q(wp), bp)              # kept in fp32 by the fused kernel
    ln1 = q(F.layer_norm(y1, (c,), nw, nb, 1e-5))
    ref_pm = y1 + F.linear(q(F.gelu(F.linear(ln1, q(w1), b1))), q(w2), b2)
    pm = ops.SwinFused(nw, nb, w1, b1, w2, b2, proj_w=wp, proj_b=bp, device=gpu_device)
    got_pm = pm(aa, xa).t[..., :c].float().cpu()
    assert (got_pm - ref_pm).abs().max().item() < 2e-2
    proj = ops.PackedConv(wp[:, :, None, None], bp, torch.float16, gpu_device, c)
    chain_pm = mlp(proj(aa, residual=xa)).t[..., :c].float().cpu()        # unfused projection + the fused MLP
    assert (got_pm - chain_pm).abs().max().item() < 2e-2
    assert rms(got_pm, ref_pm) < 1.5 * rms(chain_pm, ref_pm) + 1e-4
    with pytest.raises(ValueError):
        pm(aa)                                                            # the projection form needs the residual stream


@pytest.mark.parametrize("cfg", [(128, 128, 20, 36), (48, 192, 18, 70), (256, 320, 34, 64)])
def test_downsample_conv_space_to_depth_compensated(gpu_device, cfg):
    """The same space-to-depth form on fp32 tensors with the compensated f16 MFMA (ELVIS_F32X3; the x3 / dec_f16 modes'
    encoder): fp32-grade against the direct fp32 conv, where the exact strided kernel ran at a sixteenth of the rate."""
    from elvis_amd import ops
    cin, cout, h, w = cfg
    g = torch.Generator().manual_seed(24)
    n = 2
    x = torch.randn(n, cin, h, w, generator=g) * 2.0
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    assert ops.PackedDownConv.supported(torch.float32, cin, cout, x3=True) and not ops.PackedDownConv.supported(torch.float32, cin, cout)
    conv = ops.PackedDownConv(wt, b, torch.float32, gpu_device, cin)
    y = conv(_act(x, torch.float32, gpu_device), want_stats=True)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, b, stride=2)
    got = _nchw(y)
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 2e-4
    assert y.stats is not None


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("cin,cout,k,stride", [(32, 189, 3, 1), (16, 20, 3, 1), (64, 1, 3, 1), (24, 3, 1, 1), (16, 12, 3, 2)])
def test_conv_writes_its_pad_channels(gpu_device, dtype, cin, cout, k, stride):
    """Activations keep channels [cout, pitch) at zero (the next conv's staging reads them).  The PRODUCING conv writes
    those zeros itself - ops no longer zero-fills every output through torch (5 % of the DCT slot) - on
    every epilogue path: whole 4-channel groups, a ragged last group, pad-only groups, the generic strided kernel."""
    from elvis_amd import ops
    g = torch.Generator().manual_seed(51)
    n, h, w = 2, 18, 34
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    conv = ops.PackedConv(wt, b, dtype, gpu_device, cin)
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    out = ops.new_act(n, ho, wo, cout, dtype, gpu_device, zero=False)
    out.t.fill_(float("nan"))
    y = conv(_act(x, dtype, gpu_device), stride=stride, out=out)
    assert y.pitch > cout or cout % 8 == 0
    assert torch.count_nonzero(y.t[..., cout:]).item() == 0 and not torch.isnan(y.t).any()
    ref = F.conv2d(x.to(dtype).float(), wt.to(dtype).float(), b, stride=stride, padding=k // 2)
    assert (_nchw(y) - ref).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize("cfg", [
    # c1, c2, cout, h, w, act   (narrow f16 3x3 layers on >= 128 K pixels: the persistent weight-stationary kernel, conv_ws.inc)
    (64, 0, 64, 270, 500, 3),     # the DCT slot's quality conv; ragged right and bottom edges, odd tile count
    (32, 32, 32, 264, 512, 3),    # virtual concat: one K chunk per input
    (7, 0, 32, 264, 512, 3),      # cin below one K chunk (pitch 8)
    (32, 0, 189, 264, 512, 0),    # three 64-channel cout tiles, ragged cout padded to 192
    (64, 0, 1, 264, 512, 0),      # 16-channel tile, one real channel
])
def test_conv_weight_stationary(gpu_device, cfg):
    from elvis_amd import ops
    c1, c2, cout, h, w, act = cfg
    dtype = torch.float16
    g = torch.Generator().manual_seed(31)
    n = 2
    x1 = torch.randn(n, c1, h, w, generator=g)
    x2 = torch.randn(n, c2, h, w, generator=g) if c2 else None
    ctot = c1 + c2
    wt = torch.randn(cout, ctot, 3, 3, generator=g) / math.sqrt(ctot * 9)
    b = torch.randn(cout, generator=g) * 0.1
    conv = ops.PackedConv(wt, b, dtype, gpu_device, c1, c2)
    a1 = _act(x1, dtype, gpu_device)
    a2 = _act(x2, dtype, gpu_device) if c2 else None
    d = ops.ConvDesc()
    d.dtype, d.n, d.h, d.w, d.ho, d.wo = ops.L.dtype_code(dtype), n, h, w, h, w
    d.cin, d.cin_pitch, d.cin2, d.cin2_pitch = c1, a1.pitch, c2, (a2.pitch if c2 else 0)
    d.cout, d.cout_pitch, d.ksize, d.stride, d.pad_before, d.act = conv.cout_k, ops.pitch_for(cout), 3, 1, 1, act
    assert ops.conv_kernel_name(d).startswith("conv3x3_ws_kernel")
    y = conv(a1, a2, act=act)
    xcat = (torch.cat([x1, x2], 1) if c2 else x1).to(dtype).float()
    ref = F.conv2d(xcat, wt.to(dtype).float(), b, padding=1)
    if act == 3:
        ref = torch.relu(ref)
    got = _nchw(y)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL[dtype]
    assert y.t[..., cout:].abs().max().item() == 0 if y.t.shape[-1] > cout else True   # pad channels are written as zeros
    # same result as the halo-tile kernels (ELVIS_NO_HALO also switches this kernel off: both fall back to the generic one)
    from elvis_amd._lib import lib, check
    check(lib().elvis_conv_debug_set(b"no_halo", 1))
    try:
        y0 = conv(a1, a2, act=act)
    finally:
        check(lib().elvis_conv_debug_set(b"no_halo", -1))
    assert (_nchw(y0) - got).abs().max().item() < TOL[dtype]


def test_conv_weight_stationary_odd_tile_count(gpu_device):
    """One image of 33 x 17 = 561 tiles: the last tile pair has only its first half (conv_ws.inc: `t < ntiles`)."""
    from elvis_amd import ops
    g = torch.Generator().manual_seed(32)
    cin, cout, h, w = 32, 64, 264, 544
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    conv = ops.PackedConv(wt, b, torch.float16, gpu_device, cin)
    y = conv(_act(x, torch.float16, gpu_device), act=3)
    ref = torch.relu(F.conv2d(x.half().float(), wt.half().float(), b, padding=1))
    assert (_nchw(y) - ref).abs().max().item() < TOL[torch.float16]
