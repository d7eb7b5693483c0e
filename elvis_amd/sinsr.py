"""Device graph of the SinSR-style single-step 4x super-resolver (ELVIS v2 Downsample slot).

This fills the model slot where the reference calls `RealESRGANer.enhance`
(elvis.py:2507-2519, a5 in SURVEY.md 8a); the north star names SinSR (README.md:27-28,46,50),
whose code is absent from the reference, so the architecture is the published
SinSR/ResShift recipe with hyper-parameters in `weights.SinSRConfig`.

Every arithmetic op is a libelvis_amd.so kernel (ops.py); torch only owns the HBM buffers
and the stream.  NHWC throughout; `dtype=torch.float32` is the exact-parity mode (fp32 MFMA),
`torch.float16` the fast mode (f16 MFMA, fp32 accumulate).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import ops
from .ops import Act, PackedConv, PackedUpConv
from .weights import SinSRConfig, frame_noise, make_sinsr_weights, timestep_embedding, unet_layout


import os as _os_mod
_SWIN_FUSE = not _os_mod.environ.get("ELVIS_NO_SWIN_FUSE")   # A/B switch, read once at import


class _GN:
    def __init__(self, sd, p, device):
        self.gamma = sd[p + ".weight"].to(device=device, dtype=torch.float32).contiguous()
        self.beta = sd[p + ".bias"].to(device=device, dtype=torch.float32).contiguous()


def _conv(sd, p, dtype, device, cin, cin2=0) -> PackedConv:
    return PackedConv(sd[p + ".weight"], sd[p + ".bias"], dtype, device, cin, cin2)


class _DownConv:
    """The autoencoder's Downsample: pad (0,1,0,1) + 3x3 conv, stride 2.  f16 (whole 32-channel chunks) and
    compensated-fp32 sections (16-channel chunks) with even sizes: the space-to-depth form on the halo-tile
    kernel (ops.PackedDownConv, with fused GroupNorm statistics); otherwise (exact fp32 mode, ELVIS_NO_S2D=1,
    odd sizes) the generic strided kernel - same result to rounding."""

    def __init__(self, sd, p, dtype, device, cin):
        import os
        self.direct = _conv(sd, p, dtype, device, cin)
        self.s2d = None
        if ops.PackedDownConv.supported(dtype, cin, sd[p + ".weight"].shape[0], x3=ops._X3_DEFAULT) and not os.environ.get("ELVIS_NO_S2D"):
            self.s2d = ops.PackedDownConv(sd[p + ".weight"], sd[p + ".bias"], dtype, device, cin)

    def __call__(self, x: Act, want_stats=False) -> Act:
        if self.s2d is not None and x.h % 2 == 0 and x.w % 2 == 0:
            return self.s2d(x, want_stats=want_stats)
        return self.direct(x, stride=2, pad=0, ho=x.h // 2, wo=x.w // 2)


class _UpConv:
    """nearest-2x upsample + 3x3 conv: sub-pixel 2x2 decomposition (2.25x fewer FLOPs) or the
    fused-upsample 3x3 kernel (ELVIS_NO_SUBPIXEL=1 / subpixel=False), same results to rounding."""

    def __init__(self, sd, p, dtype, device, cin, subpixel=True):
        self.sub = PackedUpConv(sd[p + ".weight"], sd[p + ".bias"], dtype, device, cin) if subpixel else None
        self.full = None if subpixel else _conv(sd, p, dtype, device, cin)

    def __call__(self, x: Act, want_stats=False) -> Act:
        if self.sub is not None:
            return self.sub(x, want_stats=want_stats)
        return self.full(x, upsample=True, want_stats=want_stats)


def _linear(sd, p, dtype, device) -> PackedConv:
    w = sd[p + ".weight"]
    return PackedConv(w[:, :, None, None], sd.get(p + ".bias"), dtype, device, w.shape[1])


class _ResBlock:
    """GN -> SiLU -> conv3x3 -> GN(*(1+scale)+shift) -> SiLU -> conv3x3, + skip.
    Serves both the UNet ResBlock (scale/shift from the timestep embedding) and the
    autoencoder ResnetBlock (no embedding)."""

    def __init__(self, sd, names, cin, cin2, cout, dtype, device, groups, eps, scale=None, shift=None):
        n1, c1, n2, c2, skip = names
        self.groups, self.eps = groups, eps
        self.norm1, self.norm2 = _GN(sd, n1, device), _GN(sd, n2, device)
        self.conv1 = _conv(sd, c1, dtype, device, cin, cin2)
        self.conv2 = _conv(sd, c2, dtype, device, cout)
        self.skip = _conv(sd, skip, dtype, device, cin, cin2) if (skip + ".weight") in sd else None
        self.scale, self.shift = scale, shift
        if self.skip is None and cin2:
            raise ValueError("a concatenated input needs a skip projection")

    # The fused prologue recomputes normalise+SiLU once per 128-channel output tile (and 1.33x for
    # the halo); with >= 4 output tiles it is cheaper to materialise the activated tensor once
    # (one extra read+write) and run the faster prologue-free 16x32-tile kernel.
    import os as _os
    FUSE_MAX_COUT = int(_os.environ.get("ELVIS_FUSE_MAX_COUT", str(1 << 30)))   # A/B switch, see DESIGN.md 5.2

    def __call__(self, x: Act, x2: Optional[Act] = None, fuse_gn=False) -> Act:
        xs = [x] if x2 is None else [x, x2]
        pa, pb = ops.groupnorm_affine(xs, self.norm1.gamma, self.norm1.beta, self.groups, self.eps)
        fuse1 = fuse_gn and self.conv1.cout <= self.FUSE_MAX_COUT
        fuse2 = fuse_gn and self.conv2.cout <= self.FUSE_MAX_COUT
        if fuse1:
            h = self.conv1(x, x2, prologue=(pa, pb), want_stats=True)
        else:
            a1 = ops.affine_act(x, pa[:, :x.c].contiguous(), pb[:, :x.c].contiguous(), act=2)
            a2 = None
            if x2 is not None:
                a2 = ops.affine_act(x2, pa[:, x.c:].contiguous(), pb[:, x.c:].contiguous(), act=2)
            h = self.conv1(a1, a2, want_stats=fuse_gn)
            del a1, a2
        pa, pb = ops.groupnorm_affine([h], self.norm2.gamma, self.norm2.beta, self.groups, self.eps,
                                      scale=self.scale, shift=self.shift)
        res = x if self.skip is None else self.skip(x, x2)
        if fuse2:
            return self.conv2(h, prologue=(pa, pb), residual=res, want_stats=True)
        ops.affine_act(h, pa, pb, act=2, out=h)
        return self.conv2(h, residual=res, want_stats=fuse_gn)


class _SwinLayer:
    def __init__(self, sd, p, ch, cfg: SinSRConfig, dtype, device):
        E = cfg.swin_embed_dim
        self.cfg = cfg
        f32 = dict(device=device, dtype=torch.float32)
        self.embed = _conv(sd, p + ".patch_embed.proj", dtype, device, ch)
        self.embed_norm = (sd[p + ".patch_embed.norm.weight"].to(**f32), sd[p + ".patch_embed.norm.bias"].to(**f32))
        self.blocks = []
        fuse = _SWIN_FUSE and ops.SwinFused.supported(dtype, E, 3 * E) and ops.SwinFused.supported(dtype, E, cfg.mlp_ratio * E)
        for d in range(cfg.swin_depth):
            b = f"{p}.blocks.{d}"
            w = lambda n: sd[b + n]
            self.blocks.append(dict(
                # f16: LayerNorm + qkv in one kernel, LayerNorm + fc1 + GELU + fc2 + residual in one kernel (csrc/swin.hip)
                qkv_f=ops.SwinFused(w(".norm1.weight"), w(".norm1.bias"), w(".attn.qkv.weight"), w(".attn.qkv.bias"), device=device) if fuse else None,
                # ... and the attention output projection folded in front of the MLP: y' = y + proj(a) never reaches HBM
                mlp_f=ops.SwinFused(w(".norm2.weight"), w(".norm2.bias"), w(".mlp.fc1.weight"), w(".mlp.fc1.bias"),
                                    w(".mlp.fc2.weight"), w(".mlp.fc2.bias"), device=device,
                                    **(dict(proj_w=w(".attn.proj.weight"), proj_b=w(".attn.proj.bias")) if ops.SwinFused.proj_pays(E) else {})) if fuse else None,
                n1=(sd[b + ".norm1.weight"].to(**f32), sd[b + ".norm1.bias"].to(**f32)),
                n2=(sd[b + ".norm2.weight"].to(**f32), sd[b + ".norm2.bias"].to(**f32)),
                qkv=_linear(sd, b + ".attn.qkv", dtype, device),
                proj=_linear(sd, b + ".attn.proj", dtype, device),
                fc1=_linear(sd, b + ".mlp.fc1", dtype, device),
                fc2=_linear(sd, b + ".mlp.fc2", dtype, device),
                table=sd[b + ".attn.relative_position_bias_table"].to(**f32).contiguous(),
                shift=0 if d % 2 == 0 else cfg.window_size // 2))
        self.unembed = _conv(sd, p + ".patch_unembed.proj", dtype, device, E)

    def __call__(self, x: Act) -> Act:
        cfg = self.cfg
        y = self.embed(x)
        ops.layernorm(y, *self.embed_norm, out=y)
        for b in self.blocks:
            qkv = b["qkv_f"](y) if b["qkv_f"] is not None else b["qkv"](ops.layernorm(y, *b["n1"]))
            a = ops.window_attention(qkv, cfg.heads, cfg.num_head_channels, cfg.window_size, b["shift"], b["table"],
                                     cfg.num_head_channels ** -0.5)
            del qkv
            if b["mlp_f"] is not None:
                y = b["mlp_f"](a, y) if b["mlp_f"].proj else b["mlp_f"](b["proj"](a, residual=y))
            else:
                y = b["proj"](a, residual=y)
                t = ops.layernorm(y, *b["n2"])
                t = b["fc1"](t, act=1)
                y = b["fc2"](t, residual=y)
        return self.unembed(y)


class SinSRModel:
    """Weights resident in HBM, packed once; `forward` runs one batch of LR frames."""

    # Sections of the graph that may run at different precisions (precision="mixed[:sec+sec...]" runs the
    # listed sections in f16 and the rest in fp32; tools/precision_study.py).
    SECTIONS = ("enc0", "enc1", "enc2", "unet", "dec2", "dec1", "dec0")
    MIXED_F16_DEFAULT = ("dec0",)
    DEC_SECTIONS = ("dec2", "dec1", "dec0")

    def __init__(self, cfg: SinSRConfig = SinSRConfig(), state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 device="cuda:0", dtype=torch.float16, weight_seed: int = 0, fuse_gn: bool = True,
                 precision: Optional[str] = None):
        self.cfg, self.device, self.dtype, self.fuse_gn = cfg, torch.device(device), dtype, fuse_gn
        self.precision = precision or ("f16" if dtype == torch.float16 else "f32")
        if precision == "dec_f16":
            # encoder + Swin-UNet compensated (fp32-grade latent: no VQ code flips), the whole decoder - everything
            # behind the nearest-code lookup - in f16: the u8 frame stays within 1 LSB of the CPU path's
            precision = "mixed:" + "+".join(self.DEC_SECTIONS)
        self.sec_dtype = {sec: dtype for sec in self.SECTIONS}
        # "x3" / "mixed..." : fp32 convs run on the f16 matrix pipe with the rounding error compensated (ops.x3_default,
        # conv.hip mma_tile_x); "f32" / "mixed_exact..." keep the exact fp32 MFMA
        self.x3 = precision is not None and (precision == "x3" or (precision.startswith("mixed") and not precision.startswith("mixed_exact")))
        if precision == "x3":
            self.dtype = dtype = torch.float32
            self.sec_dtype = {sec: dtype for sec in self.SECTIONS}
            precision = "f32"
        if precision is not None and precision.startswith("mixed"):
            f16 = tuple(precision.split(":", 1)[1].split("+")) if ":" in precision else self.MIXED_F16_DEFAULT
            unknown = [sec for sec in f16 if sec not in self.SECTIONS]
            if unknown:
                raise ValueError(f"unknown section(s) {unknown}; sections are {self.SECTIONS}")
            self.sec_dtype = {sec: (torch.float16 if sec in f16 else torch.float32) for sec in self.SECTIONS}
            self.dtype = self.sec_dtype["unet"]
        elif precision not in (None, "f16", "f32"):
            raise ValueError(f"unknown precision '{precision}'")
        import os
        self.subpixel_up = not os.environ.get("ELVIS_NO_SUBPIXEL")
        sd = state_dict if state_dict is not None else make_sinsr_weights(cfg, weight_seed)
        dev = self.device
        with torch.cuda.device(dev), ops.x3_default(self.x3):
            self._build_unet(sd)
            self._build_ae(sd)
        self.codebook = sd["ae.quantize.embedding.weight"].to(device=dev, dtype=torch.float32).contiguous()

    # ------------------------------------------------------------------ construction
    def _build_unet(self, sd):
        cfg, dev, dt = self.cfg, self.device, self.sec_dtype["unet"]
        # timestep path: load-time only, always in fp32 kernels
        te0 = _linear(sd, "model.time_embed.0", torch.float32, dev)
        te2 = _linear(sd, "model.time_embed.2", torch.float32, dev)
        e = timestep_embedding(cfg.steps - 1, cfg.model_channels).to(dev)
        e = Act(e.view(1, 1, 1, -1).contiguous(), cfg.model_channels)
        e = te2(te0(e, act=2))                      # Linear -> SiLU -> Linear
        ones = torch.ones((1, e.c), device=dev)
        zeros = torch.zeros((1, e.c), device=dev)
        silu_e = ops.affine_act(e, ones, zeros, act=2)  # emb_layers.0 = SiLU
        plan = unet_layout(cfg)
        self.plan = plan

        def make(kind, p, meta, cin2=0):
            p = "model." + p
            if kind == "conv_in":
                return _conv(sd, p, dt, dev, meta[0])
            if kind == "res":
                cin, cout = meta[0] - cin2, meta[1]
                emb = _linear(sd, p + ".emb_layers.1", torch.float32, dev)(silu_e)
                ss = emb.t.view(-1)[: 2 * cout].clone()
                names = (p + ".in_layers.0", p + ".in_layers.2", p + ".out_layers.0", p + ".out_layers.3",
                         p + ".skip_connection")
                return _ResBlock(sd, names, cin, cin2, cout, dt, dev, cfg.gn_groups, 1e-5,
                                 scale=ss[:cout].contiguous(), shift=ss[cout:].contiguous())
            if kind == "swin":
                return _SwinLayer(sd, p, meta[0], cfg, dt, dev)
            if kind == "up":
                return _UpConv(sd, p, dt, dev, meta[0], self.subpixel_up)
            if kind == "down":
                return _conv(sd, p, dt, dev, meta[0])
            raise ValueError(kind)

        self.u_input, chans = [], []
        for kind, p, meta in plan["input"]:
            ops_ = meta if kind == "seq" else [(kind, p, meta)]
            mods = [(k, make(k, pp, m)) for k, pp, m in ops_]
            self.u_input.append(mods)
            chans.append(ops_[-1][2][-1] if ops_[-1][0] != "swin" else ops_[0][2][1])
        self.u_middle = [(k, make(k, pp, m)) for k, pp, m in plan["middle"]]
        self.u_output = []
        for kind, p, meta in plan["output"]:
            ich = chans.pop()
            mods = []
            for j, (k, pp, m) in enumerate(meta):
                mods.append((k, make(k, pp, m, cin2=ich if j == 0 else 0)))
            self.u_output.append(mods)
        self.u_out_norm = _GN(sd, "model.out.0", dev)
        self.u_out_conv = _conv(sd, "model.out.2", dt, dev, plan["out_ch"])

    def _ae_section(self, side: str, lvl: int) -> str:
        """Section of autoencoder level `lvl` (0 = full resolution); deeper levels than the named ones
        (narrow test configs have the same three) fold into the last."""
        return f"{side}{min(lvl, 2)}"

    def _build_ae(self, sd):
        cfg, dev = self.cfg, self.device
        ch, mults, nrb, g = cfg.ae_ch, cfg.ae_ch_mult, cfg.ae_num_res_blocks, cfg.gn_groups
        last = len(mults) - 1

        def rb(p, cin, cout, dt):
            names = (p + ".norm1", p + ".conv1", p + ".norm2", p + ".conv2", p + ".nin_shortcut")
            return _ResBlock(sd, names, cin, 0, cout, dt, dev, g, 1e-6)

        self.e_conv_in = _conv(sd, "ae.encoder.conv_in", self.sec_dtype["enc0"], dev, 3)
        self.e_down = []
        cin = ch
        for lvl, m in enumerate(mults):
            dt = self.sec_dtype[self._ae_section("enc", lvl)]
            blocks = []
            for b in range(nrb):
                blocks.append(rb(f"ae.encoder.down.{lvl}.block.{b}", cin, ch * m, dt))
                cin = ch * m
            ds = _DownConv(sd, f"ae.encoder.down.{lvl}.downsample.conv", dt, dev, cin) if lvl != last else None
            self.e_down.append((blocks, ds, dt))
        dt = self.sec_dtype[self._ae_section("enc", last)]
        self.e_mid = [rb("ae.encoder.mid.block_1", cin, cin, dt), rb("ae.encoder.mid.block_2", cin, cin, dt)]
        self.e_norm_out = _GN(sd, "ae.encoder.norm_out", dev)
        self.e_conv_out = _conv(sd, "ae.encoder.conv_out", dt, dev, cin)
        self.quant_conv = _conv(sd, "ae.quant_conv", dt, dev, cfg.z_channels)
        dt = self.sec_dtype[self._ae_section("dec", last)]
        self.post_quant_conv = _conv(sd, "ae.post_quant_conv", dt, dev, cfg.embed_dim)
        cin = ch * mults[-1]
        self.d_conv_in = _conv(sd, "ae.decoder.conv_in", dt, dev, cfg.z_channels)
        self.d_mid = [rb("ae.decoder.mid.block_1", cin, cin, dt), rb("ae.decoder.mid.block_2", cin, cin, dt)]
        self.d_up = []
        for lvl in reversed(range(len(mults))):
            dt = self.sec_dtype[self._ae_section("dec", lvl)]
            blocks = []
            for b in range(nrb + 1):
                blocks.append(rb(f"ae.decoder.up.{lvl}.block.{b}", cin, ch * mults[lvl], dt))
                cin = ch * mults[lvl]
            us = _UpConv(sd, f"ae.decoder.up.{lvl}.upsample.conv", dt, dev, cin, self.subpixel_up) if lvl != 0 else None
            self.d_up.append((blocks, us, dt))
        self.d_norm_out = _GN(sd, "ae.decoder.norm_out", dev)
        self.d_conv_out = _conv(sd, "ae.decoder.conv_out", self.sec_dtype["dec0"], dev, cin)

    # ------------------------------------------------------------------ stages
    def _gn_silu_conv(self, x: Act, norm: _GN, conv: PackedConv, eps: float) -> Act:
        pa, pb = ops.groupnorm_affine([x], norm.gamma, norm.beta, self.cfg.gn_groups, eps)
        if self.fuse_gn:
            return conv(x, prologue=(pa, pb))
        ops.affine_act(x, pa, pb, act=2, out=x)
        return conv(x)

    def unet(self, x: Act) -> Act:
        def run(mods, h, h2=None):
            for j, (kind, m) in enumerate(mods):
                if kind == "res":
                    h = m(h, h2 if j == 0 else None, fuse_gn=self.fuse_gn)
                elif kind == "swin":
                    h = m(h)
                elif kind == "down":
                    h = m(h, stride=2)
                elif kind == "up":
                    h = m(h, want_stats=self.fuse_gn)
                else:
                    h = m(h)
            return h

        hs: List[Act] = []
        h = x
        for mods in self.u_input:
            h = run(mods, h)
            hs.append(h)
        h = run(self.u_middle, h)
        for mods in self.u_output:
            h = run(mods, h, hs.pop())
        return self._gn_silu_conv(h, self.u_out_norm, self.u_out_conv, 1e-5)

    def encode(self, x: Act) -> Act:
        h = self.e_conv_in(ops.convert_act(x, self.sec_dtype["enc0"]), want_stats=self.fuse_gn)
        for blocks, ds, dt in self.e_down:
            h = ops.convert_act(h, dt)
            for b in blocks:
                h = b(h, fuse_gn=self.fuse_gn)
            if ds is not None:
                h = ds(h, want_stats=self.fuse_gn)
        for b in self.e_mid:
            h = b(h, fuse_gn=self.fuse_gn)
        h = self._gn_silu_conv(h, self.e_norm_out, self.e_conv_out, 1e-6)
        return self.quant_conv(h)

    def decode(self, z: Act, quantize: Optional[bool] = None, want_idx=False):
        idx = None
        if self.cfg.quantize if quantize is None else quantize:
            z, idx = ops.vq_nearest(z, self.codebook, want_idx=True)
        z = ops.convert_act(z, self.sec_dtype[self._ae_section("dec", len(self.cfg.ae_ch_mult) - 1)])
        h = self.d_conv_in(self.post_quant_conv(z), want_stats=self.fuse_gn)
        for b in self.d_mid:
            h = b(h, fuse_gn=self.fuse_gn)
        for blocks, us, dt in self.d_up:
            h = ops.convert_act(h, dt)
            for b in blocks:
                h = b(h, fuse_gn=self.fuse_gn)
            if us is not None:
                h = us(h, want_stats=self.fuse_gn)
        out = self._gn_silu_conv(h, self.d_norm_out, self.d_conv_out, 1e-6)
        return (out, idx) if want_idx else out

    def padded_latent_shape(self, h, w):
        a = self.cfg.unet_align
        return (math.ceil(h / a) * a, math.ceil(w / a) * a)

    def forward(self, lr_u8: torch.Tensor, noise: torch.Tensor, *, swap_rb=False, want_f32=False,
                stages: Optional[dict] = None):
        """lr_u8: [n,h,w,3] u8 on the device.  noise: [n,latent_ch,Hp,Wp] f32 on the device.
        Returns [n,4h,4w,3] u8 (and the pre-quantisation f32 image in [0,1] if want_f32)."""
        cfg = self.cfg
        n, h, w, _ = lr_u8.shape
        hp, wp = self.padded_latent_shape(h, w)
        if tuple(noise.shape) != (n, cfg.latent_ch, hp, wp):
            raise ValueError(f"noise must be {(n, cfg.latent_ch, hp, wp)}, got {tuple(noise.shape)}")
        y = ops.u8_to_float(lr_u8, self.dtype, 2.0, -1.0, swap_rb=swap_rb, div255=True)   # self.dtype = the UNet's
        y_up = ops.bicubic_upsample(ops.convert_act(y, self.sec_dtype["enc0"]), cfg.sf)
        z_y = ops.convert_act(self.encode(y_up), self.dtype)
        if stages is not None:
            stages["y_up"], stages["z_y"] = y_up, z_y
        del y_up
        std = math.sqrt(cfg.etas_end * cfg.kappa ** 2 + 1.0)
        uin = ops.new_act(n, hp, wp, 2 * cfg.latent_ch, self.dtype, self.device, zero=True)
        ops.pad_reflect_axpy(z_y, hp, wp, uin, 0, mul=1.0 / std, add=noise.contiguous(),
                             add_mul=cfg.kappa * math.sqrt(cfg.etas_end) / std)
        ops.pad_reflect_axpy(y, hp, wp, uin, cfg.latent_ch, mul=1.0)
        z0 = self.unet(uin)
        if hp != h or wp != w:
            z0 = ops.crop_copy(z0, h, w)
        if stages is not None:
            stages["z0"] = z0
        dec = self.decode(z0)   # (converts z0 to the decoder's first section itself)
        if stages is not None:
            stages["dec"] = dec
        return ops.float_to_u8(dec, 0.5, 0.5, mode=0, swap_rb=swap_rb, want_f32=want_f32)

    def make_noise(self, seed: int, frame_indices, h: int, w: int) -> torch.Tensor:
        hp, wp = self.padded_latent_shape(h, w)
        ns = [frame_noise(self.cfg, seed, fi, hp, wp) for fi in frame_indices]
        return torch.cat(ns, 0).to(self.device, non_blocking=True)
