"""Seeded synthetic weights for the SinSR-style 4x super-resolver.

The reference ships no SinSR code or checkpoint (README.md:27-28 is the
only mention), and there is no network in the build or GPU environment, so weights are
random-initialised from a fixed seed (weight seed 0).  Key names
follow the upstream `state_dict` layout of ResShift/SinSR (`UNetModelSwin`) and the
LDM VQ-f4 autoencoder (`VQModelTorch`) as far as they are public knowledge, so a real
checkpoint could be dropped in later by passing its state_dict instead.

All tensors are fp32, CPU, PyTorch-native layouts (conv OIHW, linear [out,in]).
The device-side packing (NHWC / MFMA fragment order) happens in `elvis_amd.sinsr`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Tuple

import torch


@dataclass(frozen=True)
class SinSRConfig:
    # --- diffusion (ResShift/SinSR single step) ---
    sf: int = 4
    steps: int = 15
    kappa: float = 2.0
    etas_end: float = 0.99
    # --- UNetModelSwin ---
    latent_ch: int = 3
    model_channels: int = 160
    channel_mult: Tuple[int, ...] = (1, 2, 2, 4)
    num_res_blocks: int = 2
    swin_embed_dim: int = 192
    num_head_channels: int = 32
    window_size: int = 8
    swin_depth: int = 2
    mlp_ratio: int = 4
    gn_groups: int = 32
    # --- VQ-f4 autoencoder ---
    ae_ch: int = 128
    ae_ch_mult: Tuple[int, ...] = (1, 2, 4)
    ae_num_res_blocks: int = 2
    z_channels: int = 3
    embed_dim: int = 3
    n_embed: int = 8192
    quantize: bool = True

    @property
    def heads(self) -> int:
        return self.swin_embed_dim // self.num_head_channels

    @property
    def unet_align(self) -> int:
        """latent H,W must be a multiple of window * 2**(levels-1)."""
        return self.window_size * (2 ** (len(self.channel_mult) - 1))

    @property
    def time_embed_dim(self) -> int:
        return self.model_channels * 4


def tiny_config() -> SinSRConfig:
    """A structurally identical but narrow model for fast CPU/GPU parity tests."""
    return SinSRConfig(model_channels=32, channel_mult=(1, 2), num_res_blocks=1, swin_embed_dim=64,
                       num_head_channels=32, window_size=8, swin_depth=2, gn_groups=8,
                       ae_ch=32, ae_ch_mult=(1, 2, 2), ae_num_res_blocks=1, n_embed=256)


class _Init:
    def __init__(self, seed: int):
        self.g = torch.Generator().manual_seed(seed)
        self.sd: Dict[str, torch.Tensor] = {}

    def randn(self, *shape, std=1.0):
        return torch.randn(*shape, generator=self.g, dtype=torch.float32) * std

    def conv(self, name, cin, cout, k, gain=1.0):
        fan_in = cin * k * k
        self.sd[name + ".weight"] = self.randn(cout, cin, k, k, std=gain / math.sqrt(fan_in))
        self.sd[name + ".bias"] = self.randn(cout, std=0.02)

    def linear(self, name, cin, cout, gain=1.0, bias=True):
        self.sd[name + ".weight"] = self.randn(cout, cin, std=gain / math.sqrt(cin))
        if bias:
            self.sd[name + ".bias"] = self.randn(cout, std=0.02)

    def norm(self, name, c):
        self.sd[name + ".weight"] = 1.0 + self.randn(c, std=0.05)
        self.sd[name + ".bias"] = self.randn(c, std=0.05)


def _unet_resblock(I: _Init, p: str, cin: int, cout: int, emb: int):
    # ResBlock(use_scale_shift_norm=True): in_layers.{0 GN,2 conv}, emb_layers.1 linear,
    # out_layers.{0 GN,3 conv}, skip_connection (1x1 conv when cin != cout)
    I.norm(p + ".in_layers.0", cin)
    I.conv(p + ".in_layers.2", cin, cout, 3)
    I.linear(p + ".emb_layers.1", emb, 2 * cout, gain=0.3)
    I.norm(p + ".out_layers.0", cout)
    I.conv(p + ".out_layers.3", cout, cout, 3, gain=0.5)
    if cin != cout:
        I.conv(p + ".skip_connection", cin, cout, 1)


def _swin_layer(I: _Init, p: str, ch: int, cfg: SinSRConfig):
    E, ws, heads = cfg.swin_embed_dim, cfg.window_size, cfg.heads
    I.conv(p + ".patch_embed.proj", ch, E, 1)
    I.norm(p + ".patch_embed.norm", E)
    for d in range(cfg.swin_depth):
        b = f"{p}.blocks.{d}"
        I.norm(b + ".norm1", E)
        I.sd[b + ".attn.relative_position_bias_table"] = I.randn((2 * ws - 1) ** 2, heads, std=0.2)
        I.linear(b + ".attn.qkv", E, 3 * E)
        I.linear(b + ".attn.proj", E, E, gain=0.5)
        I.norm(b + ".norm2", E)
        I.linear(b + ".mlp.fc1", E, cfg.mlp_ratio * E)
        I.linear(b + ".mlp.fc2", cfg.mlp_ratio * E, E, gain=0.5)
    I.conv(p + ".patch_unembed.proj", E, ch, 1, gain=0.5)


def unet_layout(cfg: SinSRConfig):
    """Walk the UNet topology once; returns a list of (kind, prefix, meta) used by weight
    init, the oracle and the device graph alike (topology only - no arithmetic)."""
    mc = cfg.model_channels
    plan = {"input": [], "middle": [], "output": []}
    ch = mc
    chans = [ch]
    plan["input"].append(("conv_in", "input_blocks.0.0", (2 * cfg.latent_ch, mc)))
    idx = 1
    nlev = len(cfg.channel_mult)
    for level, mult in enumerate(cfg.channel_mult):
        for jj in range(cfg.num_res_blocks):
            ops = [("res", f"input_blocks.{idx}.0", (ch, mult * mc))]
            ch = mult * mc
            if jj == 0:
                ops.append(("swin", f"input_blocks.{idx}.1", (ch,)))
            plan["input"].append(("seq", f"input_blocks.{idx}", ops))
            chans.append(ch)
            idx += 1
        if level != nlev - 1:
            plan["input"].append(("down", f"input_blocks.{idx}.0.op", (ch, ch)))
            chans.append(ch)
            idx += 1
    plan["middle"] = [("res", "middle_block.0", (ch, ch)), ("swin", "middle_block.1", (ch,)),
                      ("res", "middle_block.2", (ch, ch))]
    oidx = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            ops = [("res", f"output_blocks.{oidx}.0", (ch + ich, mult * mc))]
            ch = mult * mc
            k = 1
            if i == 0:
                ops.append(("swin", f"output_blocks.{oidx}.{k}", (ch,)))
                k += 1
            if level and i == cfg.num_res_blocks:
                ops.append(("up", f"output_blocks.{oidx}.{k}.conv", (ch, ch)))
            plan["output"].append(("seq", f"output_blocks.{oidx}", ops))
            oidx += 1
    plan["out_ch"] = ch
    return plan


def _ae_resblock(I: _Init, p: str, cin: int, cout: int):
    I.norm(p + ".norm1", cin)
    I.conv(p + ".conv1", cin, cout, 3)
    I.norm(p + ".norm2", cout)
    I.conv(p + ".conv2", cout, cout, 3, gain=0.5)
    if cin != cout:
        I.conv(p + ".nin_shortcut", cin, cout, 1)


def make_sinsr_weights(cfg: SinSRConfig = SinSRConfig(), seed: int = 0) -> Dict[str, torch.Tensor]:
    I = _Init(seed)
    mc, emb = cfg.model_channels, cfg.time_embed_dim
    # ---------------- UNet ("model.") ----------------
    I.linear("model.time_embed.0", mc, emb)
    I.linear("model.time_embed.2", emb, emb)
    plan = unet_layout(cfg)

    def emit(ops):
        for kind, p, meta in ops:
            p = "model." + p
            if kind == "conv_in":
                I.conv(p, meta[0], meta[1], 3)
            elif kind == "res":
                _unet_resblock(I, p, meta[0], meta[1], emb)
            elif kind == "swin":
                _swin_layer(I, p, meta[0], cfg)
            elif kind in ("down", "up"):
                I.conv(p, meta[0], meta[1], 3)

    for kind, p, meta in plan["input"]:
        emit(meta if kind == "seq" else [(kind, p, meta)])
    emit(plan["middle"])
    for kind, p, meta in plan["output"]:
        emit(meta)
    I.norm("model.out.0", plan["out_ch"])
    I.conv("model.out.2", plan["out_ch"], cfg.latent_ch, 3, gain=0.5)

    # ---------------- VQ-f4 autoencoder ("ae.") ----------------
    ch, mults, nrb = cfg.ae_ch, cfg.ae_ch_mult, cfg.ae_num_res_blocks
    I.conv("ae.encoder.conv_in", 3, ch, 3)
    cin = ch
    for lvl, m in enumerate(mults):
        for b in range(nrb):
            _ae_resblock(I, f"ae.encoder.down.{lvl}.block.{b}", cin, ch * m)
            cin = ch * m
        if lvl != len(mults) - 1:
            I.conv(f"ae.encoder.down.{lvl}.downsample.conv", cin, cin, 3)
    _ae_resblock(I, "ae.encoder.mid.block_1", cin, cin)
    _ae_resblock(I, "ae.encoder.mid.block_2", cin, cin)
    I.norm("ae.encoder.norm_out", cin)
    I.conv("ae.encoder.conv_out", cin, cfg.z_channels, 3)
    I.conv("ae.quant_conv", cfg.z_channels, cfg.embed_dim, 1)
    I.sd["ae.quantize.embedding.weight"] = (torch.rand(cfg.n_embed, cfg.embed_dim, generator=I.g) * 2 - 1) * 1.5
    I.conv("ae.post_quant_conv", cfg.embed_dim, cfg.z_channels, 1)
    cin = ch * mults[-1]
    I.conv("ae.decoder.conv_in", cfg.z_channels, cin, 3)
    _ae_resblock(I, "ae.decoder.mid.block_1", cin, cin)
    _ae_resblock(I, "ae.decoder.mid.block_2", cin, cin)
    for lvl in reversed(range(len(mults))):
        for b in range(nrb + 1):
            _ae_resblock(I, f"ae.decoder.up.{lvl}.block.{b}", cin, ch * mults[lvl])
            cin = ch * mults[lvl]
        if lvl != 0:
            I.conv(f"ae.decoder.up.{lvl}.upsample.conv", cin, cin, 3)
    I.norm("ae.decoder.norm_out", cin)
    I.conv("ae.decoder.conv_out", cin, 3, 3, gain=0.5)
    return I.sd


def relative_position_index(ws: int) -> torch.Tensor:
    """Swin relative-position index [ws*ws, ws*ws] into the (2ws-1)^2 bias table."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def timestep_embedding(t: int, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = float(t) * freqs
    return torch.cat([torch.cos(args), torch.sin(args)])[None, :]


def frame_noise(cfg: SinSRConfig, seed: int, frame_index: int, hp: int, wp: int) -> torch.Tensor:
    """Sampler noise keyed on the GLOBAL frame index (not the rank / chunk), so results do
    not depend on the GPU count (fixes elvis.py:3127's seed+chunk_index dependence).
    Generated on the host with a torch.Generator and shipped to the device as an explicit
    input tensor (SURVEY.md 7.2 "Stochastic sampler")."""
    g = torch.Generator().manual_seed(int(seed) * 1000003 + int(frame_index))
    return torch.randn(1, cfg.latent_ch, hp, wp, generator=g, dtype=torch.float32)


# =========================================================================================
# The other two model slots of ELVIS v2.  Neither model has source,
# weights or tests in the reference (README.md:11-25 only names them), so - like SinSR above - the
# architectures are the build's documented choice and the weights are seeded synthetic.

@dataclass(frozen=True)
class DCNRestorerConfig:
    """LaplacianVCAR slot (ELVIS v2 DCT): STDF-style spatio-temporal deformable restorer.
    T = 2*radius+1 planes -> small U-Net -> 9 offsets (+ mask) per plane -> DCNv2(T -> feat, 3x3,
    deformable_groups = T) -> plain 3x3 CNN -> residual on the centre plane."""
    radius: int = 3
    off_ch: int = 32
    feat: int = 64
    qe_layers: int = 3

    @property
    def t(self) -> int:
        return 2 * self.radius + 1


@dataclass(frozen=True)
class SwinDeblurConfig:
    """SwinTormer slot (ELVIS v2 Blur): Restormer-like U-shape whose transformer blocks use Swin
    (shifted-)window attention (window 8, head_dim 32) and a GELU MLP."""
    ch: int = 64
    blocks: Tuple[int, ...] = (2, 2, 2)     # encoder level 1, level 2, bottleneck (decoder mirrors 2,1)
    mlp_ratio: int = 2
    window_size: int = 8
    head_dim: int = 32

    @property
    def align(self) -> int:
        return self.window_size * 4


def make_dcn_weights(cfg: DCNRestorerConfig = DCNRestorerConfig(), seed: int = 0) -> Dict[str, torch.Tensor]:
    I = _Init(seed + 101)
    t, oc, f = cfg.t, cfg.off_ch, cfg.feat
    I.conv("off.c1", t, oc, 3)
    I.conv("off.d1", oc, oc, 3)
    I.conv("off.d2", oc, oc, 3)
    I.conv("off.u1", oc, oc, 3)
    I.conv("off.f", 2 * oc, oc, 3)
    I.conv("off.om", oc, 27 * t, 3, gain=0.5)
    I.conv("dcn", t, f, 3)
    for i in range(cfg.qe_layers):
        I.conv(f"qe.{i}", f, f, 3)
    I.conv("qe.out", f, 1, 3, gain=0.3)
    return I.sd


def make_deblur_weights(cfg: SwinDeblurConfig = SwinDeblurConfig(), seed: int = 0) -> Dict[str, torch.Tensor]:
    I = _Init(seed + 202)
    C, ws = cfg.ch, cfg.window_size

    def blocks(prefix, ch, n):
        heads = ch // cfg.head_dim
        for i in range(n):
            b = f"{prefix}.{i}"
            I.norm(b + ".norm1", ch)
            I.sd[b + ".attn.relative_position_bias_table"] = I.randn((2 * ws - 1) ** 2, heads, std=0.2)
            I.linear(b + ".attn.qkv", ch, 3 * ch)
            I.linear(b + ".attn.proj", ch, ch, gain=0.5)
            I.norm(b + ".norm2", ch)
            I.linear(b + ".mlp.fc1", ch, cfg.mlp_ratio * ch)
            I.linear(b + ".mlp.fc2", cfg.mlp_ratio * ch, ch, gain=0.5)

    I.conv("embed", 3, C, 3)
    blocks("enc1", C, cfg.blocks[0])
    I.conv("down1", C, 2 * C, 3)
    blocks("enc2", 2 * C, cfg.blocks[1])
    I.conv("down2", 2 * C, 4 * C, 3)
    blocks("mid", 4 * C, cfg.blocks[2])
    I.conv("up2", 4 * C, 2 * C, 3)
    I.conv("red2", 4 * C, 2 * C, 1)
    blocks("dec2", 2 * C, cfg.blocks[1])
    I.conv("up1", 2 * C, C, 3)
    I.conv("red1", 2 * C, C, 1)
    blocks("dec1", C, cfg.blocks[0])
    I.conv("out", C, 3, 3, gain=0.3)
    return I.sd
