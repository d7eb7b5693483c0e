"""CPU oracle (PyTorch fp32, NCHW) for the SinSR-style 4x super-resolver forward pass.

TEST INFRASTRUCTURE - NOT PRODUCT CODE (see oracle/glue_ref.py header for the rule).

PARITY UNPINNED: the reference contains no SinSR source, call site, checkpoint, test or
fixture (the only mention is README.md:27-28,46,50).  This file is
therefore the build's own restatement of the *published* SinSR / ResShift inference
recipe (single-step `x_T = z_y + kappa*sqrt(eta_T)*eps`, `UNetModelSwin`, LDM VQ-f4
autoencoder; hyper-parameters in `elvis_amd.weights.SinSRConfig`) with seeded synthetic
weights.  It checks GPU-vs-CPU numerics of the hand-written kernels, not fidelity to an
upstream checkpoint.  Documented deviations from upstream: no mid-block attention in the
autoencoder, alignment padding applied at the UNet only.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from elvis_amd.weights import (SinSRConfig, frame_noise, relative_position_index,  # noqa: F401
                               timestep_embedding, unet_layout)

SD = Dict[str, torch.Tensor]


def conv(sd: SD, p: str, x, stride=1, padding=None):
    w = sd[p + ".weight"]
    if padding is None:
        padding = w.shape[-1] // 2
    return F.conv2d(x, w, sd[p + ".bias"], stride=stride, padding=padding)


def gn(sd: SD, p: str, x, groups: int, eps: float):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


# ------------------------------------------------------------------ swin
def window_partition(x, ws):
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)


def window_reverse(win, ws, h, w):
    b = win.shape[0] // ((h // ws) * (w // ws))
    x = win.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


def shift_mask(h, w, ws, shift):
    img = torch.zeros(1, h, w, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).squeeze(-1)
    am = mw[:, None, :] - mw[:, :, None]
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def swin_block(sd: SD, p: str, x, h, w, cfg: SinSRConfig, shift: int):
    E, ws, heads = cfg.swin_embed_dim, cfg.window_size, cfg.heads
    b = x.shape[0]
    shortcut = x
    y = F.layer_norm(x, (E,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5).view(b, h, w, E)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    win = window_partition(y, ws)
    qkv = F.linear(win, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
    nw, n = win.shape[0], ws * ws
    qkv = qkv.view(nw, n, 3, heads, E // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * ((E // heads) ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    rpi = relative_position_index(ws)
    bias = sd[p + ".attn.relative_position_bias_table"][rpi.view(-1)].view(n, n, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if shift:
        m = shift_mask(h, w, ws, shift)
        attn = attn.view(b, m.shape[0], heads, n, n) + m[None, :, None]
        attn = attn.view(-1, heads, n, n)
    attn = attn.softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(nw, n, E)
    o = F.linear(o, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    y = window_reverse(o, ws, h, w)
    if shift:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + y.view(b, h * w, E)
    y = F.layer_norm(x, (E,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    y = F.linear(y, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    return x + y


def swin_layer(sd: SD, p: str, x, cfg: SinSRConfig):
    b, c, h, w = x.shape
    E = cfg.swin_embed_dim
    y = conv(sd, p + ".patch_embed.proj", x).flatten(2).transpose(1, 2)
    y = F.layer_norm(y, (E,), sd[p + ".patch_embed.norm.weight"], sd[p + ".patch_embed.norm.bias"], 1e-5)
    for d in range(cfg.swin_depth):
        y = swin_block(sd, f"{p}.blocks.{d}", y, h, w, cfg, 0 if d % 2 == 0 else cfg.window_size // 2)
    y = y.transpose(1, 2).reshape(b, E, h, w)
    return conv(sd, p + ".patch_unembed.proj", y)


# ------------------------------------------------------------------ UNet
def unet_resblock(sd: SD, p: str, x, emb, cfg: SinSRConfig):
    g = cfg.gn_groups
    h = conv(sd, p + ".in_layers.2", F.silu(gn(sd, p + ".in_layers.0", x, g, 1e-5)))
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    scale, shift = e[:, :, None, None].chunk(2, dim=1)
    h = gn(sd, p + ".out_layers.0", h, g, 1e-5) * (1 + scale) + shift
    h = conv(sd, p + ".out_layers.3", F.silu(h))
    if (p + ".skip_connection.weight") in sd:
        x = conv(sd, p + ".skip_connection", x)
    return x + h


def unet_forward(sd: SD, cfg: SinSRConfig, x, t: int):
    """x: [B, 2*latent_ch, H, W] = cat(scaled x_T, lq).  Returns predicted x0 latent."""
    sd = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    emb = timestep_embedding(t, cfg.model_channels)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    plan = unet_layout(cfg)

    def run(ops, h):
        for kind, p, meta in ops:
            if kind == "conv_in":
                h = conv(sd, p, h)
            elif kind == "res":
                h = unet_resblock(sd, p, h, emb, cfg)
            elif kind == "swin":
                h = swin_layer(sd, p, h, cfg)
            elif kind == "down":
                h = conv(sd, p, h, stride=2)
            elif kind == "up":
                h = conv(sd, p, F.interpolate(h, scale_factor=2, mode="nearest"))
        return h

    hs = []
    h = x
    for kind, p, meta in plan["input"]:
        h = run(meta if kind == "seq" else [(kind, p, meta)], h)
        hs.append(h)
    h = run(plan["middle"], h)
    for kind, p, meta in plan["output"]:
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(meta, h)
    h = F.silu(gn(sd, "out.0", h, cfg.gn_groups, 1e-5))
    return conv(sd, "out.2", h)


# ------------------------------------------------------------------ VQ-f4 autoencoder
def ae_resblock(sd: SD, p: str, x, g):
    h = conv(sd, p + ".conv1", F.silu(gn(sd, p + ".norm1", x, g, 1e-6)))
    h = conv(sd, p + ".conv2", F.silu(gn(sd, p + ".norm2", h, g, 1e-6)))
    if (p + ".nin_shortcut.weight") in sd:
        x = conv(sd, p + ".nin_shortcut", x)
    return x + h


def ae_encode(sd: SD, cfg: SinSRConfig, x):
    g = cfg.gn_groups
    h = conv(sd, "ae.encoder.conv_in", x)
    for lvl in range(len(cfg.ae_ch_mult)):
        for b in range(cfg.ae_num_res_blocks):
            h = ae_resblock(sd, f"ae.encoder.down.{lvl}.block.{b}", h, g)
        if lvl != len(cfg.ae_ch_mult) - 1:
            h = conv(sd, f"ae.encoder.down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = ae_resblock(sd, "ae.encoder.mid.block_1", h, g)
    h = ae_resblock(sd, "ae.encoder.mid.block_2", h, g)
    h = conv(sd, "ae.encoder.conv_out", F.silu(gn(sd, "ae.encoder.norm_out", h, g, 1e-6)))
    return conv(sd, "ae.quant_conv", h)


def vq_quantize(sd: SD, z):
    """Nearest codebook entry: argmin_k sum_c (z_c - e_kc)^2 evaluated as ((d0*d0)+d1*d1)+d2*d2 in
    plain fp32 (no FMA contraction), first index wins ties - the same evaluation order as the HIP
    kernel, so identical inputs give identical codes."""
    cb = sd["ae.quantize.embedding.weight"]
    b, c, h, w = z.shape
    zf = z.permute(0, 2, 3, 1).reshape(-1, c)
    idx = torch.empty(zf.shape[0], dtype=torch.long)
    for s0 in range(0, zf.shape[0], 4096):
        zc = zf[s0:s0 + 4096]
        d = None
        for k in range(c):
            df = zc[:, k:k + 1] - cb[None, :, k]
            sq = df * df
            d = sq if d is None else d + sq
        idx[s0:s0 + 4096] = d.argmin(1)
    return cb[idx].view(b, h, w, c).permute(0, 3, 1, 2).contiguous(), idx.view(b, h, w)


def ae_decode(sd: SD, cfg: SinSRConfig, z, quantize: Optional[bool] = None):
    g = cfg.gn_groups
    if cfg.quantize if quantize is None else quantize:
        z, _ = vq_quantize(sd, z)
    h = conv(sd, "ae.post_quant_conv", z)
    h = conv(sd, "ae.decoder.conv_in", h)
    h = ae_resblock(sd, "ae.decoder.mid.block_1", h, g)
    h = ae_resblock(sd, "ae.decoder.mid.block_2", h, g)
    for lvl in reversed(range(len(cfg.ae_ch_mult))):
        for b in range(cfg.ae_num_res_blocks + 1):
            h = ae_resblock(sd, f"ae.decoder.up.{lvl}.block.{b}", h, g)
        if lvl != 0:
            h = conv(sd, f"ae.decoder.up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2, mode="nearest"))
    return conv(sd, "ae.decoder.conv_out", F.silu(gn(sd, "ae.decoder.norm_out", h, g, 1e-6)))


# ------------------------------------------------------------------ full single-step SR
def padded_latent_shape(cfg: SinSRConfig, h: int, w: int):
    a = cfg.unet_align
    return (math.ceil(h / a) * a, math.ceil(w / a) * a)


def sinsr_forward(sd: SD, cfg: SinSRConfig, lr_u8: torch.Tensor, noise: torch.Tensor, return_stages: bool = False):
    """lr_u8: [h,w,3] uint8 (RGB).  noise: [1,latent_ch,Hp,Wp] fp32 for the PADDED latent.
    Returns fp32 [4h,4w,3] in [0,1] (pre-quantisation) - `to_u8` gives the frame."""
    h, w, _ = lr_u8.shape
    y = (lr_u8.permute(2, 0, 1)[None].float() / 255.0) * 2.0 - 1.0
    y_up = F.interpolate(y, scale_factor=cfg.sf, mode="bicubic", align_corners=False)
    z_y = ae_encode(sd, cfg, y_up)
    hp, wp = padded_latent_shape(cfg, h, w)
    pad = (0, wp - w, 0, hp - h)
    if hp != h or wp != w:
        z_p = F.pad(z_y, pad, mode="reflect")
        y_p = F.pad(y, pad, mode="reflect")
    else:
        z_p, y_p = z_y, y
    eta_T = cfg.etas_end
    x_T = z_p + cfg.kappa * math.sqrt(eta_T) * noise
    inp = x_T / math.sqrt(eta_T * cfg.kappa ** 2 + 1.0)
    z0 = unet_forward(sd, cfg, torch.cat([inp, y_p], dim=1), cfg.steps - 1)
    z0 = z0[:, :, :h, :w].contiguous()
    x = ae_decode(sd, cfg, z0)
    out = (x * 0.5 + 0.5).clamp(0.0, 1.0)[0].permute(1, 2, 0).contiguous()
    if return_stages:
        return out, {"y_up": y_up, "z_y": z_y, "x_T": x_T, "z0": z0, "dec": x}
    return out


def to_u8(img01: torch.Tensor) -> torch.Tensor:
    return torch.round(img01 * 255.0).clamp(0, 255).to(torch.uint8)
