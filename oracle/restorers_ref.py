"""CPU oracle (PyTorch fp32) for the DCT-slot (LaplacianVCAR-style DCNv2) and Blur-slot
(SwinTormer-style) restorers.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

PARITY UNPINNED: the reference has no source, weights, call site or test for either model
(SURVEY.md F1, rows a7/a8); these are the build's own architectures (elvis_amd.weights configs) with
seeded synthetic weights.  DCNv2 is restated with an explicit bilinear gather (torchvision's
deform_conv2d is not installed here), following the mmcv/DCNv2 conventions: offset channel
(g*9+k)*2+{dy,dx}, mask channel g*9+k after a sigmoid, zero for out-of-image corners.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from elvis_amd.weights import DCNRestorerConfig, SwinDeblurConfig, relative_position_index
from oracle.sinsr_ref import shift_mask, window_partition, window_reverse


def conv(sd, p, x, stride=1, padding=None):
    w = sd[p + ".weight"]
    return F.conv2d(x, w, sd[p + ".bias"], stride=stride, padding=w.shape[-1] // 2 if padding is None else padding)


def dcnv2(x, offset, mask, weight, bias, groups):
    """x [N,C,H,W]; offset [N,18G,H,W]; mask [N,9G,H,W] (already in [0,1]); weight [Co,C,3,3]."""
    n, c, h, w = x.shape
    cpg = c // groups
    ys = torch.arange(h, dtype=torch.float32).view(1, h, 1)
    xs = torch.arange(w, dtype=torch.float32).view(1, 1, w)
    cols = []
    for ch in range(c):
        g = ch // cpg
        for k in range(9):
            dy, dx = offset[:, (g * 9 + k) * 2], offset[:, (g * 9 + k) * 2 + 1]
            sy = ys + (k // 3 - 1) + dy
            sx = xs + (k % 3 - 1) + dx
            y0, x0 = torch.floor(sy), torch.floor(sx)
            ly, lx = sy - y0, sx - x0
            y0, x0 = y0.long(), x0.long()
            plane = x[:, ch]

            def tap(yy, xx):
                ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                v = plane.reshape(n, -1).gather(1, (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).reshape(n, -1)).view(n, h, w)
                return torch.where(ok, v, torch.zeros_like(v))

            v = (1 - ly) * ((1 - lx) * tap(y0, x0) + lx * tap(y0, x0 + 1)) + ly * ((1 - lx) * tap(y0 + 1, x0) + lx * tap(y0 + 1, x0 + 1))
            cols.append(v * mask[:, g * 9 + k])
    col = torch.stack(cols, 1)                                  # [N, C*9, H, W]
    out = torch.einsum("ok,nkhw->nohw", weight.reshape(weight.shape[0], -1), col)
    return out + bias.view(1, -1, 1, 1)


def dcn_restorer_forward(sd, cfg: DCNRestorerConfig, planes):
    """planes [N, T, H, W] in [0,1] (temporal window of one colour plane) -> residual [N,1,H,W]."""
    t = cfg.t
    c1 = F.relu(conv(sd, "off.c1", planes))
    d1 = F.relu(conv(sd, "off.d1", c1, stride=2))
    d2 = F.relu(conv(sd, "off.d2", d1))
    u1 = F.relu(conv(sd, "off.u1", F.interpolate(d2, scale_factor=2, mode="nearest")))
    f = F.relu(conv(sd, "off.f", torch.cat([u1, c1], 1)))
    om = conv(sd, "off.om", f)
    offset, mask = om[:, :18 * t], torch.sigmoid(om[:, 18 * t:])
    feat = F.relu(dcnv2(planes, offset, mask, sd["dcn.weight"], sd["dcn.bias"], t))
    for i in range(cfg.qe_layers):
        feat = F.relu(conv(sd, f"qe.{i}", feat))
    return conv(sd, "qe.out", feat)


def dcn_restore_frames(sd, cfg: DCNRestorerConfig, frames_u8: torch.Tensor):
    """frames [F,H,W,3] u8 -> restored [F,H,W,3] u8 (every colour plane with its 2R+1 temporal window,
    edge-replicated); also returns the pre-quantisation float image."""
    nf = frames_u8.shape[0]
    x = frames_u8.float() / 255.0
    outs = []
    for f in range(nf):
        idx = [min(max(f + d, 0), nf - 1) for d in range(-cfg.radius, cfg.radius + 1)]
        planes = x[idx].permute(3, 0, 1, 2)                     # [3, T, H, W]
        res = dcn_restorer_forward(sd, cfg, planes)[:, 0]       # [3, H, W]
        outs.append((x[f].permute(2, 0, 1) + res).clamp(0, 1).permute(1, 2, 0))
    f32 = torch.stack(outs)
    return torch.round(f32 * 255).to(torch.uint8), f32


# ------------------------------------------------------------------ SwinTormer-style deblur
def swin_block(sd, p, x, h, w, ch, cfg: SwinDeblurConfig, shift):
    ws, hd = cfg.window_size, cfg.head_dim
    heads = ch // hd
    b = x.shape[0]
    y = F.layer_norm(x, (ch,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5).view(b, h, w, ch)
    if shift:
        y = torch.roll(y, (-shift, -shift), (1, 2))
    win = window_partition(y, ws)
    nw, n = win.shape[0], ws * ws
    qkv = F.linear(win, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"]).view(nw, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    attn = (qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1)
    bias = sd[p + ".attn.relative_position_bias_table"][relative_position_index(ws).view(-1)].view(n, n, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if shift:
        m = shift_mask(h, w, ws, shift)
        attn = (attn.view(b, m.shape[0], heads, n, n) + m[None, :, None]).view(-1, heads, n, n)
    o = (attn.softmax(-1) @ qkv[2]).transpose(1, 2).reshape(nw, n, ch)
    o = F.linear(o, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    y = window_reverse(o, ws, h, w)
    if shift:
        y = torch.roll(y, (shift, shift), (1, 2))
    x = x + y.view(b, h * w, ch)
    y = F.layer_norm(x, (ch,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    y = F.linear(F.gelu(F.linear(y, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])), sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    return x + y


def stage(sd, prefix, x, nblocks, cfg):
    b, ch, h, w = x.shape
    t = x.flatten(2).transpose(1, 2)
    for i in range(nblocks):
        t = swin_block(sd, f"{prefix}.{i}", t, h, w, ch, cfg, 0 if i % 2 == 0 else cfg.window_size // 2)
    return t.transpose(1, 2).reshape(b, ch, h, w)


def deblur_forward(sd, cfg: SwinDeblurConfig, img01):
    """img01 [N,3,H,W] in [0,1], H,W multiples of cfg.align -> restored [N,3,H,W] (unclamped)."""
    e1 = stage(sd, "enc1", conv(sd, "embed", img01), cfg.blocks[0], cfg)
    e2 = stage(sd, "enc2", conv(sd, "down1", e1, stride=2), cfg.blocks[1], cfg)
    m = stage(sd, "mid", conv(sd, "down2", e2, stride=2), cfg.blocks[2], cfg)
    u2 = conv(sd, "up2", F.interpolate(m, scale_factor=2, mode="nearest"))
    d2 = stage(sd, "dec2", conv(sd, "red2", torch.cat([u2, e2], 1)), cfg.blocks[1], cfg)
    u1 = conv(sd, "up1", F.interpolate(d2, scale_factor=2, mode="nearest"))
    d1 = stage(sd, "dec1", conv(sd, "red1", torch.cat([u1, e1], 1)), cfg.blocks[0], cfg)
    return img01 + conv(sd, "out", d1)


def deblur_restore_frames(sd, cfg: SwinDeblurConfig, frames_u8: torch.Tensor):
    """frames [F,H,W,3] u8 RGB -> (restored u8, float image); reflect-pads bottom/right to cfg.align."""
    n, h, w, _ = frames_u8.shape
    a = cfg.align
    hp, wp = (h + a - 1) // a * a, (w + a - 1) // a * a
    x = frames_u8.permute(0, 3, 1, 2).float() / 255.0
    if hp != h or wp != w:
        x = F.pad(x, (0, wp - w, 0, hp - h), mode="reflect")
    y = deblur_forward(sd, cfg, x)[:, :, :h, :w].clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    return torch.round(y * 255).to(torch.uint8), y
