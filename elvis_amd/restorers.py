"""Device graphs of the other two ELVIS v2 model slots:

* `DCNRestorer`  - LaplacianVCAR slot (ELVIS v2 DCT): STDF-style restorer built on the hand-written
  DCNv2 kernel (README.md:11-16 names only an absent `ops/dcn` CUDA build).
* `SwinDeblur`   - SwinTormer slot (ELVIS v2 Blur): Restormer-like U-shape with Swin window
  attention, filling the slot of `restore_images_batch` (elvis.py:2963-2970).

Both return restored uint8 frames for `rounds_recompose_device` / `restore_frames_rounds`
(the round loop of elvis.py:2947-2981).  Architectures and weights are the build's own
(`weights.DCNRestorerConfig`, `weights.SwinDeblurConfig`; seeded synthetic), see DESIGN.md.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .ops import Act, PackedConv, PackedUpConv
from .weights import (DCNRestorerConfig, SwinDeblurConfig, make_dcn_weights, make_deblur_weights)

RELU, GELU = 3, 1


def _pc(sd, name, dtype, device, cin, cin2=0):
    return PackedConv(sd[name + ".weight"], sd[name + ".bias"], dtype, device, cin, cin2)


class _DownConv:
    """conv3x3(x, stride=2, padding=1): the space-to-depth form on the halo-tile kernel where it exists (f16, whole
    32-channel chunks: ops.PackedDownConv with pad1), the generic strided kernel otherwise."""

    def __init__(self, sd, name, dtype, device, cin):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        self.s2d = ops.PackedDownConv(w, b, dtype, device, cin, pad1=True) if ops.PackedDownConv.supported_pad1(dtype, cin, w.shape[0]) else None
        self.direct = None if self.s2d is not None else PackedConv(w, b, dtype, device, cin)

    def __call__(self, x: Act, act: int = 0) -> Act:
        if self.s2d is not None and x.h % 2 == 0 and x.w % 2 == 0:
            return self.s2d(x, act=act)
        if self.direct is None:
            raise ValueError("down conv: odd input size for the space-to-depth form")
        return self.direct(x, stride=2, act=act)


class DCNRestorer:
    def __init__(self, cfg: DCNRestorerConfig = DCNRestorerConfig(), state_dict: Optional[Dict] = None,
                 device="cuda:0", dtype=torch.float16, weight_seed: int = 0):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        sd = state_dict if state_dict is not None else make_dcn_weights(cfg, weight_seed)
        t, oc, f = cfg.t, cfg.off_ch, cfg.feat
        dev = self.device
        with torch.cuda.device(dev):
            self.c1 = _pc(sd, "off.c1", dtype, dev, t)
            self.d1 = _DownConv(sd, "off.d1", dtype, dev, oc)
            self.d2 = _pc(sd, "off.d2", dtype, dev, oc)
            self.u1 = PackedUpConv(sd["off.u1.weight"], sd["off.u1.bias"], dtype, dev, oc)
            self.f = _pc(sd, "off.f", dtype, dev, oc, oc)
            self.om = _pc(sd, "off.om", dtype, dev, oc)
            self.dcn_w = sd["dcn.weight"].to(device=dev, dtype=dtype).contiguous()
            self.dcn_b = sd["dcn.bias"].to(device=dev, dtype=torch.float32).contiguous()
            self.qe = [_pc(sd, f"qe.{i}", dtype, dev, f) for i in range(cfg.qe_layers)]
            self.qe_out = _pc(sd, "qe.out", dtype, dev, f)

    def forward_planes(self, planes: Act) -> Act:
        """planes [N, H, W, T] in [0,1] -> residual [N, H, W, 1]."""
        cfg = self.cfg
        c1 = self.c1(planes, act=RELU)
        d1 = self.d1(c1, act=RELU)
        d2 = self.d2(d1, act=RELU)
        u1 = self.u1(d2, act=RELU)
        f = self.f(u1, c1, act=RELU)
        om = self.om(f)
        feat = ops.dcnv2(planes, om, self.dcn_w, self.dcn_b, cfg.t, cfg.feat, mask_sigmoid=True, act=RELU)
        for q in self.qe:
            feat = q(feat, act=RELU)
        return self.qe_out(feat)

    def restore(self, frames_d: torch.Tensor, chunk: int = 2, frame_range=None) -> torch.Tensor:
        """[F,H,W,3] u8 on the device -> restored [F,H,W,3] u8.  Every colour plane is restored from
        its 2R+1 temporal window (edge-replicated), `chunk` frames (3*chunk planes) per invocation.
        `frame_range=(a, b)` restores frames [a, b) only - their windows still read the whole clip -
        and returns [b-a,H,W,3] (the host-to-host pipeline restores a clip range by range)."""
        nf, h, w, _ = frames_d.shape
        if h % 2 or w % 2:
            raise ValueError("DCNRestorer needs even H and W")
        a, b = (0, nf) if frame_range is None else frame_range
        if not 0 <= a <= b <= nf:
            raise ValueError(f"frame_range {frame_range} outside the clip's {nf} frames")
        out = torch.empty((b - a, h, w, 3), dtype=torch.uint8, device=frames_d.device)
        with torch.cuda.device(self.device):
            for f0 in range(a, b, chunk):
                nsel = min(chunk, b - f0)
                planes = ops.temporal_stack(frames_d, f0, nsel, self.cfg.radius, self.dtype)
                res = self.forward_planes(planes)
                ops.plane_merge(frames_d, res, f0, nsel, out=out[f0 - a:f0 - a + nsel])
        return out


class _SwinStage:
    def __init__(self, sd, prefix, ch, nblocks, cfg: SwinDeblurConfig, dtype, device):
        self.cfg, self.ch, self.heads = cfg, ch, ch // cfg.head_dim
        f32 = dict(device=device, dtype=torch.float32)
        self.blocks = []
        from .sinsr import _SWIN_FUSE
        fuse = _SWIN_FUSE and ops.SwinFused.supported(dtype, ch, 3 * ch) and ops.SwinFused.supported(dtype, ch, cfg.mlp_ratio * ch)
        for i in range(nblocks):
            b = f"{prefix}.{i}"
            lin = lambda n: PackedConv(sd[n + ".weight"][:, :, None, None], sd[n + ".bias"], dtype, device,
                                       sd[n + ".weight"].shape[1])
            w = lambda n: sd[b + n]
            self.blocks.append(dict(
                qkv_f=ops.SwinFused(w(".norm1.weight"), w(".norm1.bias"), w(".attn.qkv.weight"), w(".attn.qkv.bias"), device=device) if fuse else None,
                # ... and the attention output projection folded in front of the MLP: y' = y + proj(a) never reaches HBM
                mlp_f=ops.SwinFused(w(".norm2.weight"), w(".norm2.bias"), w(".mlp.fc1.weight"), w(".mlp.fc1.bias"),
                                    w(".mlp.fc2.weight"), w(".mlp.fc2.bias"), device=device,
                                    **(dict(proj_w=w(".attn.proj.weight"), proj_b=w(".attn.proj.bias")) if ops.SwinFused.proj_pays(ch) else {})) if fuse else None,
                n1=(sd[b + ".norm1.weight"].to(**f32), sd[b + ".norm1.bias"].to(**f32)),
                n2=(sd[b + ".norm2.weight"].to(**f32), sd[b + ".norm2.bias"].to(**f32)),
                qkv=lin(b + ".attn.qkv"), proj=lin(b + ".attn.proj"), fc1=lin(b + ".mlp.fc1"), fc2=lin(b + ".mlp.fc2"),
                table=sd[b + ".attn.relative_position_bias_table"].to(**f32).contiguous(),
                shift=0 if i % 2 == 0 else cfg.window_size // 2))

    def __call__(self, y: Act) -> Act:
        cfg = self.cfg
        for b in self.blocks:
            qkv = b["qkv_f"](y) if b["qkv_f"] is not None else b["qkv"](ops.layernorm(y, *b["n1"]))
            a = ops.window_attention(qkv, self.heads, cfg.head_dim, cfg.window_size, b["shift"], b["table"],
                                     cfg.head_dim ** -0.5)
            del qkv
            if b["mlp_f"] is not None:
                y = b["mlp_f"](a, y) if b["mlp_f"].proj else b["mlp_f"](b["proj"](a, residual=y))
            else:
                y = b["proj"](a, residual=y)
                t = ops.layernorm(y, *b["n2"])
                t = b["fc1"](t, act=GELU)
                y = b["fc2"](t, residual=y)
        return y


class SwinDeblur:
    def __init__(self, cfg: SwinDeblurConfig = SwinDeblurConfig(), state_dict: Optional[Dict] = None,
                 device="cuda:0", dtype=torch.float16, weight_seed: int = 0):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        sd = state_dict if state_dict is not None else make_deblur_weights(cfg, weight_seed)
        C, dev = cfg.ch, self.device
        with torch.cuda.device(dev):
            self.embed = _pc(sd, "embed", dtype, dev, 3)
            self.enc1 = _SwinStage(sd, "enc1", C, cfg.blocks[0], cfg, dtype, dev)
            self.down1 = _DownConv(sd, "down1", dtype, dev, C)
            self.enc2 = _SwinStage(sd, "enc2", 2 * C, cfg.blocks[1], cfg, dtype, dev)
            self.down2 = _DownConv(sd, "down2", dtype, dev, 2 * C)
            self.mid = _SwinStage(sd, "mid", 4 * C, cfg.blocks[2], cfg, dtype, dev)
            self.up2 = PackedUpConv(sd["up2.weight"], sd["up2.bias"], dtype, dev, 4 * C)
            self.red2 = _pc(sd, "red2", dtype, dev, 2 * C, 2 * C)
            self.dec2 = _SwinStage(sd, "dec2", 2 * C, cfg.blocks[1], cfg, dtype, dev)
            self.up1 = PackedUpConv(sd["up1.weight"], sd["up1.bias"], dtype, dev, 2 * C)
            self.red1 = _pc(sd, "red1", dtype, dev, C, C)
            self.dec1 = _SwinStage(sd, "dec1", C, cfg.blocks[0], cfg, dtype, dev)
            self.out = _pc(sd, "out", dtype, dev, C)

    def forward(self, img: Act) -> Act:
        """img [N,Hp,Wp,3(8)] in [0,1], Hp,Wp multiples of cfg.align -> restored image (unclamped)."""
        e1 = self.enc1(self.embed(img))
        e2 = self.enc2(self.down1(e1))
        m = self.mid(self.down2(e2))
        d2 = self.dec2(self.red2(self.up2(m), e2))
        d1 = self.dec1(self.red1(self.up1(d2), e1))
        return self.out(d1, residual=img)

    def restore(self, frames_d: torch.Tensor, swap_rb: bool = True, want_f32: bool = False):
        """[n,H,W,3] u8 on the device -> restored [n,H,W,3] u8 (reflect-padded to cfg.align inside)."""
        n, h, w, _ = frames_d.shape
        a = self.cfg.align
        hp, wp = (h + a - 1) // a * a, (w + a - 1) // a * a
        with torch.cuda.device(self.device):
            x = ops.u8_to_float(frames_d, self.dtype, 1.0, 0.0, swap_rb=swap_rb, div255=True)
            if hp != h or wp != w:
                xp = ops.new_act(n, hp, wp, 3, self.dtype, self.device, zero=True)
                ops.pad_reflect_axpy(x, hp, wp, xp, 0, mul=1.0)
                x = xp
            y = self.forward(x)
            if hp != h or wp != w:
                y = ops.crop_copy(y, h, w)
            return ops.float_to_u8(y, 1.0, 0.0, mode=0, swap_rb=swap_rb, want_f32=want_f32)
