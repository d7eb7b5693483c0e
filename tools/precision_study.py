#!/usr/bin/env python3
"""CPU study (test tooling, not product code): how much of the 1e-3 max-abs bar each f16 rounding of
the SinSR path costs, and which sections must run in fp32 for a mixed mode to meet it.

The device's f16 mode is emulated inside the fp32 oracle (oracle/sinsr_ref.py): a conv / linear whose
layer name is selected rounds its activations and weights to f16 (the MFMA operands), accumulates in
fp32 and - if `storage` is on - rounds what it writes to HBM to f16 (residual added in fp32 first, as
the epilogue does).  Unselected layers run in exact fp32, like the device's fp32 MFMA mode.

    python tools/precision_study.py            # table 1: operand vs storage rounding; table 2: sections
"""
import dataclasses
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elvis_amd.synth import synth_clip  # noqa: E402
from elvis_amd.weights import SinSRConfig, frame_noise, make_sinsr_weights  # noqa: E402
from oracle import sinsr_ref as R  # noqa: E402

MODE = {"sel": lambda name: True, "operands": True, "storage": True}


def q(x):
    return x.half().float()


def _full(p):   # UNet layer names reach the oracle's helpers without their "model." prefix
    return p if p.startswith("ae.") else "model." + p


def _mm(fn, x, w, b, on, res=None, act=None, **kw):
    if on and MODE["operands"]:
        x, w = q(x), q(w)
    y = fn(x, w, b, **kw)
    if act is not None:
        y = act(y)
    if res is not None:
        y = res + y
    return q(y) if on and MODE["storage"] else y


def conv_q(sd, p, x, stride=1, padding=None, res=None):
    w = sd[p + ".weight"]
    if padding is None:
        padding = w.shape[-1] // 2
    return _mm(F.conv2d, x, w, sd[p + ".bias"], MODE["sel"](_full(p)), res=res, stride=stride, padding=padding)


def ae_resblock_q(sd, p, x, g):
    h = conv_q(sd, p + ".conv1", F.silu(R.gn(sd, p + ".norm1", x, g, 1e-6)))
    skip = conv_q(sd, p + ".nin_shortcut", x) if (p + ".nin_shortcut.weight") in sd else x
    return conv_q(sd, p + ".conv2", F.silu(R.gn(sd, p + ".norm2", h, g, 1e-6)), res=skip)


def unet_resblock_q(sd, p, x, emb, cfg):
    g = cfg.gn_groups
    h = conv_q(sd, p + ".in_layers.2", F.silu(R.gn(sd, p + ".in_layers.0", x, g, 1e-5)))
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    scale, shift = e[:, :, None, None].chunk(2, dim=1)
    h = F.silu(R.gn(sd, p + ".out_layers.0", h, g, 1e-5) * (1 + scale) + shift)
    skip = conv_q(sd, p + ".skip_connection", x) if (p + ".skip_connection.weight") in sd else x
    return conv_q(sd, p + ".out_layers.3", h, res=skip)


def swin_block_q(sd, p, x, h, w, cfg, shift):
    E, ws, heads = cfg.swin_embed_dim, cfg.window_size, cfg.heads
    b = x.shape[0]
    on = MODE["sel"](_full(p))
    st = (lambda t: q(t)) if on and MODE["storage"] else (lambda t: t)

    def lin(name, t, res=None, act=None):
        return _mm(F.linear, t, sd[p + name + ".weight"], sd[p + name + ".bias"], on, res=res, act=act)

    y = st(F.layer_norm(x, (E,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)).view(b, h, w, E)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    win = R.window_partition(y, ws)
    qkv = lin(".attn.qkv", win)
    nw, n = win.shape[0], ws * ws
    qkv = qkv.view(nw, n, 3, heads, E // heads).permute(2, 0, 3, 1, 4)
    attn = (qkv[0] @ qkv[1].transpose(-2, -1)) * ((E // heads) ** -0.5)
    rpi = R.relative_position_index(ws)
    attn = attn + sd[p + ".attn.relative_position_bias_table"][rpi.view(-1)].view(n, n, heads).permute(2, 0, 1)[None]
    if shift:
        m = R.shift_mask(h, w, ws, shift)
        attn = (attn.view(b, m.shape[0], heads, n, n) + m[None, :, None]).view(-1, heads, n, n)
    attn = attn.softmax(-1)
    if on and MODE["operands"]:
        attn = q(attn)
    o = st((attn @ qkv[2]).transpose(1, 2).reshape(nw, n, E))
    # the projection is per token, so it commutes with window_reverse / roll
    o = lin(".attn.proj", o, res=R.window_partition(torch.roll(x.view(b, h, w, E), (-shift, -shift), (1, 2)) if shift else x.view(b, h, w, E), ws))
    y = R.window_reverse(o, ws, h, w)
    if shift:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = y.view(b, h * w, E)
    y = st(F.layer_norm(x, (E,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5))
    y = lin(".mlp.fc1", y, act=F.gelu)
    return lin(".mlp.fc2", y, res=x)


def emulate(cfg, sd, lr, noise, sel, operands=True, storage=True):
    MODE.update(sel=sel, operands=operands, storage=storage)
    saved = (R.conv, R.ae_resblock, R.unet_resblock, R.swin_block)
    R.conv, R.ae_resblock, R.unet_resblock, R.swin_block = conv_q, ae_resblock_q, unet_resblock_q, swin_block_q
    try:
        return R.sinsr_forward(sd, cfg, lr, noise)
    finally:
        R.conv, R.ae_resblock, R.unet_resblock, R.swin_block = saved


def pre(*prefixes):
    return lambda name: name.startswith(prefixes)


SECTIONS = {   # name -> (layer-name prefixes, what runs there at 1080p)
    "enc0": (("ae.encoder.conv_in", "ae.encoder.down.0"), "encoder 128 ch @ 1080x1920"),
    "enc1": (("ae.encoder.down.1",), "encoder 256 ch @ 540x960"),
    "enc2": (("ae.encoder.down.2", "ae.encoder.mid", "ae.encoder.conv_out", "ae.quant_conv"), "encoder 512 ch @ 270x480 + z"),
    "unet": (("model.",), "Swin-UNet @ 320x512 latent"),
    "dec2": (("ae.post_quant_conv", "ae.decoder.conv_in", "ae.decoder.mid", "ae.decoder.up.2"), "decoder 512 ch @ 270x480"),
    "dec1": (("ae.decoder.up.1",), "decoder 256 ch @ 540x960"),
    "dec0": (("ae.decoder.up.0", "ae.decoder.conv_out"), "decoder 128 ch @ 1080x1920"),
}


def main():
    torch.set_num_threads(int(os.environ.get("ELVIS_CPU_THREADS", "8")))
    cfg = dataclasses.replace(SinSRConfig(), quantize=False)
    sd = make_sinsr_weights(cfg, 0)
    S = 64
    lr = torch.from_numpy(synth_clip(20260501, 1, S, S)[0])
    noise = frame_noise(cfg, 42, 0, *R.padded_latent_shape(cfg, S, S))
    ref = R.sinsr_forward(sd, cfg, lr, noise)

    def report(label, out):
        d = (out - ref).abs()
        print(f"{label:58s} max {d.max().item():.3e}  rms {d.pow(2).mean().sqrt().item():.3e}", flush=True)

    print("# table 1: every layer f16 - which rounding costs what (256x256 output tile, full-width config, continuous path)")
    report("f16 operands + f16 storage (the device's f16 mode)", emulate(cfg, sd, lr, noise, lambda n: True))
    report("f16 operands, fp32 storage", emulate(cfg, sd, lr, noise, lambda n: True, storage=False))
    report("fp32 operands, f16 storage", emulate(cfg, sd, lr, noise, lambda n: True, operands=False))
    print("# table 2: ONE section in f16 (operands + storage), everything else exact fp32")
    for k, (pf, what) in SECTIONS.items():
        report(f"{k:5s} {what}", emulate(cfg, sd, lr, noise, pre(*pf)))
    print("# table 3: mixed modes - the listed sections in f16, the rest in fp32")
    for combo in (("enc0", "enc1", "dec1", "dec0"), ("enc0", "enc1", "enc2", "dec1", "dec0"), ("enc0", "dec0"),
                  ("enc0", "enc1", "dec0"), ("enc0", "dec1", "dec0"), ("dec1", "dec0"), ("enc0", "enc1", "enc2", "dec0")):
        pf = sum((SECTIONS[c][0] for c in combo), ())
        report("f16: " + "+".join(combo), emulate(cfg, sd, lr, noise, pre(*pf)))


if __name__ == "__main__":
    main()
