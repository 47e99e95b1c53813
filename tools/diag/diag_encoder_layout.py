#!/usr/bin/env python3
"""Encoder (ResNet-34 trunk, MIOpen fp32) forward and forward+backward time at the training batch shape, NCHW against
channels_last, and the difference of the latents between the two (a layout experiment; the product path is NCHW)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd.config import EncoderResnetCfg  # noqa: E402
from neural_jacobian_field_amd.encoder import EncoderResnet  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = EncoderResnet(EncoderResnetCfg()).to(dev)
rgb = torch.rand(7, 3, 256, 256, device=dev)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fwd(x):
    with torch.no_grad():
        return enc._latents(x)


def fwd_bwd(x):
    lat = enc._latents(x)
    sum(l.square().mean() for l in lat).backward()
    enc.zero_grad(set_to_none=True)


for mode in ("eval", "train"):
    enc.train(mode == "train")
    for layout in ("nchw", "channels_last"):
        enc.to(memory_format=torch.channels_last if layout == "channels_last" else torch.contiguous_format)
        x = rgb.contiguous(memory_format=torch.channels_last) if layout == "channels_last" else rgb.contiguous()
        t_f = timed(lambda: fwd(x))
        t_fb = timed(lambda: fwd_bwd(x))
        t_c = timed(lambda: [l.contiguous() for l in fwd(x)]) - t_f
        print(f"{mode:5s} {layout:13s} fwd {t_f:.3f} ms   fwd+bwd {t_fb:.3f} ms   (+ NCHW copies of the latents {t_c:.3f} ms)")
enc.eval()
enc.to(memory_format=torch.contiguous_format)
a = fwd(rgb)
enc.to(memory_format=torch.channels_last)
b = fwd(rgb.contiguous(memory_format=torch.channels_last))
for l, (p, q) in enumerate(zip(a, b)):
    print("level", l, "max rel diff", ((p - q).abs().max() / p.abs().max()).item())
