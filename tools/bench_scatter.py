#!/usr/bin/env python3
"""njf_scatter_footprint against the four index_add_ calls it replaces, at the training batch shape
(7 scenes x 256 rays x 64 samples = 114,688 points, 128 channels, 128x128 texels per scene).  GPU box only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
scenes, rays, samples, hf = 7, 256, 64, 128
g = torch.Generator().manual_seed(0)
x0 = torch.rand(scenes, rays, 1, generator=g) * (hf - 40)
y0 = torch.rand(scenes, rays, 1, generator=g) * (hf - 40)
s = torch.arange(samples)[None, None, :] * 0.3                      # ~3 consecutive samples share a texel
x, y = (x0 + s).floor().long(), (y0 + 0.5 * s).floor().long()
base = torch.arange(scenes)[:, None, None] * hf * hf + y * hf + x
idx = torch.stack([base, base + 1, base + hf, base + hf + 1], dim=-1).reshape(-1, 4).to(torch.int32).to(dev)
w = torch.rand(idx.shape[0], 4, generator=g).to(dev)
grad = torch.randn(idx.shape[0], 128, generator=g).to(dev)
texels = scenes * hf * hf


def run_hip():
    out = torch.zeros(texels, 128, device=dev)
    hip.scatter_footprint(grad, idx, w, out, run_length=samples)
    return out


def run_torch():
    out = torch.zeros(texels, 128, device=dev)
    il = idx.long()
    for c in range(4):
        out.index_add_(0, il[:, c], grad * w[:, c:c + 1])
    return out


a, b = run_hip(), run_torch()
print("max rel diff", ((a - b).abs().max() / b.abs().max()).item())
for name, fn in (("njf_scatter_footprint", run_hip), ("4 x index_add_", run_torch)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. zeroing the output)")
