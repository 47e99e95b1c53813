#!/usr/bin/env python3
"""SHA-256 and time of njf_project_pyramid's hoisted map for the library selected by NJF_HIP_LIB, on several pyramids
(4 x 4-blocked up-sampled add vs the per-texel form built with -DNJF_UPSAMPLE_PER_TEXEL: must agree bit for bit)."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = {
    "256x256 b1 n768": (1, 768, ((64, 128, 128), (64, 64, 64), (128, 32, 32), (256, 16, 16))),
    "256x256 b7 n1152": (7, 1152, ((64, 128, 128), (64, 64, 64), (128, 32, 32), (256, 16, 16))),
    "480x640 b2 n384": (2, 384, ((64, 240, 320), (64, 120, 160), (128, 60, 80), (256, 30, 40))),
    "32x64 b3 n384 (coarsest level 1 texel high)": (3, 384, ((64, 8, 16), (64, 4, 8), (128, 2, 4), (256, 1, 2))),
    "250x250 b1 n384 (not 2^-s: per-texel form)": (1, 384, ((64, 125, 125), (64, 63, 63), (128, 32, 32), (256, 16, 16))),
}
for name, (batch, n, pyr) in SHAPES.items():
    g = torch.Generator().manual_seed(len(name))
    levels = [torch.randn(batch, c, h, w, generator=g).to(dev).contiguous() for c, h, w in pyr]
    wz, bz = (torch.randn(512, n, generator=g) * 0.05).to(dev).contiguous(), torch.randn(n, generator=g).to(dev)
    out = torch.empty(batch, pyr[0][1], pyr[0][2], n, device=dev)
    for prec in ("f32", "f16x2"):
        hip.project_pyramid(levels, wz, bz, out, precision=prec)
        torch.cuda.synchronize()
        digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.project_pyramid(levels, wz, bz, out, precision=prec)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:48s} {prec:6s} {digest}  {e0.elapsed_time(e1) / 10:.4f} ms  finite={bool(torch.isfinite(out).all())}")
