#!/usr/bin/env python3
"""SHA-256 + time of the exact-fp32 per-image projection (njf_project_features_ld, precision f32) per library
(``NJF_HIP_LIB``): the 64-texel-per-wave kernel of round 4 against the 32-texel one.  Run once per library; the digests must
agree (every accumulator sees the same MFMA sequence).  GPU box."""
import hashlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd import hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
print("library:", hip.LIB_PATH)
for b, hw_shape, n in ((1, (128, 128), 1152), (7, (128, 128), 1152), (1, (256, 256), 1152), (2, (125, 125), 384), (3, (16, 16), 768), (1, (9, 7), 200)):
    feats = torch.randn(b, 512, *hw_shape, generator=g).to(dev)
    wz, bz = (torch.randn(512, n, generator=g) * 0.05).to(dev), torch.randn(n, generator=g).to(dev)
    out = torch.empty(b, *hw_shape, n, device=dev)
    for _ in range(3):
        hip.project_features(feats, wz, bz, out, precision="f32")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.project_features(feats, wz, bz, out, precision="f32")
    e1.record(); torch.cuda.synchronize()
    print(f"b{b} {hw_shape} n{n}: sha {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}  {e0.elapsed_time(e1) / 20:.4f} ms  finite {bool(torch.isfinite(out).all())}")
