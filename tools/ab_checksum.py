#!/usr/bin/env python3
"""SHA-256 of the fused path's outputs on a fixed seeded frame, for the library selected by NJF_HIP_LIB.

Used by tools/ab_variants.sh: scheduling-only kernel variants (DMA encoding, compiler flags, instruction selection of
the hi/lo split) must reproduce the shipped library's outputs BIT FOR BIT, in both MFMA precisions, before their
timing is even looked at.  Not part of the product path."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import parity_harness as ph  # noqa: E402
from neural_jacobian_field_amd.renderer import RenderRequest  # noqa: E402

dev = torch.device("cuda:0")
case = ph.make_case(2, 32, 48, 700, 8, seed=7, identity_context=False)
req = RenderRequest(vis=True, sample_weights=True, per_sample=True)
for prec in ("f16x2", "f32"):
    res, _, _ = ph.hip_forward(case, 64, 64, dev, request=req, precision=prec)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in [res.rgb, res.depth, res.optical_flow, res.bins_list[-1]] + [res.extras[k] for k in sorted(res.extras)]:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    print(prec, h.hexdigest()[:16])
