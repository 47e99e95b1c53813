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
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd import hip, synthetic  # noqa: E402
from neural_jacobian_field_amd.renderer import FusedRenderer, RenderRequest  # noqa: E402

dev = torch.device("cuda:0")
case = synthetic.synthetic_case(2, 32, 48, 700, 8, seed=7, device=dev, identity_context=False)   # package-only inputs: no oracle/
c = case["cams"]
req = RenderRequest(vis=True, sample_weights=True, per_sample=True)
for prec in (sys.argv[1:] or ["f16x2", "f32"]):   # (round 5: pass e.g. `f16` for the plain-fp16 kernels)
    fr = FusedRenderer(dev, 1, 8, precision=prec)
    fr.load_weights(case["params"])
    res = fr.render(case["feats"], case["origins"], case["directions"], c["ctxt_c2w"], c["ctxt_k_norm"], c["z_near"], c["z_far"],
                    [64], 64, trgt_c2w=c["trgt_c2w"], trgt_k_pix=case["k_pix"], action=case["action"], request=req)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in [res.rgb, res.depth, res.optical_flow, res.bins_list[-1]] + [res.extras[k] for k in sorted(res.extras)]:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    print(prec, h.hexdigest()[:16])
