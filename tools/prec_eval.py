#!/usr/bin/env python3
"""Precision-mode evaluation on the GPU box: parity errors of every MFMA precision against the CPU oracle on the
parity-suite cases (with the fp32-vs-fp64 noise floor of the oracle beside them) and C2 kernel times, in ONE process.

    python tools/prec_eval.py [--quick] > gpurun_out/prec_eval.log

Experiment tooling; not part of the product path.  It VERIFIES against the CPU oracle (oracle/parity_harness.py), so it needs
oracle/ next to the package -- inputs-only tools build their frames with synthetic.synthetic_case instead."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import __graft_entry__ as entry  # noqa: E402

entry.build()
import parity_harness as ph  # noqa: E402
from neural_jacobian_field_amd import geometry, hip, synthetic  # noqa: E402
from neural_jacobian_field_amd.renderer import FusedRenderer  # noqa: E402

dev = torch.device("cuda:0")
PRECS = [p for p in ("f32", "f16x2", "f16f6") if p in hip.PRECISIONS]
quick = "--quick" in sys.argv

cases = [
    dict(batch=1, height=16, width=16, rays=96, s_prop=32, s_final=32),
    dict(batch=2, height=16, width=24, rays=50, s_prop=64, s_final=64),
    dict(batch=1, height=32, width=32, rays=None, s_prop=64, s_final=64, action_dim=6),
    dict(batch=2, height=16, width=16, rays=40, s_prop=32, s_final=32, identity_context=False),
    dict(batch=4, height=16, width=16, rays=48, s_prop=128, s_final=128),
]
if quick:
    cases = cases[:2]
for cfg in cases:
    print("case", cfg)
    for prec in PRECS:
        rep = ph.run_parity_case(device=dev, tol=1e-4, precision=prec, **cfg)
        keys = ("rgb", "depth", "optical_flow", "prop_weights", "final_bins", "s_rgb", "s_depth", "s_optical_flow", "s_density",
                "s_color", "s_jacobian", "s_action_features")
        print(f"  {prec:6s} ok={rep['ok']} worst={rep['worst']:.2e} | " + " ".join(f"{k}={rep['errors'][k]:.1e}" for k in keys))
    print("  floor        | " + " ".join(f"{k}={rep['fp32_noise_floor'].get(k, 0):.1e}" for k in keys))

# ---- C2 kernel times
H = W = 256
S, A = 64, 8
params = synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", A, with_encoder=False), seed=0)
cams = synthetic.synthetic_cameras(1)
d = lambda t: t.to(dev)
feats = d(synthetic.synthetic_features(1, H, W, seed=1))
action = d(synthetic.synthetic_action(1, A, seed=2))
ctxt_c2w, ctxt_k, trgt_c2w = d(cams["ctxt_c2w"]), d(cams["ctxt_k_norm"]), d(cams["trgt_c2w"])
z_near, z_far = d(cams["z_near"]), d(cams["z_far"])
origins, directions, _ = geometry.full_frame_rays(H, W, d(cams["trgt_k_norm"]), trgt_c2w)
k_pix = geometry.denormalize_intrinsics(d(cams["trgt_k_norm"]), W, H)
ctxt_w2c, trgt_w2c = torch.linalg.inv(ctxt_c2w), torch.linalg.inv(trgt_c2w)
dp = {k: d(v) for k, v in params.items()}
results, ref = {}, None


def kernel_ms(records, steps):
    """Mean duration per step of the fused launches, from the per-launch HIP events hip.set_profile_sink collects."""
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in records:
        key = {"njf_proposal_forward": "proposal_ms", "njf_render_forward": "render_ms"}.get(name)
        if key:
            out[key] = out.get(key, 0.0) + e0.elapsed_time(e1)
    return {k: round(v / steps, 3) for k, v in out.items()}


for prec in PRECS:
    fr = FusedRenderer(dev, 1, A, precision=prec)   # sets the whole model (proposal pass included) to `prec`'s policy
    fr.load_weights(dp)
    gmap = feats.contiguous()
    steps = 3 if prec == "f32" else 8
    render = lambda: fr.render(gmap, origins, directions, ctxt_c2w, ctxt_k, z_near, z_far, [S], S, trgt_c2w=trgt_c2w,
                               trgt_k_pix=k_pix, action=action, ctxt_w2c=ctxt_w2c, trgt_w2c=trgt_w2c)
    for _ in range(2):
        res = render()
    launches = []
    hip.set_profile_sink(launches)
    for _ in range(steps):
        res = render()
    hip.set_profile_sink(None)
    if ref is None:
        ref = res
    results[prec] = dict(**kernel_ms(launches, steps),
                         rgb_vs_f32=ph.rel_err(res.rgb, ref.rgb), depth_vs_f32=ph.rel_err(res.depth, ref.depth),
                         flow_vs_f32=ph.rel_err(res.optical_flow, ref.optical_flow))
    print("C2", prec, json.dumps(results[prec]))
