#!/usr/bin/env python3
"""Frame time of the fused path through the drop-in Model API for both Jacobian heads (not the headline metric):
C2 shape (B=1, 256x256 rays, 64+64 samples, A=8), encoder excluded (features given), eval mode, default precision."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd import synthetic
from neural_jacobian_field_amd.config import model_cfg_from_dict
from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput

dev = torch.device("cuda:0")
B, H, W, S, A = 1, 256, 256, 64, 8
case = synthetic.synthetic_case(B, H, W, None, A, seed=0, device=dev)   # package-only inputs: the tool travels without oracle/
c = case["cams"]; d = lambda t: t.to(dev)
res = {}
for kind in ("jacobian_mlp", "jacobian_transformer"):
    model = Model(model_cfg_from_dict({"action_dim": A, "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                                       "action_decoder": {"name": kind}}))
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes(kind, A), seed=0))
    model.to(dev).eval().requires_grad_(False)
    cam = CameraInput(d(torch.rand(B, 3, H, W)), d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(c["trgt_c2w"]), d(case["k_pix"]))
    rin = RenderingInput(d(case["origins"]), d(case["directions"]), d(c["z_near"]), d(c["z_far"]))
    rob = RobotInput(d(case["action"]))
    feats = d(case["feats"])
    with torch.no_grad():
        run = lambda: model._fused_render(cam, rin, rob, feats, want_lists=False, want_vis=False, want_samples=False)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            model.decoder._hoist.key = None            # re-project every frame, as for a new image
            model.proposal_networks[0]._hoist.key = None
            run()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    res[kind] = {"ms_per_frame": round(ms, 3), "rays_per_s": round(B * H * W / ms * 1e3, 1)}
print(json.dumps(res))
