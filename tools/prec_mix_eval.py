#!/usr/bin/env python3
"""Which network's precision matters for which output?  The two-level golden fixture (tests/golden/model_mlp2.npz) and
the standard one (model_mlp.npz) through Model.forward under every decoder precision mix; errors against the reference's
fp32 outputs next to the reference's own fp32-vs-fp64 floor.  Experiment tooling (GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import __graft_entry__ as entry  # noqa: E402

entry.build()
from neural_jacobian_field_amd import synthetic  # noqa: E402
from neural_jacobian_field_amd.config import model_cfg_from_dict  # noqa: E402
from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput  # noqa: E402

dev = torch.device("cuda:0")


def load(name):
    with np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")) as f:
        return {k: torch.from_numpy(f[k]).to(dev) for k in f.files}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


for fixture, props, nerf, f64 in (("model_mlp2", [16, 12], 10, None), ("model_mlp", [16], 12, "model_mlp_f64")):
    g = load(fixture)
    if f64:
        g.update({k + "_f64": v for k, v in load(f64).items()})
    cfg = model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": props, "num_nerf_samples": nerf},
                               "action_decoder": {"name": "jacobian_mlp"}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", 8, num_proposal_networks=len(props)), seed=0))
    model.to(dev).eval().requires_grad_(False)
    cam = CameraInput(g["image"], g["ctxt_c2w"], g["ctxt_k_norm"], g["trgt_c2w"], g["trgt_k_pix"])
    rin, rob = RenderingInput(g["origins"], g["directions"], g["z_near"], g["z_far"]), RobotInput(g["action"])
    print(fixture, "floors:", {k: f"{rel(g[k], g[k + '_f64']):.1e}" for k in ("rgb", "depth", "optical_flow")},
          "max|flow| =", f"{g['optical_flow'].abs().max().item():.3g}")
    for d, p, j in (("f32", "f32", "f32"), ("f16x2", "f16x2", "f16x2"), ("f16f6", "f16x2", "f16f6"), ("f16f6", "f16x2", "f16x2"),
                    ("f16x2", "f16x2", "f16f6"), ("f16f6", "f16f6", "f16f6")):
        model.set_precision(d, p, j)
        out = model.forward(cam, rin, rob).standard_output
        print(f"  density={d:6s} proposal={p:6s} jacobian={j:6s} | rgb {rel(out.rgb, g['rgb']):.1e} depth {rel(out.depth, g['depth']):.1e} "
              f"flow {rel(out.optical_flow, g['optical_flow']):.1e}")
