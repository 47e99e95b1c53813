#!/usr/bin/env python3
"""Diagnosis (GPU box): how accurate is the hoisted map G = lin_z(features) by route (pyramid producer vs projection of the
concatenated map) and MFMA precision, against float64 -- and how far do the level-0 proposal weights move with the route?"""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import __graft_entry__ as g_; g_.build()
import njf_oracle as orc, parity_harness as ph
from neural_jacobian_field_amd import hip, synthetic
from neural_jacobian_field_amd.config import model_cfg_from_dict
from neural_jacobian_field_amd.encoder import FeaturePyramid
from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().cpu().abs().max() + 1e-30)).item()
dev = torch.device("cuda:0")
B, H, W, R, S = 2, 16, 16, 40, 32
case = ph.make_case(B, H, W, R, 8, seed=4)
full = synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", 8), seed=4); full.update(case["params"])
model = Model(model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S}, "action_decoder": {"name": "jacobian_mlp"}}))
model.load_state_dict(full, strict=True); model.to(dev).eval().requires_grad_(False)
image = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(9)).to(dev)
c = case["cams"]; d = lambda t: t.to(dev)
cam = CameraInput(image, d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(c["trgt_c2w"]), d(case["k_pix"]))
rin = RenderingInput(d(case["origins"]), d(case["directions"]), d(c["z_near"]), d(c["z_far"])); rob = RobotInput(d(case["action"]))
with torch.no_grad():
    lat = model.encoder._latents(image)
    feats = model.encoder.forward(image)
    pn = model.proposal_networks[0]
    for prec in ("f32", "f16x2"):
        pn.precision = prec; pn._packed_version = None; pn.packed()
        wz, bz = pn._wz, pn._bz
        ref = torch.einsum("bkhw,kn->bhwn", feats.double(), wz.double()) + bz.double()
        g_dir = torch.empty(B, feats.shape[2], feats.shape[3], wz.shape[1], device=dev); hip.project_features(feats.contiguous(), wz, bz, g_dir, precision=prec)
        g_pyr = torch.empty_like(g_dir); hip.project_pyramid(lat, wz, bz, g_pyr, precision=prec)
        g_t32 = torch.einsum("bkhw,kn->bhwn", feats, wz) + bz
        print(f"{prec}: G direct vs f64 {rel(g_dir, ref):.2e}  pyramid vs f64 {rel(g_pyr, ref):.2e}  torch fp32 einsum vs f64 {rel(g_t32, ref):.2e}  pyramid vs direct {rel(g_pyr, g_dir):.2e}")
        # the interpolated features themselves: ATen vs float64
    up64 = torch.cat([F.interpolate(l.double(), lat[0].shape[-2:], mode="bilinear", align_corners=False) for l in lat], 1)
    print(f"features (ATen upsample+cat fp32) vs float64 interpolation of the same latents: {rel(feats, up64):.2e}")
    # level-0 weights by route
    model.set_precision("f32")
    res = {}
    for name, f in (("pyramid", FeaturePyramid(lat)), ("tensor", feats)):
        outs, bins, wl, bl, _ = model._fused_render(cam, rin, rob, f, want_lists=True, want_vis=False, want_samples=False)
        res[name] = wl[0].clone()
    params = {k: v for k, v in full.items()}
    ref = orc.model_forward(params, features=feats.cpu(), ctxt_c2w=c["ctxt_c2w"], ctxt_k_norm=c["ctxt_k_norm"], trgt_c2w=c["trgt_c2w"], trgt_k_pix=case["k_pix"],
                            origins=case["origins"], directions=case["directions"], z_near=c["z_near"], z_far=c["z_far"], action=case["action"],
                            num_proposal_samples=[S], num_nerf_samples=S, decoder_kind="jacobian_mlp")
    r64 = orc.model_forward({k: (v.double() if v.is_floating_point() else v) for k, v in full.items()}, features=feats.cpu().double(), ctxt_c2w=c["ctxt_c2w"].double(), ctxt_k_norm=c["ctxt_k_norm"].double(), trgt_c2w=c["trgt_c2w"].double(), trgt_k_pix=case["k_pix"].double(),
                            origins=case["origins"].double(), directions=case["directions"].double(), z_near=c["z_near"].double(), z_far=c["z_far"].double(), action=case["action"].double(),
                            num_proposal_samples=[S], num_nerf_samples=S, decoder_kind="jacobian_mlp")
    print(f"level-0 weights (f32 MFMA, same HIP features): pyramid route vs oracle {rel(res['pyramid'], ref.weights_list[0]):.2e}  tensor route vs oracle {rel(res['tensor'], ref.weights_list[0]):.2e}"
          f"  oracle fp32 vs fp64 {rel(ref.weights_list[0], r64.weights_list[0]):.2e}  pyramid vs f64 {rel(res['pyramid'], r64.weights_list[0]):.2e}  tensor vs f64 {rel(res['tensor'], r64.weights_list[0]):.2e}")
