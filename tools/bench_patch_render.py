#!/usr/bin/env python3
"""The reference's only published workloads, timed like for like: ``Model.patch_render`` on ONE 480 x 640
frame with 256 + 256 samples per ray, ``jacobian_mlp`` with A = 6 (notebooks/real_world/1_visualize_jacobian_fields.ipynb:477,
19.6 k rays/s) and ``jacobian_transformer`` with A = 8 (2_inverse_dynamics.ipynb:331, 16.3 k rays/s) -- numbers printed by tqdm on
the authors' machine (a single consumer GPU; context, NOT a same-node comparison).  Encoder INCLUDED (ResNet-34 on MIOpen, once
per frame), every output of RenderingOutput produced (rgb, depth + colour map, flow + colour map, positions, action features,
steps, weights); a new image every call.  Exact fp32 products ("f32") and the package default precision; the plain-fp16 mode is
listed for the 512 x 512 config-5 frame by bench.py.  Seeded random weights (no checkpoint travels), synthetic image.

    python tools/bench_patch_render.py [--frames 5] > profiles/r05_patch_render.json          (on the GPU box)"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.build()
    from neural_jacobian_field_amd import geometry, hip, synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput

    dev = torch.device("cuda:0")
    H, W, S = 480, 640, 256
    published = {"jacobian_mlp": {"action_dim": 6, "rays_per_s": 19.6e3, "source": "notebooks/real_world/1_visualize_jacobian_fields.ipynb:477"},
                 "jacobian_transformer": {"action_dim": 8, "rays_per_s": 16.3e3, "source": "notebooks/real_world/2_inverse_dynamics.ipynb:331"}}
    out = {"workload": f"Model.patch_render, one {H}x{W} frame, {S}+{S} samples/ray, encoder included, all RenderingOutput fields",
           "frames_timed": args.frames, "published_context": "BASELINE.md section 1: tqdm rates on the authors' machine (presumed one "
           "consumer GPU); another node, real checkpoints -- context only", "rows": {}}
    cams = synthetic.synthetic_cameras(1)
    d = lambda t: t.to(dev)
    for kind, pub in published.items():
        A = pub["action_dim"]
        model = Model(model_cfg_from_dict({"action_dim": A, "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                                           "action_decoder": {"name": kind}}))
        model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes(kind, A), seed=0))
        model.to(dev).eval().requires_grad_(False)
        origins, directions, _ = geometry.full_frame_rays(H, W, d(cams["trgt_k_norm"]), d(cams["trgt_c2w"]))
        k_pix = geometry.denormalize_intrinsics(d(cams["trgt_k_norm"]), W, H)
        rin = RenderingInput(origins, directions, d(cams["z_near"]), d(cams["z_far"]))
        rob = RobotInput(d(synthetic.synthetic_action(1, A, seed=2)))
        g = torch.Generator().manual_seed(3)
        images = [d(torch.rand(1, 3, H, W, generator=g)) for _ in range(args.frames + args.warmup)]
        for prec in ("f32", hip.DEFAULT_PRECISION):
            model.set_precision(prec)
            cam_of = lambda img: CameraInput(img, d(cams["ctxt_c2w"]), d(cams["ctxt_k_norm"]), d(cams["trgt_c2w"]), k_pix)
            with torch.no_grad():
                for i in range(args.warmup):
                    model.patch_render(cam_of(images[i]), rin, rob, render_height=H, render_width=W)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(args.frames):
                    ro = model.patch_render(cam_of(images[args.warmup + i]), rin, rob, render_height=H, render_width=W)
                e1.record()
                torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.frames
            assert torch.isfinite(ro.rgb).all() and torch.isfinite(ro.flow_raw).all() and ro.rgb.shape == (1, H, W, 3)
            out["rows"][f"{kind}[A={A}, {prec}]"] = {
                "ms_per_frame": round(ms, 2), "rays_per_s": round(H * W / ms * 1e3, 1), "published_rays_per_s": pub["rays_per_s"],
                "published_source": pub["source"], "ratio_to_published_context_only": round(H * W / ms * 1e3 / pub["rays_per_s"], 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
