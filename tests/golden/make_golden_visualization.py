#!/usr/bin/env python3
"""Golden vectors for the Jacobian-field colour mapping (reference: inference/jacobian_color_map.py:53-154), generated
by importing the reference in the build container -- same rules as make_golden.py (the reference never travels).

Usage:  python tests/golden/make_golden_visualization.py     # rewrites tests/golden/visualization.npz
cv2 / matplotlib are absent here and only used by the reference's plotting helpers, so they are stubbed.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.install_shims()
    mg._module("cv2")
    mg._module("matplotlib")
    mg._module("matplotlib.pyplot")
    from neural_jacobian_field.inference import jacobian_color_map as ref

    h, w, a = 6, 9, 8
    jac = mg.randn(41, 2, h, w, 3 * a) * 0.3
    ext = mg.rigid(42, 2)[:, None, None, None]                   # broadcast over (h, w, action)
    cmap = torch.tensor(ref.JACOBIAN_COLORMAP["model_allegro"]).t().contiguous()   # [rgb, action]
    s0 = ref.compute_joint_sensitivity(jac, None, mode=0)
    s1 = ref.compute_joint_sensitivity(jac, ext, mode=1)
    v0 = ref.visualize_joint_sensitivity(s0, cmap)
    pts = mg.randn(43, 50, a, 3)
    p0 = ref.compute_joint_sensitivity_point_cloud(pts)
    pc0 = ref.visualize_joint_sensitivity_point_cloud(p0, cmap, mode=0)
    pc1 = ref.visualize_joint_sensitivity_point_cloud(p0, cmap, mode=1)
    mg.save("visualization", jacobians=jac, extrinsics=ext, color_map=cmap, sensitivity_mode0=s0, sensitivity_mode1_ext=s1,
            image_mode0=v0, points=pts, point_sensitivity=p0, point_colors_mode0=pc0, point_colors_mode1=pc1,
            colormap_model_toy_arm=ref.JACOBIAN_COLORMAP["model_toy_arm"],
            colormap_model_pneumatic_hand_only=ref.JACOBIAN_COLORMAP["model_pneumatic_hand_only"],
            colormap_model_allegro_transformer=ref.JACOBIAN_COLORMAP["model_allegro_transformer"])


if __name__ == "__main__":
    main()
