"""Jacobian-field colour mapping on the device -- mirror of ``inference/jacobian_color_map.py:13-154``.

Same function names, arguments and return types as the reference (images come back as numpy arrays, point colours as
tensors); the arithmetic runs on whatever device the Jacobians live on, so a ``patch_render`` frame is coloured
without a host round trip of the [H,W,3A] field.  The plotting helpers of that file (matplotlib / cv2 overlays,
``visualize_jacobian_chain_structure``) are not part of the rendering path and are not mirrored.
"""

from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

# per-command display colours, one RGB triple per action dimension (jacobian_color_map.py:13-50)
_ALLEGRO = [[0.0, 0.5, 0.5], [0, 1, 0], [0.8, 0.1, 0.1], [0.8, 0.0, 0.8], [0.0, 0.8, 0], [1.0, 0.8, 0], [1, 1, 0], [1, 0.0, 0.0]]
JACOBIAN_COLORMAP: Dict[str, List[List[float]]] = {
    "model_allegro": _ALLEGRO,
    "model_allegro_transformer": [list(c) for c in _ALLEGRO],
    "model_toy_arm": [[0.5, 0.8, 0.2], [0.9, 0.2, 0.0], [0, 0.8, 0], [1.0, 0.0, 1.0], [0, 0, 1], [0.1, 0.9, 0.7]],
    "model_pneumatic_hand_only": [[0, 0, 1], [0.9, 0.2, 0.0], [0, 0.9, 0], [1.0, 0.0, 1.0], [0.1, 0.9, 0.7], [0.5, 0.8, 0.2]],
}


def _unit_range(x: torch.Tensor, dims) -> torch.Tensor:
    lo, hi = x.amin(dim=dims, keepdim=True), x.amax(dim=dims, keepdim=True)
    return (x - lo) / (hi - lo + 1e-10)


def compute_joint_sensitivity(jacobians: torch.Tensor, extrinsics: Optional[torch.Tensor] = None, mode: int = 0) -> torch.Tensor:
    """jacobian_color_map.py:53-91.  jacobians [..., H, W, 3A] (action-major, as rendered ``action_features``) ->
    per-command sensitivity [..., A, H, W] in [0,1]: the norm of each command's 3-vector (optionally rotated by
    ``extrinsics`` -- the vectors are directions, so only the rotation acts), min-max normalised per command image;
    ``mode=1`` inverts around 1.1 before the clip."""
    *lead, h, w, d = jacobians.shape
    vec = jacobians.reshape(*lead, h, w, d // 3, 3)
    if extrinsics is not None:
        vec = torch.einsum("...ij,...j->...i", extrinsics[..., :3, :3], vec)
    sens = _unit_range(torch.linalg.vector_norm(vec, dim=-1).movedim(-1, -3), (-2, -1))
    if mode == 1:
        sens = 1.1 - sens
    return sens.clip(0, 1)


def visualize_joint_sensitivity(sensitivity: torch.Tensor, color_map: torch.Tensor) -> np.ndarray:
    """jacobian_color_map.py:94-110.  sensitivity [..., A, H, W], color_map [3, A] -> uint8 image [..., H, W, 3]
    (each colour channel min-max normalised over the image, then inverted: strong response = saturated colour on
    white)."""
    mixed = torch.einsum("...ahw,ca->...chw", sensitivity, color_map.to(sensitivity))
    mixed = _unit_range(mixed, (-2, -1)).clip(0, 1)
    return ((1 - mixed.movedim(-3, -1)).cpu().numpy() * 255).astype(np.uint8)


def compute_joint_sensitivity_point_cloud(jacobians: torch.Tensor, extrinsics: Optional[torch.Tensor] = None,
                                          mode: int = 0) -> torch.Tensor:
    """jacobian_color_map.py:113-131.  jacobians [N, A, 3] -> [N, A] in [0,1] (per-command min-max over the points;
    ``extrinsics`` and ``mode`` are accepted and, as in the reference, unused)."""
    return _unit_range(torch.linalg.vector_norm(jacobians, dim=-1), (0,)).clip(0, 1)


def visualize_joint_sensitivity_point_cloud(sensitivity: torch.Tensor, color_map: torch.Tensor, mode: int = 0) -> torch.Tensor:
    """jacobian_color_map.py:134-160.  sensitivity [N, A], color_map [3, A] -> colours [N, 3]."""
    colors = _unit_range(sensitivity @ color_map.to(sensitivity).t(), (0,))
    if mode == 0:
        colors = 1 - colors.clip(0, 1)
    elif mode == 1:
        colors = (1.1 - colors).clip(0, 1)
    return colors
