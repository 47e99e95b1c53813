"""Ray generation and camera projections -- host-side mirror of ``rendering/geometry.py``.

Per-ray work (``get_world_rays_with_z``) runs in the HIP ray-generation kernel
(``njf_generate_rays``); the few per-camera 3x3/4x4 inverses and the pixel-centre grid are tiny
torch ops on the same device (plumbing, launched once per frame).
"""

from __future__ import annotations

from typing import Tuple

import torch

from . import hip


def get_pixel_coordinates(height: int, width: int, device: torch.device = torch.device("cpu")) -> Tuple[torch.Tensor, torch.Tensor]:
    """rendering/geometry.py:117-134: normalised pixel-centre xy [H,W,2] and int64 (row,col) selector [H,W,2]."""
    row = torch.arange(height, device=device)
    col = torch.arange(width, device=device)
    selector = torch.stack(torch.meshgrid(row, col, indexing="ij"), dim=-1)
    x = (col + 0.5) / width
    y = (row + 0.5) / height
    coordinates = torch.stack(torch.meshgrid(x, y, indexing="xy"), dim=-1)
    return coordinates, selector


def get_world_rays_with_z(coordinates_xy: torch.Tensor, intrinsics: torch.Tensor, cam2world: torch.Tensor):
    """rendering/geometry.py:170-203.  coordinates_xy [B,R,2] (normalised), intrinsics [B,3,3] (normalised),
    cam2world [B,4,4] -> origins [B,R,3], unit directions [B,R,3], z [B,R,1]."""
    b, r = coordinates_xy.shape[:2]
    dev = coordinates_xy.device
    origins = torch.empty(b, r, 3, dtype=torch.float32, device=dev)
    directions = torch.empty(b, r, 3, dtype=torch.float32, device=dev)
    z = torch.empty(b, r, 1, dtype=torch.float32, device=dev)
    hip.generate_rays(coordinates_xy.contiguous(), 0, 0, hip.inverse(intrinsics).contiguous(),
                      cam2world.contiguous(), origins, directions, z)
    return origins, directions, z


def get_world_rays(coordinates_xy, intrinsics, cam2world):
    """rendering/geometry.py:84-114."""
    o, d, _ = get_world_rays_with_z(coordinates_xy, intrinsics, cam2world)
    return o, d


def full_frame_rays(height: int, width: int, intrinsics: torch.Tensor, cam2world: torch.Tensor):
    """All H*W rays of a frame without materialising the coordinate grid (pixel centres are generated
    in-kernel): equivalent to get_world_rays_with_z(get_pixel_coordinates(H,W).view(1,-1,2), K, c2w)."""
    b = intrinsics.shape[0]
    dev = intrinsics.device
    r = height * width
    origins = torch.empty(b, r, 3, dtype=torch.float32, device=dev)
    directions = torch.empty(b, r, 3, dtype=torch.float32, device=dev)
    z = torch.empty(b, r, 1, dtype=torch.float32, device=dev)
    hip.generate_rays(None, height, width, hip.inverse(intrinsics).contiguous(), cam2world.contiguous(),
                      origins, directions, z)
    return origins, directions, z


def denormalize_intrinsics(intrinsics: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """utils/convention.py:110-125."""
    k = intrinsics.clone()
    k[..., 0, :] *= width
    k[..., 1, :] *= height
    return k
