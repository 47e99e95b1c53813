"""Colour-mapped depth and optical flow for ``Model.patch_render`` (model.py:598-626), on the device.

The reference calls two third-party helpers here, neither vendored nor pinned (SURVEY.md section 8c -> parity
unpinned): ``nerfstudio.utils.colormaps.apply_depth_colormap`` and ``torchvision.utils.flow_to_image``.  Both are
restated from their published algorithms:
  * flow_to_image: the Middlebury colour wheel (Baker et al., 55 hues), flow normalised by the largest magnitude in
    the batch, uint8 output;
  * apply_depth_colormap: min-max normalisation followed by the "turbo" colour map.  nerfstudio indexes matplotlib's
    256-entry turbo table; matplotlib is not available to this build, so the published degree-5 polynomial fit of
    turbo (A. Mikhailov, Google, 2019) is evaluated instead -- within ~0.01 of the table in the interior of the range
    and up to ~0.1 at the two ends.  It is a display aid, not a numerical output.
"""

from __future__ import annotations

import math
from typing import Optional

import torch


def _color_wheel(device) -> torch.Tensor:
    """[55, 3] hue wheel: red->yellow (15), yellow->green (6), green->cyan (4), cyan->blue (11), blue->magenta (13),
    magenta->red (6); the rising/falling channel steps by floor(255 k / n)."""
    segments = [(15, 0, 1, True), (6, 1, 0, False), (4, 1, 2, True), (11, 2, 1, False), (13, 2, 0, True), (6, 0, 2, False)]
    rows = []
    for n, full, ramp, rising in segments:
        step = torch.floor(255 * torch.arange(n, dtype=torch.float32) / n)
        seg = torch.zeros(n, 3)
        seg[:, full] = 255
        seg[:, ramp] = step if rising else 255 - step
        rows.append(seg)
    return torch.cat(rows).to(device)


def flow_to_image(flow: torch.Tensor) -> torch.Tensor:
    """torchvision.utils.flow_to_image semantics: flow [N,2,H,W] or [2,H,W] float -> uint8 [N,3,H,W] / [3,H,W]."""
    single = flow.dim() == 3
    if single:
        flow = flow[None]
    if flow.dim() != 4 or flow.shape[1] != 2:
        raise ValueError(f"flow must have shape [2,H,W] or [N,2,H,W], got {tuple(flow.shape)}")
    flow = flow.float()
    mag = torch.linalg.vector_norm(flow, dim=1)
    flow = flow / (mag.max() + torch.finfo(flow.dtype).eps)
    mag = torch.linalg.vector_norm(flow, dim=1)
    wheel = _color_wheel(flow.device)
    n = wheel.shape[0]
    pos = (torch.atan2(-flow[:, 1], -flow[:, 0]) / math.pi + 1) / 2 * (n - 1)
    k0 = torch.floor(pos).long()
    k1 = torch.where(k0 + 1 == n, torch.zeros_like(k0), k0 + 1)
    frac = (pos - k0)[..., None]
    col = (1 - frac) * wheel[k0] / 255.0 + frac * wheel[k1] / 255.0        # [N,H,W,3]
    col = 1 - mag[..., None] * (1 - col)                                     # desaturate towards white at small flow
    img = torch.floor(255 * col).to(torch.uint8).permute(0, 3, 1, 2)
    return img[0] if single else img


_TURBO = ((0.13572138, 4.61539260, -42.66032258, 132.13108234, -152.94239396, 59.28637943),
          (0.09140261, 2.19418839, 4.84296658, -14.18503333, 4.27729857, 2.82956604),
          (0.10667330, 12.64194608, -60.58204836, 110.36276771, -89.90310912, 27.34824973))


def turbo(x: torch.Tensor) -> torch.Tensor:
    """x [...] in [0,1] -> rgb [...,3] in [0,1]: degree-5 polynomial fit of the turbo colour map."""
    x = x.clip(0, 1)
    powers = torch.stack([x ** k for k in range(6)], dim=-1)
    coeff = torch.tensor(_TURBO, dtype=x.dtype, device=x.device)
    return (powers @ coeff.t()).clip(0, 1)


def apply_depth_colormap(depth: torch.Tensor, accumulation: Optional[torch.Tensor] = None,
                         near_plane: Optional[float] = None, far_plane: Optional[float] = None) -> torch.Tensor:
    """nerfstudio.utils.colormaps.apply_depth_colormap semantics: depth [...,1] -> rgb [...,3]; the range defaults to
    the tensor's min / max (kept on the device: no host synchronisation), ``accumulation`` blends towards white."""
    near = depth.min() if near_plane is None else near_plane
    far = depth.max() if far_plane is None else far_plane
    d = torch.nan_to_num(((depth - near) / (far - near + 1e-10)).clip(0, 1))
    rgb = turbo(torch.floor(d[..., 0] * 255) / 255)          # same 256-level quantisation as the table lookup
    if accumulation is not None:
        rgb = rgb * accumulation + (1 - accumulation)
    return rgb
