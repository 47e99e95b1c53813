"""Low-level driver of the fused hot path for callers that hold a feature map instead of an image.

``FusedRenderer`` is a thin facade over ``Model``: it builds a ``Model`` whose encoder entry is ``"precomputed"``
(encoder.EncoderPrecomputed) and forwards to ``Model._fused_render`` -- the ONE orchestration of the fused path
(project the feature map per network, one ``njf_proposal_forward`` per level, ``njf_render_forward``).  It adds what
tests and tools need and the reference API has no place for: per-sample outputs, injected final bins, inverses
computed by the caller.  All tensors stay on the device; nothing here synchronises with the host.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

from . import hip


@dataclass
class RenderRequest:
    """Which optional outputs ``render`` should materialise (per-ray outputs rgb/depth/flow are always produced)."""
    vis: bool = False            # pos / pos_warped / action_features / weights / steps   (ModelVisOutput)
    sample_weights: bool = False  # per-level weights + bins (ModelTrainingOutput)
    per_sample: bool = False     # density / color / flow / jacobian per sample (DecoderOutput, encode_image)


@dataclass
class RenderResult:
    rgb: torch.Tensor
    depth: torch.Tensor
    optical_flow: Optional[torch.Tensor]
    bins_list: List[torch.Tensor] = field(default_factory=list)      # spacing bins per level [B,R,S+1]
    weights_list: List[torch.Tensor] = field(default_factory=list)   # [B,R,S,1] per level (when requested)
    extras: Dict[str, torch.Tensor] = field(default_factory=dict)


_CONST_CACHE: Dict[tuple, torch.Tensor] = {}


def uniform_bins(num_samples: int, device) -> torch.Tensor:
    """``torch.linspace(0, 1, S+1)`` evaluated on the CPU (bit-identical to the reference path), cached on device."""
    key = ("bins", num_samples, hip.device_key(device))
    if key not in _CONST_CACHE:
        _CONST_CACHE[key] = torch.linspace(0.0, 1.0, num_samples + 1).to(device)
    return _CONST_CACHE[key]


def pdf_u_eval(num_samples: int, device) -> torch.Tensor:
    """Eval-mode ``u`` of PDFSampler (ray_samplers.py:402-408)."""
    key = ("u", num_samples, hip.device_key(device))
    if key not in _CONST_CACHE:
        nb = num_samples + 1
        u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
        _CONST_CACHE[key] = (u + 1.0 / (2 * nb)).to(device)
    return _CONST_CACHE[key]


class FusedRenderer:
    """Feature map in, rendered rays out (``jacobian_mlp`` decoder by default): see the module docstring."""

    def __init__(self, device: torch.device, num_proposal_networks: int = 1, action_dim: int = 8,
                 precision: Optional[str] = None, proposal_precision: Optional[str] = None, decoder: str = "jacobian_mlp"):
        if torch.device(device).type != "cuda":
            raise ValueError("FusedRenderer needs a GPU device; the rendering hot path has no CPU fallback")
        hip.load_library()
        from .config import model_cfg_from_dict
        from .model import Model
        self.device = torch.device(device)
        self.n_prop = num_proposal_networks
        self.action_dim = action_dim
        cfg = model_cfg_from_dict({"action_dim": action_dim, "encoder": {"name": "precomputed"},
                                   "rendering": {"num_proposal_samples": [64] * num_proposal_networks, "num_nerf_samples": 64},
                                   "action_decoder": {"name": decoder}})
        self.model = Model(cfg).to(self.device).eval().requires_grad_(False)
        self.model.set_precision(hip.DEFAULT_PRECISION if precision is None else precision, proposal_precision)
        self.precision = self.model.decoder.precision

    # ------------------------------------------------------------------ weights
    def load_weights(self, params: Dict[str, torch.Tensor]) -> None:
        """``params``: reference state-dict names -> tensors (``decoder.*``, ``proposal_networks.i.*``; encoder entries,
        if present, are ignored)."""
        own = self.model.state_dict()
        missing = [k for k in own if k not in params]
        if missing:
            raise KeyError(f"FusedRenderer.load_weights: missing {missing[:4]}{'...' if len(missing) > 4 else ''}")
        self.model.load_state_dict({k: params[k] for k in own}, strict=True)

    # ------------------------------------------------------------------ per ray batch
    def render(self, features: torch.Tensor, origins: torch.Tensor, directions: torch.Tensor, ctxt_c2w: torch.Tensor,
               ctxt_k_norm: torch.Tensor, z_near: torch.Tensor, z_far: torch.Tensor,
               num_proposal_samples: Sequence[int], num_nerf_samples: int, trgt_c2w: Optional[torch.Tensor] = None,
               trgt_k_pix: Optional[torch.Tensor] = None, action: Optional[torch.Tensor] = None, anneal: float = 1.0,
               request: Optional[RenderRequest] = None, ctxt_w2c: Optional[torch.Tensor] = None,
               trgt_w2c: Optional[torch.Tensor] = None, clip_depth: bool = True,
               final_bins: Optional[torch.Tensor] = None) -> RenderResult:
        """Eval-mode rendering of ``Model.forward`` (models/model.py:316-396) from a feature map.  ``final_bins``
        ([B,R,S+1] spacing bins) skips the proposal levels and renders exactly those samples."""
        from .model import CameraInput, RenderingInput, RobotInput
        req = request or RenderRequest()
        m = self.model
        if len(num_proposal_samples) != self.n_prop:
            raise ValueError(f"expected {self.n_prop} proposal level(s), got {list(num_proposal_samples)}")
        if trgt_c2w is None or trgt_k_pix is None or action is None:
            raise ValueError("FusedRenderer.render needs the target camera and the robot action (Model.forward's inputs)")
        m.cfg.rendering.num_proposal_samples = tuple(num_proposal_samples)
        m.cfg.rendering.num_nerf_samples = num_nerf_samples
        m.proposal_sampler.num_proposal_samples_per_ray = tuple(num_proposal_samples)
        m.proposal_sampler.num_nerf_samples_per_ray = num_nerf_samples
        m.proposal_sampler.set_anneal(anneal)
        m.encoder.set_features(features)
        cam = CameraInput(input_image=None, ctxt_extrinsics=ctxt_c2w, ctxt_intrinsics=ctxt_k_norm, trgt_extrinsics=trgt_c2w,
                          trgt_intrinsics=trgt_k_pix)
        rin = RenderingInput(origins, directions, z_near, z_far)
        with torch.no_grad():
            outs, bins, weights_list, bins_list, _ = m._fused_render(
                cam, rin, RobotInput(action), m._encode_for_render(None), want_lists=req.sample_weights, want_vis=req.vis,
                want_samples=req.per_sample, want_sample_outputs=req.per_sample, final_bins=final_bins, ctxt_w2c=ctxt_w2c,
                trgt_w2c=trgt_w2c, clip_depth=clip_depth)
        res = RenderResult(rgb=outs["rgb"], depth=outs["depth"], optical_flow=outs["flow"])
        if req.sample_weights:
            res.bins_list = list(bins_list) + [bins]
            res.weights_list = list(weights_list) + [outs["weights"][..., None]]
        res.extras = {k: v for k, v in outs.items() if k not in ("rgb", "depth", "flow") and torch.is_tensor(v)}
        res.extras["final_bins"] = bins
        return res
