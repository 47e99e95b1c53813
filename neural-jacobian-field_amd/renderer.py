"""Host orchestration of the fused HIP hot path (one call = ``Model.forward``'s rendering part).

``FusedRenderer`` owns the packed weight blobs and the hoisted feature map and issues, per forward:

    project_features (per image)  ->  proposal_forward (per proposal level)  ->  render_forward

which replaces ``Model.compute_proposal`` + ``decoder.forward`` + ``get_weights`` + ``render_*``
(reference ``models/model.py:316-396``).  All tensors stay on the device; nothing here
synchronises with the host.
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
    key = ("bins", num_samples, str(device))
    if key not in _CONST_CACHE:
        _CONST_CACHE[key] = torch.linspace(0.0, 1.0, num_samples + 1).to(device)
    return _CONST_CACHE[key]


def pdf_u_eval(num_samples: int, device) -> torch.Tensor:
    """Eval-mode ``u`` of PDFSampler (ray_samplers.py:402-408)."""
    key = ("u", num_samples, str(device))
    if key not in _CONST_CACHE:
        nb = num_samples + 1
        u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
        _CONST_CACHE[key] = (u + 1.0 / (2 * nb)).to(device)
    return _CONST_CACHE[key]


class FusedRenderer:
    """Packed weights + fused forward for one ``Model`` (jacobian_mlp decoder)."""

    def __init__(self, device: torch.device, num_proposal_networks: int = 1, action_dim: int = 8,
                 precision: Optional[str] = None):
        if torch.device(device).type != "cuda":
            raise ValueError("FusedRenderer needs a GPU device; the rendering hot path has no CPU fallback")
        hip.load_library()
        self.device = torch.device(device)
        self.n_prop = num_proposal_networks
        self.action_dim = action_dim
        self.precision = hip.DEFAULT_PRECISION if precision is None else precision
        hip.precision_code(self.precision)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.w_prop = [torch.empty(hip.RESNET_W_FLOATS, **f32) for _ in range(self.n_prop)]
        self.b_prop = [torch.empty(hip.RESNET_B_FLOATS, **f32) for _ in range(self.n_prop)]
        self.w_dec = torch.empty(2 * hip.RESNET_W_FLOATS + hip.COLOR_W_FLOATS, **f32)
        self.b_density = torch.empty(hip.RESNET_B_FLOATS, **f32)
        self.b_color = torch.empty(hip.COLOR_B_FLOATS, **f32)
        self.b_jacobian = torch.empty(hip.RESNET_B_FLOATS, **f32)
        self.n_maps = self.n_prop + 2
        self.gstride = hip.ZDIM * self.n_maps
        self.wz = torch.empty(512, self.gstride, **f32)
        self.bz = torch.empty(self.gstride, **f32)
        self.goff_density = hip.ZDIM * self.n_prop
        self.goff_jacobian = hip.ZDIM * (self.n_prop + 1)
        self.has_jacobian_mlp = False

    # ------------------------------------------------------------------ weights
    def load_weights(self, params: Dict[str, torch.Tensor]) -> None:
        """``params``: reference state-dict names -> device tensors (``decoder.*``, ``proposal_networks.i.*``)."""
        for i in range(self.n_prop):
            hip.pack_resnetfc(params, f"proposal_networks.{i}.density_head.", self.w_prop[i], self.b_prop[i],
                              self.wz, hip.ZDIM * i, self.bz, precision=self.precision)
        w_d = self.w_dec[: hip.RESNET_W_FLOATS]
        w_c = self.w_dec[hip.RESNET_W_FLOATS: hip.RESNET_W_FLOATS + hip.COLOR_W_FLOATS]
        w_j = self.w_dec[hip.RESNET_W_FLOATS + hip.COLOR_W_FLOATS:]
        hip.pack_resnetfc(params, "decoder.density_head.", w_d, self.b_density, self.wz, self.goff_density, self.bz,
                          precision=self.precision)
        hip.pack_color_head(params, "decoder.color_head.", w_c, self.b_color, precision=self.precision)
        self.has_jacobian_mlp = "decoder.jacobian_head.lin_in.weight" in params
        if self.has_jacobian_mlp:
            hip.pack_resnetfc(params, "decoder.jacobian_head.", w_j, self.b_jacobian, self.wz, self.goff_jacobian,
                              self.bz, precision=self.precision)
        else:
            self.wz[:, self.goff_jacobian:].zero_()
            self.bz[self.goff_jacobian:].zero_()

    # ------------------------------------------------------------------ per image
    def project(self, features: torch.Tensor) -> torch.Tensor:
        """Encoder output [B,512,Hf,Wf] -> hoisted channels-last map [B,Hf,Wf,384*(n_prop+2)]."""
        b, _, hf, wf = features.shape
        gmap = torch.empty(b, hf, wf, self.gstride, dtype=torch.float32, device=self.device)
        hip.project_features(features.contiguous(), self.wz, self.bz, gmap, precision=self.precision)
        return gmap

    # ------------------------------------------------------------------ per ray batch
    def render(self, gmap: torch.Tensor, origins: torch.Tensor, directions: torch.Tensor, ctxt_c2w: torch.Tensor,
               ctxt_k_norm: torch.Tensor, z_near: torch.Tensor, z_far: torch.Tensor,
               num_proposal_samples: Sequence[int], num_nerf_samples: int, trgt_c2w: Optional[torch.Tensor] = None,
               trgt_k_pix: Optional[torch.Tensor] = None, action: Optional[torch.Tensor] = None, anneal: float = 1.0,
               request: Optional[RenderRequest] = None, bins0: Optional[torch.Tensor] = None,
               u_list: Optional[Sequence[torch.Tensor]] = None, ctxt_w2c: Optional[torch.Tensor] = None,
               trgt_w2c: Optional[torch.Tensor] = None, clip_depth: bool = True,
               final_bins: Optional[torch.Tensor] = None, _events=None) -> RenderResult:
        """Eval-mode by default (shared linspace bins / mid-point u).  Training-mode stratified jitter is
        injected by the caller through ``bins0`` ([B,R,S0+1]) and ``u_list`` (one [B,R,S+1] per level).
        ``final_bins`` ([B,R,S+1] spacing bins) skips the proposal levels and renders exactly those samples."""
        req = request or RenderRequest()
        b, r = origins.shape[:2]
        dev, f32 = self.device, dict(dtype=torch.float32, device=self.device)
        origins, directions = origins.contiguous(), directions.contiguous()
        if ctxt_w2c is None:
            ctxt_w2c = hip.inverse(ctxt_c2w)
        if trgt_w2c is None and trgt_c2w is not None:
            trgt_w2c = hip.inverse(trgt_c2w)
        cams = hip.make_cameras(ctxt_w2c.contiguous(), ctxt_k_norm.contiguous(), z_near.contiguous(), z_far.contiguous(),
                                None if trgt_w2c is None else trgt_w2c.contiguous(),
                                None if trgt_k_pix is None else trgt_k_pix.contiguous(),
                                None if action is None else action.contiguous())
        fmap = hip.make_feature_map(gmap)

        levels = list(num_proposal_samples) + [num_nerf_samples]
        res = RenderResult(rgb=None, depth=None, optical_flow=None)
        bins = bins0.contiguous() if bins0 is not None else uniform_bins(levels[0], dev)
        if _events:
            _events[0].record()
        for lvl in range(self.n_prop if final_bins is None else 0):
            s_in, s_out = levels[lvl], levels[lvl + 1]
            u = u_list[lvl].contiguous() if u_list is not None else pdf_u_eval(s_out, dev)
            bins_out = torch.empty(b, r, s_out + 1, **f32)
            w_out = torch.empty(b, r, s_in, **f32) if req.sample_weights else None
            hip.proposal_forward(origins, directions, cams, fmap, hip.ZDIM * lvl, self.w_prop[lvl], self.b_prop[lvl],
                                 bins, s_in, u, s_out, anneal, bins_out, w_out, precision=self.precision)
            if req.sample_weights:
                res.bins_list.append(bins if bins.dim() > 1 else bins.expand(b, r, -1))
                res.weights_list.append(w_out[..., None])
            bins = bins_out

        if final_bins is not None:
            bins = final_bins.contiguous()
        if _events:
            _events[1].record()
        s = levels[-1]
        with_j = self.has_jacobian_mlp and action is not None
        outs: Dict[str, torch.Tensor] = {
            "rgb": torch.empty(b, r, 3, **f32),
            "depth": torch.empty(b, r, 1, **f32),
            "step_minmax": torch.empty(b, r, 2, **f32),
        }
        if with_j and trgt_w2c is not None:
            outs["flow"] = torch.empty(b, r, 2, **f32)
        if req.vis or req.sample_weights or req.per_sample:
            outs["weights"] = torch.empty(b, r, s, **f32)
        if req.vis:
            outs["pos"] = torch.empty(b, r, 3, **f32)
            outs["pos_warped"] = torch.empty(b, r, 3, **f32)
            if with_j:
                outs["action_features"] = torch.empty(b, r, 3 * self.action_dim, **f32)
        if req.per_sample:
            outs["density"] = torch.empty(b, r, s, 1, **f32)
            outs["color"] = torch.empty(b, r, s, 3, **f32)
            if with_j:
                outs["sample_flow"] = torch.empty(b, r, s, 3, **f32)
                outs["jacobian"] = torch.empty(b, r, s, 3 * self.action_dim, **f32)
        hip.render_forward(origins, directions, cams, fmap, self.goff_density, self.goff_jacobian, self.w_dec,
                           self.b_density, self.b_color, self.b_jacobian, bins, s, outs,
                           jacobian_kind=hip.JACOBIAN_MLP if with_j else hip.JACOBIAN_NONE, precision=self.precision)
        if _events:
            _events[2].record()
        depth = outs["depth"]
        if clip_depth:  # tensor-global clip of model.py:277
            depth = torch.clamp(depth, min=outs["step_minmax"][..., 0].min(), max=outs["step_minmax"][..., 1].max())
        res.rgb, res.depth, res.optical_flow = outs["rgb"], depth, outs.get("flow")
        if req.sample_weights:
            res.bins_list.append(bins)
            res.weights_list.append(outs["weights"][..., None])
        res.extras = {k: v for k, v in outs.items() if k not in ("rgb", "depth", "flow")}
        res.extras["final_bins"] = bins
        return res
