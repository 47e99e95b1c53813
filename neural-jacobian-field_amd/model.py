"""``Model`` -- host-side mirror of ``models/model.py`` (same dataclasses, methods and state-dict
names) whose rendering runs in the fused HIP kernels.

Reference: ``models/model.py:35-144`` (I/O records), ``:147-213`` (construction, annealing),
``:215-314`` (ray bundle / proposal / render_*), ``:316-396`` (forward), ``:398-525`` (inference helpers),
``:527-628`` (patch_render).

``Model.forward`` picks its path from the autograd state: an inference pass (in-kernel compositing) when no
gradient is required, otherwise one of the two training paths of ``training.py`` (action mode: gradients to the
Jacobian head; perception mode: gradients to encoder, density / colour heads and proposal networks).
"""

from __future__ import annotations

import functools
import os
import warnings
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .config import ModelCfg, RenderingCfg  # noqa: F401  (re-exported like the reference module)
from .decoder import DensityHeadOutput, PixelEncoding, _cameras, get_action_decoder, get_density_decoder
from .encoder import get_encoder
from .ray_samplers import ProposalNetworkSampler, RayBundle, RaySamples, UniformSampler


# ---- I/O records (model.py:56-144) -------------------------------------------------------
@dataclass
class CameraInput:
    input_image: torch.Tensor      # [B,C,H,W]
    ctxt_extrinsics: torch.Tensor  # [B,4,4]
    ctxt_intrinsics: torch.Tensor  # [B,3,3] normalised
    trgt_extrinsics: torch.Tensor  # [B,4,4]
    trgt_intrinsics: torch.Tensor  # [B,3,3] pixels


@dataclass
class RenderingInput:
    origins: torch.Tensor     # [B,R,3]
    directions: torch.Tensor  # [B,R,3]
    z_near: torch.Tensor      # [B]
    z_far: torch.Tensor       # [B]


@dataclass
class RobotInput:
    robot_action: torch.Tensor  # [B,A]


@dataclass
class ModelInput:
    camera_input: CameraInput
    rendering_input: RenderingInput
    robot_input: RobotInput


@dataclass
class ModelTarget:
    rgb: torch.Tensor
    depth: torch.Tensor
    optical_flow: Optional[torch.Tensor]
    visible_mask: Optional[torch.Tensor]


@dataclass
class ModelStandardOutput:
    rgb: torch.Tensor           # [B,R,3]
    depth: torch.Tensor         # [B,R,1]
    optical_flow: torch.Tensor  # [B,R,2]


@dataclass
class ModelTrainingOutput:
    weights_list: List[torch.Tensor]
    ray_samples_list: List[RaySamples]


@dataclass
class ModelVisOutput:
    action_features: torch.Tensor
    ray_positions: torch.Tensor
    ray_positions_warped: torch.Tensor
    weights: torch.Tensor
    steps: torch.Tensor


@dataclass
class ModelOutput:
    standard_output: ModelStandardOutput
    training_output: Optional[ModelTrainingOutput]
    vis_output: Optional[ModelVisOutput]


@dataclass
class ModelInferenceEncoding:
    density: torch.Tensor
    action_features: torch.Tensor
    weights: torch.Tensor
    ray_samples_positions: torch.Tensor


@dataclass
class RenderingOutput:
    rgb: torch.Tensor
    depth_raw: torch.Tensor
    depth_rgb: Optional[torch.Tensor]
    flow_raw: torch.Tensor
    flow_rgb: Optional[torch.Tensor]
    ray_positions: torch.Tensor
    ray_positions_warped: torch.Tensor
    action_features: torch.Tensor
    steps: torch.Tensor
    weights: torch.Tensor


class Model(nn.Module):
    def __init__(self, cfg: ModelCfg):
        super().__init__()
        self.cfg = cfg
        self.encoder = get_encoder(cfg.encoder)
        self.decoder = get_action_decoder(cfg.action_decoder, action_dim=cfg.action_dim,
                                          encoder_dim=self.encoder.get_output_dim())
        n_prop = len(cfg.rendering.num_proposal_samples)
        self.proposal_networks = nn.ModuleList(
            [get_density_decoder(cfg.density_decoder, encoder_dim=self.encoder.get_output_dim()) for _ in range(n_prop)])
        self.density_fns = [net.get_density for net in self.proposal_networks]
        r = cfg.rendering
        update_schedule = lambda step: np.clip(np.interp(step, [0, r.proposal_warmup], [0, r.proposal_update_every]), 1,
                                               r.proposal_update_every)
        self.proposal_sampler = ProposalNetworkSampler(
            num_nerf_samples_per_ray=r.num_nerf_samples, num_proposal_samples_per_ray=tuple(r.num_proposal_samples),
            num_proposal_network_iterations=n_prop, single_jitter=r.single_jitter, update_sched=update_schedule,
            initial_sampler=UniformSampler(single_jitter=r.single_jitter))
        # render_depth's clip (model.py:277) takes its bounds from the WHOLE step tensor; under ray sharding
        # parallel.enable_ray_sharding() swaps in the all-reduced form
        self.depth_clip = self._local_depth_clip
        # Operating-range guard of the fp16-carried MFMA precisions (VERDICT r02 #7 / ADVICE r02): the first forward pass
        # after the weights were (re)loaded -- and every `range_check_interval`-th forward pass whose weights differ from
        # the ones last checked (training) -- measures the largest matrix operand on the exact-fp32 path and moves the
        # model to "f32" with a warning if fp16 could overflow.  `auto_range_check = False` turns it off
        # (calibrate_precision stays available as the explicit form).
        # Ray-sharded steps (parallel.ShardedFrameStep) hand the forward pass its output buffers -- views of the packet the
        # rank contributes to the step's one collective -- and the loss targets; the render kernel then also writes the
        # per-workgroup partials of the frame-level reductions and the depth clip is deferred to njf_assemble_frame.
        self.frame_io: Optional[Dict[str, torch.Tensor]] = None
        # Training forwards at GIVEN final bins ([B,R,S+1] spacing bins; None = the proposal sampler places them, as in the
        # reference): the proposal levels are skipped and exactly these samples are rendered and differentiated.  For gradient
        # tests that must not inherit the inverse CDF's placement noise (tests/test_training_fixed_bins_gpu.py); the reference
        # has no such switch.
        self.training_final_bins: Optional[torch.Tensor] = None
        self._inverse_cache: Dict[str, tuple] = {}
        self.inverse_cache_enabled = True
        self._joint: Dict[str, object] = {"features": None}   # ONE per-image projection for all networks of a frame (_joint_hoist)
        self.auto_range_check = os.environ.get("NJF_AUTO_RANGE_CHECK", "1") != "0"   # (off: single-precision experiment libraries)
        if not self.auto_range_check:
            warnings.warn("NJF_AUTO_RANGE_CHECK=0: the fp16 range check of the first forward pass is OFF (meant for kernel "
                          "experiment libraries that hold one precision only); checkpoints whose activations exceed the fp16 "
                          "range will not fall back to exact fp32 products", RuntimeWarning, stacklevel=2)
        self.range_check_interval = 100
        self._range_checked = None      # weights signature of the last check
        self._range_pending = True      # weights replaced wholesale (construction, load_state_dict)
        self._range_forwards = 0
        self.set_precision(hip.DEFAULT_PRECISION)

    @staticmethod
    def _local_depth_clip(depth: torch.Tensor, step_minmax: torch.Tensor) -> torch.Tensor:
        return torch.clamp(depth, min=step_minmax[..., 0].min(), max=step_minmax[..., 1].max())

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._range_pending = True
        return out

    def _weights_signature(self):
        params = self.__dict__.get("_range_params")
        if params is None:   # Parameter objects keep their identity across .to() / load_state_dict / optimiser steps
            params = [p for n, p in self.named_parameters() if not n.startswith("encoder.")]
            self.__dict__["_range_params"] = params
        return tuple((p.data_ptr(), p._version) for p in params)

    def _maybe_check_range(self, camera_input, rendering_input, robot_input) -> None:
        """See __init__.  Costs nothing while the weights are the ones last checked (one tuple comparison); a check is a few
        small fused launches on a prefix of the rays and one host synchronisation -- so it never runs inside a HIP-graph
        capture of a warmed-up model."""
        if not self.auto_range_check:
            return
        if self.decoder.precision == "f32" and all(m.precision == "f32" for m in self.proposal_networks):
            return
        sig = self._weights_signature()
        if sig == self._range_checked and not self._range_pending:
            return
        self._range_forwards += 1
        if not self._range_pending and self._range_checked is not None and self._range_forwards < self.range_check_interval:
            return
        if torch.cuda.is_current_stream_capturing():
            return   # a check synchronises with the host; it runs on the first eager forward instead
        self._range_pending, self._range_forwards, self._range_checked = False, 0, sig
        self.calibrate_precision(camera_input, rendering_input, robot_input)
        # the same periodic check looks after the TF32-class backward forms (training.py: fp16 has a range, TF32 does not)
        from . import training
        if training.reduced_backward_overflowed(rendering_input.origins.device):
            import warnings
            warnings.warn("njf: a reduced-precision backward pass (f16x2 chain / 16-bit training storage) produced non-finite "
                          "gradients since the last check -- its fp16 range was exceeded; switching the backward pass to exact fp32 "
                          "(training.set_backward_precision('f32'), set_storage_precision('f32')).  Gradients of the steps in "
                          "between were non-finite: reload the last checkpoint.", RuntimeWarning)
            training.set_backward_precision("f32")
            training.set_storage_precision("f32")

    def _inverse(self, name: str, m: torch.Tensor) -> torch.Tensor:
        """hip.inverse(m), kept while ``m`` is the same tensor object at the same version (a camera rig that does not move
        costs no launch per forward pass; the cache holds a reference to ``m``, so its address cannot be recycled)."""
        if not self.inverse_cache_enabled:
            # a step that is being recorded into a HIP graph (parallel.ShardedFrameStep.capture, warm-up included) must
            # LAUNCH its inverses: the graph's static camera tensors are refilled in place between replays, and an inverse
            # served from the cache would be baked into the graph as a constant
            return hip.inverse(m)
        hit = self._inverse_cache.get(name)
        if hit is not None and hit[0] is m and hit[1] == m._version:
            return hit[2]
        inv = hip.inverse(m)
        if not torch.cuda.is_current_stream_capturing():   # a graph-private tensor must not outlive its capture
            self._inverse_cache[name] = (m, m._version, inv)
        return inv

    def reset_image_cache(self) -> "Model":
        """Forget the hoisted feature maps.  They are cached per feature TENSOR (object + version counter), which is right
        for a control loop that re-renders one image; a caller that refills the same tensor object behind torch's back, or
        a benchmark that must include the per-image projection in every step, calls this first."""
        for m in [self.decoder, *self.proposal_networks]:
            m._hoist.key = None
            m._joint_view = None
        self._joint["features"] = None
        return self

    def _joint_hoist(self, features):
        """The hoisted maps of EVERY network of the frame -- proposal nets and decoder -- as channel ranges of ONE map
        [B,Hf,Wf,N] produced by ONE projection (round 3: one launch instead of one per network reads the feature map once;
        0.10 -> ~0.06 ms per image, which is the part of a ray-sharded step that does not shrink with the shard).  Returns
        (map, [first channel of each proposal net], first channel of the decoder) or None when the networks cannot share a
        projection (flow_mlp adds a per-image action bias to its block; an exact-fp32 network next to split-precision ones
        uses another projection kernel) -- the callers then fall back on the per-network maps."""
        from .decoder import ActionDecoderJacobian
        nets = [*self.proposal_networks, self.decoder]
        if type(self.decoder).hoisted_map is not ActionDecoderJacobian.hoisted_map:
            return None
        precs = {n.precision for n in nets}
        if ("f32" in precs or "f16" in precs) and len(precs) > 1:   # (another projection kernel / another map element type)
            return None
        for n in nets:
            n.packed()
        versions = tuple(n._packed_version for n in nets)
        c = self._joint
        if c.get("wkey") != versions or c["wz"].device != nets[0]._wz.device:
            c["wz"] = torch.cat([n._wz for n in nets], dim=1).contiguous()
            c["bz"] = torch.cat([n._bz for n in nets]).contiguous()
            widths = [n._wz.shape[1] for n in nets]
            c["bases"] = [sum(widths[:i]) for i in range(len(nets))]
            c["wkey"], c["features"] = versions, None
        key = (features._version, tuple(features.shape), versions)
        if c["features"] is not features or c.get("key") != key:
            from .encoder import FeaturePyramid
            b, _, hf, wf = features.shape
            gmap = torch.empty(b, hf, wf, c["wz"].shape[1], dtype=hip.map_dtype(self.decoder.precision), device=c["wz"].device)
            if isinstance(features, FeaturePyramid):
                hip.project_pyramid(features.levels, c["wz"], c["bz"], gmap, precision=self.decoder.precision)
            else:
                hip.project_features(features.contiguous(), c["wz"], c["bz"], gmap, precision=self.decoder.precision)
            c["features"], c["key"], c["gmap"] = features, key, gmap
            # point queries of a network (decoder.forward / compute_density / get_density, called directly as in the reference)
            # read ITS channel range of this map instead of projecting one of their own: each network gets a plain record
            # (no back-reference to the model: a deep copy of the model or of a network must not see this model's maps)
            for n, base in zip(nets, c["bases"]):
                n._joint_view = (features, features._version, tuple(features.shape), n._packed_version, gmap, base)
        return c["gmap"], c["bases"][:-1], c["bases"][-1]

    def _joint_lookup(self, features):
        """The joint map of `features` IF one is already there (never projects): (map, [proposal bases], decoder base) or
        None.  Point queries (decoder.forward / compute_density / get_density) and renders at given bins read their channel
        range of it instead of projecting a map of their own next to it (a second per-image projection and a
        duplicate map when forward and compute_density were mixed on one image)."""
        c = self._joint
        if c.get("features") is not features or features is None or "wkey" not in c:
            return None
        nets = [*self.proposal_networks, self.decoder]
        for n in nets:
            n.packed()
        versions = tuple(n._packed_version for n in nets)
        if c["wkey"] != versions or c.get("key") != (features._version, tuple(features.shape), versions):
            return None
        return c["gmap"], c["bases"][:-1], c["bases"][-1]

    def set_precision(self, precision: str, proposal_precision: Optional[str] = None,
                      jacobian_precision: Optional[str] = None) -> "Model":
        """MFMA precision of the fused MLPs (weights are re-packed lazily):

        * ``"f32"``   exact fp32 products (v_mfma_f32_32x32x2_f32);
        * ``"f16x2"`` fp32 operands split into fp16 hi + lo, hi*hi + hi*lo + lo*hi, fp32 accumulate;
        * ``"f16f6"`` the same hi*hi, the two 2^-11-sized correction products in block-scaled fp6 -- half the matrix time
          of f16x2 at ~1.5e-5 relative error per network;
        * ``"f16"``   PLAIN fp16 products (weights and layer inputs rounded to fp16, fp32 accumulation, fp16 hoisted maps):
          a reduced-precision INFERENCE mode with its own stated tolerance (~1e-3 norm-wise per network output, DESIGN.md
          section 5; BASELINE config 5), proposal networks included unless ``proposal_precision`` says otherwise.  Never
          the default; training forwards refuse it.

        ``precision`` applies to the networks of the final pass (density, colour, Jacobian / flow head; the head can be
        given its own split precision through ``jacobian_precision``).  ``proposal_precision`` applies to the proposal
        networks and defaults to ``precision`` -- except under ``"f16f6"``, where the proposal networks stay on
        ``"f16x2"``: sample PLACEMENT feeds a positional encoding with a 2*pi*512 gain, so an error of 1e-5 in the proposal
        weights shows up as 3e-4 in depth / flow, while the same error inside the final pass (at given sample locations)
        stays 1e-5 (measured: tools/prec_eval.py, tools/prec_mix_eval.py)."""
        hip.precision_code(precision)
        if proposal_precision is None:
            proposal_precision = hip.proposal_precision_for(precision)
        if jacobian_precision is None:
            jacobian_precision = precision
        hip.precision_code(proposal_precision)
        hip.precision_code(precision, jacobian_precision)
        self.decoder.precision = precision
        self.decoder.jacobian_precision = None if jacobian_precision == precision else jacobian_precision
        for m in self.proposal_networks:
            m.precision = proposal_precision
        return self

    @torch.no_grad()
    def activation_range(self, camera_input: "CameraInput", rendering_input: "RenderingInput", robot_input: "RobotInput",
                         max_points: int = 1 << 16) -> Dict[str, float]:
        """Largest |input of any matrix product| per network on (a prefix of) the given rays, measured on the exact-fp32
        MFMA path through the training-forward activation dumps: the ReLU'd input of every ResnetFC layer, the positional
        encoding, the colour head's inputs and hidden layers; plus the largest |weight|.  These are the values the
        split-precision modes convert to fp16."""
        s_max = max(self.cfg.rendering.num_nerf_samples, *self.cfg.rendering.num_proposal_samples)
        b = rendering_input.origins.shape[0]
        rays = max(1, min(rendering_input.origins.shape[1], max_points // max(1, b * s_max)))
        rin = RenderingInput(rendering_input.origins[:, :rays].contiguous(), rendering_input.directions[:, :rays].contiguous(),
                             rendering_input.z_near, rendering_input.z_far)
        saved = (self.decoder.precision, [m.precision for m in self.proposal_networks], self.decoder.jacobian_precision)
        modes = [(m, m.training) for m in self.modules()]   # per module: a caller may run model.train() with encoder.eval()
        self.set_precision("f32")
        self.eval()
        try:
            features = self._encode_for_render(camera_input.input_image)
            if not torch.is_tensor(features):      # the dump path projects the concatenated map
                features = self.encoder.forward(camera_input.input_image)
            out = {}
            outs, *_ = self._fused_render(camera_input, rin, robot_input, features, want_lists=True, want_vis=False,
                                          want_samples=False, dump_perception=True)
            out["density_head"] = float(outs["den_act"].abs().max())
            out["color_head"] = float(torch.maximum(outs["col_in"].abs().max(), outs["col_act"].abs().max()))
            out["positional_encoding"] = float(outs["jac_pe"].abs().max())
            for i, d in enumerate(outs["proposal_dumps"]):
                out[f"proposal_networks.{i}"] = float(d["act"].abs().max())
            if self.decoder.JACOBIAN_KIND == hip.JACOBIAN_MLP:
                outs, *_ = self._fused_render(camera_input, rin, robot_input, features, want_lists=False, want_vis=True,
                                              want_samples=False, dump_jacobian=True)
                out["jacobian_head"] = float(outs["jac_act"].abs().max())
            out["weights"] = max(float(p.detach().abs().max()) for n, p in self.named_parameters() if not n.startswith("encoder."))
        finally:
            self.decoder.precision, self.decoder.jacobian_precision = saved[0], saved[2]
            for m, p in zip(self.proposal_networks, saved[1]):
                m.precision = p
            for m, mode in modes:
                m.training = mode
        return out

    def calibrate_precision(self, camera_input: "CameraInput", rendering_input: "RenderingInput", robot_input: "RobotInput",
                            fallback: str = "f32", headroom: float = 4.0) -> str:
        """Operating-range check of the split-precision modes, to be run once per checkpoint (or every few hundred
        optimiser steps) on a representative batch.  "f16x2" / "f16f6" carry the leading bits of every fp32 operand in
        fp16: values up to 65,504 are represented to fp32 accuracy, anything larger becomes inf -- and the fused kernels'
        integer ReLU can turn the resulting negative NaNs into zeros, so an overflow is NOT guaranteed to surface as a
        non-finite output.  This measures the largest operand on the exact-fp32 path (``activation_range``) and switches
        the model to ``fallback`` (with a warning) when it comes within ``headroom`` x of fp16's limit.  Returns the
        precision in force afterwards.  Costs a few small fused launches and one host synchronisation."""
        import warnings

        current = self.decoder.precision
        if current == fallback and all(m.precision == fallback for m in self.proposal_networks):
            return current
        self._range_pending, self._range_forwards, self._range_checked = False, 0, self._weights_signature()
        ranges = self.activation_range(camera_input, rendering_input, robot_input)
        worst = max(ranges, key=ranges.get)
        if not (ranges[worst] * headroom < 65504.0):   # also catches NaN
            warnings.warn(f"njf: |operand| reaches {ranges[worst]:.3g} in {worst} on the calibration batch, within {headroom:g}x of "
                          f"fp16's range (65,504): the {current!r} MFMA precision is not safe for this checkpoint.  Switching "
                          f"this model to {fallback!r}.", RuntimeWarning)
            self.set_precision(fallback)
            return fallback
        return current

    # ---- schedule hooks (model.py:201-213) ---------------------------------------------
    def step_before_iter(self, step):
        r = self.cfg.rendering
        if r.use_proposal_weight_anneal:
            frac = np.clip(step / r.proposal_weights_anneal_max_num_iters, 0, 1)
            b = r.proposal_weights_anneal_slope
            self.proposal_sampler.set_anneal((b * frac) / ((b - 1) * frac + 1))

    def step_after_iter(self, step):
        if self.cfg.rendering.use_proposal_weight_anneal:
            self.proposal_sampler.step_cb(step)

    # ---- pieces of the forward (model.py:215-314) ---------------------------------------
    def compute_ray_bundle(self, rendering_input: RenderingInput) -> RayBundle:
        # [B,R,1] views of the per-scene bounds (the reference multiplies a ones tensor, model.py:215-226: same values,
        # two kernels and two [B,R,1] tensors more per call)
        shape = (*rendering_input.origins.shape[:-1], 1)
        return RayBundle(origins=rendering_input.origins, directions=rendering_input.directions,
                         nears=rendering_input.z_near[:, None, None].expand(shape),
                         fars=rendering_input.z_far[:, None, None].expand(shape))

    def compute_proposal(self, ray_bundle: RayBundle, pixel_encoding: PixelEncoding):
        """model.py:228-255 through the generic sampler route (public pieces, arbitrary density_fns)."""
        fns = [functools.partial(fn, pixel_encoding=pixel_encoding) for fn in self.density_fns]
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler.generate_ray_samples(ray_bundle, density_fns=fns)
        positions = ray_samples.get_positions()
        directions = ray_bundle.directions[..., None, :].expand(positions.shape)
        return ray_samples, positions, directions, weights_list, ray_samples_list

    @staticmethod
    def render_rgb(rgb, weights, bg_color=None):
        comp = torch.sum(weights * rgb, dim=-2)
        if bg_color is not None:
            comp = comp + (1.0 - torch.sum(weights, dim=-2)) * bg_color
        return comp

    @staticmethod
    def render_depth(weights, ray_samples: RaySamples):
        steps = (ray_samples.starts + ray_samples.ends) / 2
        depth = torch.sum(weights * steps, dim=-2) / (torch.sum(weights, -2) + 1e-10)
        return torch.clip(depth, steps.min(), steps.max()), steps

    @staticmethod
    def render_action_features(action_features, weights):
        return torch.sum(weights * action_features, dim=-2)

    @staticmethod
    def _project(points, trgt_extrinsics, trgt_intrinsics):
        hom = torch.cat([points, torch.ones_like(points[..., :1])], dim=-1)
        cam = torch.einsum("...ij,...j->...i", hip.inverse(trgt_extrinsics)[..., None, :, :], hom)
        xyw = torch.einsum("...ij,...j->...i", trgt_intrinsics.unsqueeze(1), cam[..., :3])
        return (xyw / (xyw[..., -1:] + 1e-9))[..., :2]

    @staticmethod
    def render_optical_flow(weights, ray_positions, scene_flow, trgt_extrinsics, trgt_intrinsics):
        """model.py:288-314.  Stand-alone form on caller-supplied per-sample tensors (inverse-dynamics loops
        differentiate through it, so it stays in torch); Model.forward composites in-kernel."""
        warped = ray_positions + scene_flow
        pos = torch.sum(weights * ray_positions, dim=-2)
        pos_w = torch.sum(weights * warped, dim=-2)
        uv = Model._project(pos, trgt_extrinsics, trgt_intrinsics)
        uv_w = Model._project(pos_w, trgt_extrinsics, trgt_intrinsics)
        return uv_w - uv, pos, pos_w

    # ---- fused forward -------------------------------------------------------------------
    def _fused_render(self, camera_input: CameraInput, rendering_input: RenderingInput, robot_input: RobotInput,
                      features: torch.Tensor, want_lists: bool, want_vis: bool, want_samples: bool,
                      dump_jacobian: bool = False, dump_perception: bool = False, want_sample_outputs: bool = False,
                      final_bins: Optional[torch.Tensor] = None, ctxt_w2c: Optional[torch.Tensor] = None,
                      trgt_w2c: Optional[torch.Tensor] = None, clip_depth: bool = True):
        """THE orchestration of the fused path (one njf_proposal_forward per level, then njf_render_forward); every
        public entry point -- forward, encode_image, patch_render, the training paths, renderer.FusedRenderer -- ends here.
        ``want_sample_outputs`` adds per-sample colour and scene flow; ``final_bins`` ([B,R,S+1] spacing bins) skips the
        proposal levels and renders exactly those samples; ``ctxt_w2c`` / ``trgt_w2c`` are inverses the caller already has."""
        if ctxt_w2c is None:  # one inverse per camera rig, shared by the proposal levels and the final pass
            ctxt_w2c = self._inverse("ctxt", camera_input.ctxt_extrinsics)
        enc = PixelEncoding(features=features, extrinsics=camera_input.ctxt_extrinsics,
                            intrinsics=camera_input.ctxt_intrinsics, action=robot_input.robot_action, extrinsics_inv=ctxt_w2c)
        ray_bundle = self.compute_ray_bundle(rendering_input)
        self.proposal_sampler.train(self.training)
        proposal_dumps = [] if dump_perception else None
        # one projection for all networks of the frame; a render at GIVEN bins runs no proposal level, so it only reads a
        # joint map that already exists and otherwise projects the decoder's channels alone
        joint = self._joint_hoist(features) if final_bins is None else self._joint_lookup(features)
        joint_fmap = None if joint is None else hip.make_feature_map(joint[0])
        if final_bins is None:
            bins, weights_list, bins_list = self.proposal_sampler.generate_ray_samples_fused(
                ray_bundle, list(self.proposal_networks), enc, rendering_input.z_near, rendering_input.z_far, want_lists,
                dump_out=proposal_dumps, feature_maps=None if joint is None else [(joint_fmap, off) for off in joint[1]])
        else:
            bins, weights_list, bins_list = final_bins.contiguous(), [], []
        o, d = rendering_input.origins.contiguous(), rendering_input.directions.contiguous()
        b, r = o.shape[:2]
        s = self.cfg.rendering.num_nerf_samples
        a3 = 3 * self.decoder.kernel_action_dim  # 3A for the Jacobian decoders, 3 for flow_mlp (the flow itself)
        dev = o.device
        f32 = dict(dtype=torch.float32, device=dev)
        io = self.frame_io if not (dump_jacobian or dump_perception) else None
        if io is not None:   # caller-owned pixel buffers + in-kernel frame reductions (parallel.ShardedFrameStep)
            if tuple(io["rgb"].shape) != (b, r, 3):
                raise ValueError(f"Model.frame_io was set up for {tuple(io['rgb'].shape[:2])} rays, this call renders {(b, r)}")
            outs: Dict[str, torch.Tensor] = {k: io[k] for k in ("rgb", "depth", "flow", "frame_partials")}
            outs.update({k: io[k] for k in ("trgt_rgb", "trgt_flow") if io.get(k) is not None})
            clip_depth = False   # deferred: the global bounds exist only after the step's collective
        else:
            outs = {"rgb": torch.empty(b, r, 3, **f32), "depth": torch.empty(b, r, 1, **f32),
                    "step_minmax": torch.empty(b, r, 2, **f32), "flow": torch.empty(b, r, 2, **f32)}
        if want_lists or want_vis or want_samples:
            outs["weights"] = torch.empty(b, r, s, **f32)
        if want_vis:
            outs["pos"] = torch.empty(b, r, 3, **f32)
            outs["pos_warped"] = torch.empty(b, r, 3, **f32)
            # flow_mlp: the composited features are the flow head's 640 hidden channels (model.py:381-390 on
            # action_decoder_flow.py:168-176), weighted below from a point query -- not an output of the render kernel
            hidden_features = hasattr(self.decoder, "composited_features")
            if not (dump_jacobian or dump_perception or hidden_features):   # (never requested from a training forward: _vis_at_bins)
                outs["action_features"] = torch.empty(b, r, a3, **f32)
        if want_samples:
            outs["density"] = torch.empty(b, r, s, 1, **f32)
            outs["jacobian"] = torch.empty(b, r, s, a3, **f32)
        if want_sample_outputs:
            outs["color"] = torch.empty(b, r, s, 3, **f32)
            outs["sample_flow"] = torch.empty(b, r, s, 3, **f32)
        if dump_jacobian:  # inputs of the Jacobian head's backward pass (training.py)
            pts = b * r * s
            outs["weights"] = outs.get("weights", torch.empty(b, r, s, **f32))
            outs["pos_warped"] = outs.get("pos_warped", torch.empty(b, r, 3, **f32))
            outs["jac_forward_precision"] = self.decoder.j_precision    # (what training.py's "auto" settings follow)
            if self.decoder.JACOBIAN_KIND == hip.JACOBIAN_MLP:
                from . import training as _tr   # (fp16 under the opt-in 16-bit training storage, training.set_storage_precision)
                outs["jac_act"] = torch.empty(11, pts, 128, dtype=_tr.activation_dump_dtype(self.decoder.j_precision), device=dev)
                outs["jac_mask"] = torch.empty(11, pts, 4, dtype=torch.int32, device=dev)   # ReLU masks: what the backward chain reads
            elif self.decoder.JACOBIAN_KIND == hip.JACOBIAN_TRANSFORMER:
                # the transformer head's residual stream in front of its three layers and behind the last one: what its fused
                # backward chain (njf_transformer_backward, training.transformer_head_backward) recomputes each layer from
                outs["jac_act"] = torch.empty(4, pts, 64, **f32)
            outs["jac_pe"] = torch.empty(pts, 64, **f32)
            outs["foot_idx"] = torch.empty(pts, 4, dtype=torch.int32, device=dev)
            outs["foot_w"] = torch.empty(pts, 4, **f32)
        if dump_perception:  # inputs of the density net's and the colour head's backward pass (training.py)
            pts = b * r * s
            outs["proposal_dumps"] = proposal_dumps
            outs["weights"] = outs.get("weights", torch.empty(b, r, s, **f32))
            outs["density"] = outs.get("density", torch.empty(b, r, s, 1, **f32))
            outs["color"] = torch.empty(b, r, s, 3, **f32)
            from . import training as _tr
            outs["den_forward_precision"] = self.decoder.precision
            outs["den_act"] = torch.empty(11, pts, 128, dtype=_tr.activation_dump_dtype(self.decoder.precision), device=dev)
            outs["den_mask"] = torch.empty(11, pts, 4, dtype=torch.int32, device=dev)
            outs["jac_pe"] = torch.empty(pts, 64, **f32)
            outs["foot_idx"] = torch.empty(pts, 4, dtype=torch.int32, device=dev)
            outs["foot_w"] = torch.empty(pts, 4, **f32)
            outs["col_in"] = torch.empty(pts, 32, **f32)
            outs["col_act"] = torch.empty(2, pts, 64, **f32)
        w, bd, bc, bj = self.decoder.packed()
        if joint is None:
            fmap, goff = hip.make_feature_map(self.decoder.hoisted_map(features, enc.action)), 0
        else:
            fmap, goff = joint_fmap, joint[2]
        cams = _cameras(enc, True, rendering_input.z_near, rendering_input.z_far,
                        (self._inverse("trgt", camera_input.trgt_extrinsics) if trgt_w2c is None else trgt_w2c).contiguous(),
                        camera_input.trgt_intrinsics.contiguous(), action=self.decoder.kernel_action(enc.action))
        hip.render_forward(o, d, cams, fmap, goff + self.decoder.GOFF_DENSITY, goff + self.decoder.GOFF_JACOBIAN, w, bd, bc, bj, bins, s,
                           {k: v for k, v in outs.items() if torch.is_tensor(v)},
                           jacobian_kind=self.decoder.JACOBIAN_KIND, precision=self.decoder.precision,
                           jacobian_precision=self.decoder.j_precision)
        if clip_depth:  # tensor-global clip of model.py:277
            outs["depth"] = self.depth_clip(outs["depth"], outs["step_minmax"])
        if want_vis and not (dump_jacobian or dump_perception) and hasattr(self.decoder, "composited_features"):
            outs["action_features"] = self.decoder.composited_features(
                ray_bundle.samples_from_bins(bins).get_positions(), outs["weights"], enc)
        return outs, bins, weights_list, bins_list, ray_bundle

    def forward(self, camera_input: CameraInput, rendering_input: RenderingInput, robot_input: RobotInput,
                compute_vis_features: bool = False) -> ModelOutput:
        """model.py:316-396.  With gradients enabled and trainable parameters (reference action mode: only the
        Jacobian head -- the flow head of ``flow_mlp`` --, model_wrapper.py:75-85) ``optical_flow`` carries an autograd graph (training.py); every other
        output, and every call under ``torch.no_grad()`` / with frozen parameters, is a plain inference pass."""
        self._maybe_check_range(camera_input, rendering_input, robot_input)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from . import training
            reduced = [m.precision for m in (self.decoder, *self.proposal_networks) if m.precision in hip.REDUCED_PRECISIONS]
            if reduced:
                raise RuntimeError(f"the {reduced[0]!r} MFMA precision is an inference mode (no training forward exists for it: "
                                   "activations rounded to fp16 are not the inputs a backward pass may use); call "
                                   "model.set_precision('f16f6') / ('f32') for training, or run under torch.no_grad()")
            if training.is_action_mode(self):
                return self._forward_action_grad(camera_input, rendering_input, robot_input, compute_vis_features)
            return self._forward_perception_grad(camera_input, rendering_input, robot_input, compute_vis_features)
        return self._forward_inference(camera_input, rendering_input, robot_input, compute_vis_features)

    def _forward_perception_grad(self, camera_input, rendering_input, robot_input, compute_vis_features) -> ModelOutput:
        """Any trainable set other than action mode (reference perception mode trains everything and reads rgb, depth
        and the per-level weights, model_wrapper.py:117-146).  rgb / depth / weights_list carry autograd graphs back to
        the encoder, the density and colour heads and the proposal nets; ``optical_flow`` is a value only -- the
        perception losses never read it -- and refuses back-propagation."""
        from . import training
        from .encoder import FeaturePyramid
        # the encoder WITH its autograd graph; when it offers its un-concatenated latents the forward pass hoists from them
        # and the encoder tail (up-sampling + concatenation) exists only in FieldFunction's backward pass, as two HIP kernels
        fp = getattr(self.encoder, "forward_pyramid", None)
        features = fp(camera_input.input_image) if fp is not None else self.encoder.forward(camera_input.input_image)
        levels = list(features.levels) if isinstance(features, FeaturePyramid) else [features]
        detached = FeaturePyramid([lv.detach() for lv in levels]) if len(levels) > 1 else levels[0].detach()
        box = {}

        def run():
            with torch.no_grad():
                outs, bins, wl, bl, rb = self._fused_render(camera_input, rendering_input, robot_input, detached,
                                                            want_lists=True, want_vis=False,
                                                            want_samples=False, dump_perception=True, clip_depth=False,
                                                            final_bins=self.training_final_bins)
            box.update(outs=outs, bins=bins, bins_list=bl, weights_list=wl, ray_bundle=rb)
            return outs

        params = training.perception_params(self)
        n_prop = 0 if self.training_final_bins is not None else len(self.proposal_networks)   # (given bins: no proposal level runs)
        sigma, color, *sigma_prop = training.FieldFunction.apply(run, len(levels), n_prop, *levels, *params)
        outs, ray_bundle = box["outs"], box["ray_bundle"]
        # compositing of model.py:257-279 on the differentiable fields: the values are the fused kernels' own (weights, rgb,
        # un-clipped depth), the backward pass is one njf_composite_backward launch per level (training.CompositeFunction)
        smp = ray_bundle.samples_from_bins(box["bins"])
        steps = (smp.starts + smp.ends) / 2
        weights, rgb, depth = training.CompositeFunction.apply(smp.deltas, steps, sigma, color, outs)
        # render_depth's tensor-global clip (model.py:277) routed through self.depth_clip, so that a ray shard clips with the
        # all-reduced bounds exactly like the inference path (parallel.enable_ray_sharding)
        depth = self.depth_clip(depth, outs["step_minmax"])
        anchor = next(p for p in self.parameters() if p.requires_grad)
        flow = training.RefuseBackward.apply(outs["flow"], anchor, training.PERCEPTION_MESSAGE)
        out = ModelOutput(ModelStandardOutput(rgb=rgb, depth=depth, optical_flow=flow), None, None)
        samples_list = [ray_bundle.samples_from_bins(bn) for bn in box["bins_list"]]
        weights_list = []
        for lvl_samples, dump, sg, w_fwd in zip(samples_list, outs["proposal_dumps"], sigma_prop, box["weights_list"]):
            # the proposal nets receive gradient only on the steps the reference's `updated` schedule allows
            weights_list.append(training.CompositeFunction.apply(lvl_samples.deltas, None, sg, None, {"weights": w_fwd})
                                if dump["updated"] else w_fwd.detach().reshape(sg.shape))
        if self.training:
            out.training_output = ModelTrainingOutput(weights_list=weights_list + [weights],
                                                      ray_samples_list=samples_list + [smp])
        if compute_vis_features:
            out.vis_output = self._vis_at_bins(camera_input, rendering_input, robot_input, detached, box["bins"], smp)
        return out

    def _vis_at_bins(self, camera_input, rendering_input, robot_input, features, bins, smp) -> "ModelVisOutput":
        """ModelVisOutput of a TRAINING forward (model.py:381-394): an inference render of the same samples (``bins``).  No loss
        reads these fields, so the training forwards do not request them from the dump instantiations of the render kernel --
        which lets those be built with or without the 16 action-feature accumulators (-DNJF_TRAIN_NO_AF; the A/B of round 5,
        profiles/r05_spills_ab.txt, kept the accumulators: the leaner build was not faster) -- and the visualisation pays for
        itself on the steps that ask for it (the reference's validation / logging steps)."""
        # ... and a SEPARATE no-grad render: `weights` / positions of a training step's vis_output come from this pass, not from the
        # differentiated one (same samples, same weights).  It must not see a caller's frame_io (parallel.ShardedFrameStep): that
        # would overwrite caller-owned pixel buffers and skip the depth clip
        prev_io, self.frame_io = self.frame_io, None
        try:
            with torch.no_grad():
                vis, *_ = self._fused_render(camera_input, rendering_input, robot_input, features, want_lists=False, want_vis=True,
                                             want_samples=False, final_bins=bins)
        finally:
            self.frame_io = prev_io
        return ModelVisOutput(action_features=vis["action_features"], steps=((smp.starts + smp.ends) / 2).squeeze(-1),
                              weights=vis["weights"], ray_positions=vis["pos"], ray_positions_warped=vis["pos_warped"])

    @staticmethod
    def _weights_from_density(deltas: torch.Tensor, densities: torch.Tensor) -> torch.Tensor:
        """RaySamples.get_weights (ray_samplers.py:77-101) in differentiable torch ops (training graphs only; the
        inference path composites inside the render kernel)."""
        ds = torch.where(deltas > 0, deltas * densities, torch.zeros_like(densities))
        acc = torch.cumsum(ds[..., :-1, :], dim=-2)
        acc = torch.cat([torch.zeros_like(ds[..., :1, :]), acc], dim=-2)   # (ds, not acc: a single sample has no prefix)
        return -torch.expm1(-ds) * torch.exp(-acc)   # (alpha without the cancellation of 1 - exp(-ds): csrc alpha_of)

    def _forward_action_grad(self, camera_input, rendering_input, robot_input, compute_vis_features) -> ModelOutput:
        from . import training
        names, jparams = training.action_params(self)
        # the encoder is frozen in action mode: the forward pass hoists straight from its latents; the concatenated
        # 512-channel map is only formed in the backward pass (lin_z weight gradients), by njf_upsample_concat
        features = self._encode_for_render(camera_input.input_image)
        box = {}

        def run():
            with torch.no_grad():
                outs, bins, wl, bl, rb = self._fused_render(camera_input, rendering_input, robot_input, features,
                                                            want_lists=self.training, want_vis=False, want_samples=False,
                                                            dump_jacobian=True, final_bins=self.training_final_bins)
            box.update(outs=outs, bins=bins, weights_list=wl, bins_list=bl, ray_bundle=rb)
            return outs

        project = lambda x: self._project(x, camera_input.trgt_extrinsics, camera_input.trgt_intrinsics)
        flow = training.ActionFlowFunction.apply(run, project, robot_input.robot_action.detach(), features, names,
                                                 training.action_kind(self), *jparams)
        outs = box["outs"]
        out = ModelOutput(ModelStandardOutput(rgb=outs["rgb"], depth=outs["depth"], optical_flow=flow), None, None)
        self._attach_optional_outputs(out, outs, box["bins"], box["weights_list"], box["bins_list"], box["ray_bundle"], False)
        if compute_vis_features:
            out.vis_output = self._vis_at_bins(camera_input, rendering_input, robot_input, features, box["bins"],
                                               box["ray_bundle"].samples_from_bins(box["bins"]))
        return out

    def _attach_optional_outputs(self, out, outs, bins, weights_list, bins_list, ray_bundle, compute_vis_features):
        if self.training:
            weights_list.append(outs["weights"][..., None])
            bins_list.append(bins)
            out.training_output = ModelTrainingOutput(
                weights_list=weights_list, ray_samples_list=[ray_bundle.samples_from_bins(bn) for bn in bins_list])
        if compute_vis_features:
            smp = ray_bundle.samples_from_bins(bins)
            out.vis_output = ModelVisOutput(
                action_features=outs["action_features"], steps=((smp.starts + smp.ends) / 2).squeeze(-1),
                weights=outs["weights"], ray_positions=outs["pos"], ray_positions_warped=outs["pos_warped"])

    @torch.no_grad()
    def _encode_for_render(self, image: torch.Tensor):
        """Encoder output for the fused inference path: the un-concatenated latents when the encoder offers them
        (the hoisted map is then produced without materialising the 512-channel feature map)."""
        fp = getattr(self.encoder, "forward_pyramid", None)
        return fp(image) if fp is not None else self.encoder.forward(image)

    def _forward_inference(self, camera_input: CameraInput, rendering_input: RenderingInput, robot_input: RobotInput,
                           compute_vis_features: bool = False) -> ModelOutput:
        features = self._encode_for_render(camera_input.input_image)
        outs, bins, weights_list, bins_list, ray_bundle = self._fused_render(
            camera_input, rendering_input, robot_input, features, want_lists=self.training, want_vis=compute_vis_features,
            want_samples=False)
        out = ModelOutput(ModelStandardOutput(rgb=outs["rgb"], depth=outs["depth"], optical_flow=outs["flow"]), None, None)
        self._attach_optional_outputs(out, outs, bins, weights_list, bins_list, ray_bundle, compute_vis_features)
        return out

    # ---- inference helpers (model.py:398-525) -----------------------------------------------
    @torch.no_grad()
    def compute_pixel_encoding(self, camera_input: CameraInput, rendering_input: RenderingInput,
                               robot_input: RobotInput) -> PixelEncoding:
        return PixelEncoding(features=self.encoder.forward(camera_input.input_image),
                             extrinsics=camera_input.ctxt_extrinsics, intrinsics=camera_input.ctxt_intrinsics,
                             action=robot_input.robot_action)

    @torch.no_grad()
    def compute_density(self, world_space_xyz: torch.Tensor, pixel_encoding: PixelEncoding) -> Tuple[DensityHeadOutput, dict]:
        """model.py:416-456.  ``world_space_xyz`` [B,N,3] (the reference annotates [B,R,S,3] but feeds the
        flat form to get_pixel_aligned_features)."""
        head = self.decoder.compute_density(world_space_xyz, pixel_encoding)
        extras = {}
        if "jacobian" in self.cfg.action_decoder.name:
            extras["jacobian_head_output"] = self.decoder.compute_jacobian_at(world_space_xyz, pixel_encoding)
        return head, extras

    @torch.no_grad()
    def encode_image(self, camera_input: CameraInput, rendering_input: RenderingInput,
                     robot_input: RobotInput) -> ModelInferenceEncoding:
        """model.py:458-495: proposal sampling + per-sample density/Jacobian/weights, cached for inverse dynamics."""
        if "jacobian" not in self.cfg.action_decoder.name:
            raise NotImplementedError("encode_image caches per-sample Jacobians (model.py:458-495); flow_mlp has none -- "
                                      "the reference's own flow_mlp.encode_image is unusable as well")
        self._maybe_check_range(camera_input, rendering_input, robot_input)
        features = self._encode_for_render(camera_input.input_image)
        outs, bins, _, _, ray_bundle = self._fused_render(camera_input, rendering_input, robot_input, features,
                                                         want_lists=False, want_vis=False, want_samples=True)
        positions = ray_bundle.samples_from_bins(bins).get_positions()
        return ModelInferenceEncoding(density=outs["density"], action_features=outs["jacobian"],
                                      weights=outs["weights"][..., None], ray_samples_positions=positions)

    def infer_optical_flow(self, model_inference_encoding: ModelInferenceEncoding, camera_input: CameraInput,
                           robot_input: RobotInput) -> torch.Tensor:
        """model.py:497-525.  Differentiable w.r.t. ``robot_input.robot_action`` (the inverse-dynamics loop of
        notebooks/real_world/2_inverse_dynamics.ipynb optimises the action through this call), hence torch ops."""
        assert "jacobian" in self.cfg.action_decoder.name
        enc = model_inference_encoding
        b, r, s = enc.action_features.shape[:3]
        a = robot_input.robot_action.shape[-1]
        jac = enc.action_features.reshape(b, r, s, a, -1)
        scene_flow = torch.einsum("brsad,ba->brsd", jac, robot_input.robot_action)
        flow, _, _ = self.render_optical_flow(enc.weights, enc.ray_samples_positions, scene_flow[..., :3],
                                              camera_input.trgt_extrinsics, camera_input.trgt_intrinsics)
        return flow

    @torch.no_grad()
    def patch_render(self, camera_input: CameraInput, rendering_input: RenderingInput, robot_input: RobotInput,
                     patch_size: int = 2048, render_height: int = 480, render_width: int = 640,
                     verbose: bool = False) -> RenderingOutput:
        """model.py:527-628.  The fused path never materialises per-point feature tensors, so the whole frame
        is rendered in ONE pass (``patch_size`` is accepted for signature compatibility and ignored); the encoder
        runs once instead of once per patch.  ``depth_rgb`` / ``flow_rgb`` come from visualization.py (restated
        nerfstudio / torchvision helpers, evaluated on the device)."""
        self._maybe_check_range(camera_input, rendering_input, robot_input)
        was_training = self.training
        self.eval()
        try:
            features = self._encode_for_render(camera_input.input_image)
            outs, bins, _, _, ray_bundle = self._fused_render(camera_input, rendering_input, robot_input, features,
                                                             want_lists=False, want_vis=True, want_samples=False)
        finally:
            self.train(was_training)
        from .visualization import apply_depth_colormap, flow_to_image
        smp = ray_bundle.samples_from_bins(bins)
        img = lambda t: t.reshape(t.shape[0], render_height, render_width, -1)
        depth_raw, flow_raw = img(outs["depth"]), img(outs["flow"])
        return RenderingOutput(
            rgb=img(outs["rgb"]), depth_raw=depth_raw, depth_rgb=apply_depth_colormap(depth_raw), flow_raw=flow_raw,
            flow_rgb=flow_to_image(flow_raw.permute(0, 3, 1, 2)).permute(0, 2, 3, 1),
            ray_positions=img(outs["pos"]), ray_positions_warped=img(outs["pos_warped"]),
            action_features=img(outs["action_features"]), steps=img(((smp.starts + smp.ends) / 2).squeeze(-1)),
            weights=img(outs["weights"]))
