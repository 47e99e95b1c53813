"""Per-point decoders -- host-side mirror of ``models/decoder/`` (registry, ABC, I/O dataclasses,
state-dict-compatible parameter trees) whose forward passes run in the fused HIP kernels.

Reference: ``models/decoder/__init__.py:11-44`` (registries), ``action_decoder.py:11-64``,
``action_decoder_jacobian.py:86-337``, ``action_decoder_flow.py:64-290``, ``density_decoder.py:23-71``,
``model_components/resnet_fc.py:82-154``.
"""

from __future__ import annotations

import os

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import hip
from .encoder import FeaturePyramid
from .config import (ActionDecoderCfg, ActionDecoderFlowMlpCfg, ActionDecoderJacobianMlpCfg,
                     ActionDecoderJacobianTransformerCfg, DensityDecoderCfg, DensityDecoderMlpCfg, MlpCfg)


# --------------------------------------------------------------------------------------
# I/O records (action_decoder.py:11-30, action_decoder_jacobian.py:64-75)
# --------------------------------------------------------------------------------------
@dataclass
class PixelEncoding:
    features: torch.Tensor    # [B,C,Hf,Wf] encoder output
    extrinsics: torch.Tensor  # [B,4,4] context cam2world
    intrinsics: torch.Tensor  # [B,3,3] normalised context intrinsics
    action: torch.Tensor      # [B,A]
    extrinsics_inv: Optional[torch.Tensor] = None  # not in the reference: world->camera, if the caller already has it


@dataclass
class DecoderOutput:
    density: torch.Tensor          # [B,R,S,1]
    color: torch.Tensor            # [B,R,S,3]
    flow: torch.Tensor             # [B,R,S,3]
    action_features: torch.Tensor  # [B,R,S,3A]


@dataclass
class DecoderFeatureOnlyOutput:
    density: torch.Tensor
    action_features: torch.Tensor


@dataclass
class DensityHeadOutput:
    density: torch.Tensor
    density_features: torch.Tensor
    xyz_features: Optional[torch.Tensor]            # never materialised by the fused path (None)
    pixel_aligned_features: Optional[torch.Tensor]  # never materialised by the fused path (None)


# --------------------------------------------------------------------------------------
# ResnetFC parameter tree (names/shapes/init of resnet_fc.py:27-128)
# --------------------------------------------------------------------------------------
class ResnetBlockFC(nn.Module):
    def __init__(self, size: int):
        super().__init__()
        self.fc_0 = nn.Linear(size, size)
        self.fc_1 = nn.Linear(size, size)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)


class ResnetFC(nn.Module):
    """Parameter container with the reference's names.  Its arithmetic (resnet_fc.py:130-154) lives in
    the fused kernels, reached through the owning decoder; the geometry it supports is the shipped one
    (``MlpCfg(n_blocks=5, d_hidden=128, combine_layer=3, beta=0)``, d_in=63, d_latent=512)."""

    def __init__(self, resnet_cfg: MlpCfg, d_in: int, d_latent: int, d_out: int, extra_latent: int = 0):
        """``extra_latent``: latent columns beyond the 512 encoder channels that are CONSTANT per batch element (the
        robot action of ``flow_mlp``, action_decoder_flow.py:100-106); the owning decoder folds their ``lin_z``
        contribution into the hoisted map's bias, the kernels only ever see 512 feature channels."""
        super().__init__()
        if (resnet_cfg.n_blocks, resnet_cfg.d_hidden, resnet_cfg.combine_layer) != (5, 128, 3) or resnet_cfg.beta > 0:
            raise ValueError("fused ResnetFC supports MlpCfg(n_blocks=5, d_hidden=128, combine_layer=3, beta=0) only")
        if d_in != 63 or d_latent != 512 or not (1 <= d_out <= 32) or extra_latent < 0:
            raise ValueError("fused ResnetFC supports d_in=63, d_latent=512 (+ per-batch constants), 1 <= d_out <= 32")
        self.resnet_cfg, self.d_latent, self.d_out = resnet_cfg, d_latent + extra_latent, d_out
        h = resnet_cfg.d_hidden
        self.lin_in = nn.Linear(d_in, h)
        self.lin_out = nn.Linear(h, d_out)
        self.blocks = nn.ModuleList([ResnetBlockFC(h) for _ in range(resnet_cfg.n_blocks)])
        self.lin_z = nn.ModuleList([nn.Linear(d_latent + extra_latent, h) for _ in range(resnet_cfg.combine_layer)])
        for lin in [self.lin_in, self.lin_out, *self.lin_z]:
            nn.init.constant_(lin.bias, 0.0)
            nn.init.kaiming_normal_(lin.weight, a=0, mode="fan_in")

    def forward(self, z, x):  # pragma: no cover - documented non-goal
        raise NotImplementedError(
            "ResnetFC runs only fused with feature sampling (DensityDecoderMlp.get_density / "
            "ActionDecoderJacobian.forward); the stand-alone (z, x) form is not exported by the HIP path")


def initialize_jacobian_weights(m: nn.Module) -> None:
    """action_decoder_jacobian.py:78-83."""
    if type(m) == nn.Linear:
        nn.init.normal_(m.weight, mean=0.0, std=1e-4)
        if m.bias is not None:
            nn.init.normal_(m.bias, mean=0.0, std=1e-4)


def _version(module: nn.Module) -> Tuple:
    return ((getattr(module, "precision", None), getattr(module, "jacobian_precision", None))
            + tuple((p.data_ptr(), p._version) for p in module.parameters()))


class _HoistCache:
    """One hoisted feature map per (feature tensor object, its version counter, weight version).  The cache keeps a
    reference to the feature tensor: a freed tensor's address can be recycled by the allocator for the next image,
    so an address-based key would silently serve a stale map."""

    def __init__(self):
        self.key = None
        self.features = None
        self.gmap = None

    def get(self, features: torch.Tensor, wversion, wz: torch.Tensor, bz: torch.Tensor, precision: str) -> torch.Tensor:
        key = (features._version, tuple(features.shape), wversion, precision)
        if features is not self.features or key != self.key:
            b, _, hf, wf = features.shape
            gmap = torch.empty(b, hf, wf, wz.shape[1], dtype=hip.map_dtype(precision), device=features.device)
            if isinstance(features, FeaturePyramid):
                hip.project_pyramid(features.levels, wz, bz, gmap, precision=precision)
            else:
                hip.project_features(features.contiguous(), wz, bz, gmap, precision=precision)
            self.key, self.features, self.gmap = key, features, gmap
        return self.gmap


def _map_of(net, features, action=None):
    """(hoisted map, first channel of `net`'s block in it) for a point query: the frame's JOINT map when the owning Model has
    projected one for exactly these features and these weights (``net._joint_view``, written by Model._joint_hoist), else the
    network's own map.  The record is plain data -- feature tensor (identity + version + shape), the network's packed-weights
    version, the map, the channel base -- so a deep copy of a model never reads another model's map."""
    view = getattr(net, "_joint_view", None)
    if view is not None:
        f, fver, fshape, wver, gmap, base = view
        net.packed()
        if f is features and fver == features._version and fshape == tuple(features.shape) and wver == net._packed_version:
            return gmap, base
    # (flow_mlp adds a per-image action bias to its block: it never shares a map -- Model._joint_hoist returns None for it)
    return (net.hoisted_map(features) if action is None else net.hoisted_map(features, action)), 0


_ZEROS: Dict[tuple, torch.Tensor] = {}


def _cameras(enc: PixelEncoding, with_action: bool, z_near=None, z_far=None, trgt_w2c=None, trgt_k=None, action_dim=None,
             action=None):
    """``action``: what the kernel contracts the head's output with (default: the robot action itself)."""
    b = enc.extrinsics.shape[0]
    dev = enc.extrinsics.device
    zeros = None
    if z_near is None or z_far is None:
        zeros = _ZEROS.get((b, hip.device_key(dev)))
        if zeros is None:   # a constant: filled once per (batch, device), not per call
            zeros = torch.zeros(b, dtype=torch.float32, device=dev)
            if not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):   # (a fill captured into a graph has not run yet)
                _ZEROS[(b, hip.device_key(dev))] = zeros
    if action is None:
        action = enc.action
    w2c = hip.inverse(enc.extrinsics) if enc.extrinsics_inv is None else enc.extrinsics_inv
    return hip.make_cameras(w2c.contiguous(), enc.intrinsics.contiguous(),
                            zeros if z_near is None else z_near.contiguous(),
                            zeros if z_far is None else z_far.contiguous(), trgt_w2c, trgt_k,
                            action.contiguous() if with_action else None, action_dim)


# --------------------------------------------------------------------------------------
# proposal density decoder (density_decoder.py:23-71)
# --------------------------------------------------------------------------------------
class DensityDecoderMlp(nn.Module):
    def __init__(self, cfg: DensityDecoderMlpCfg, encoder_dim: int):
        super().__init__()
        if cfg.num_frequencies != 10:
            raise ValueError("fused path supports num_frequencies=10 (63-d positional encoding)")
        self.cfg = cfg
        self.density_head = ResnetFC(cfg.mlp, d_in=63, d_latent=encoder_dim, d_out=1)
        self.precision = hip.proposal_precision_for(hip.DEFAULT_PRECISION)  # MFMA precision of the fused MLP
        self._packed_version = None
        self._hoist = _HoistCache()

    # ---- packed state ----------------------------------------------------------------
    def packed(self):
        v = _version(self)
        if v != self._packed_version:
            dev = self.density_head.lin_in.weight.device
            f32 = dict(dtype=torch.float32, device=dev)
            self._w = torch.empty(hip.RESNET_W_FLOATS, **f32)
            self._b = torch.empty(hip.RESNET_B_FLOATS, **f32)
            self._wz = torch.empty(512, hip.ZDIM, **f32)
            self._bz = torch.empty(hip.ZDIM, **f32)
            params = {k: p for k, p in self.named_parameters()}
            hip.pack_resnetfc(params, "density_head.", self._w, self._b, self._wz, 0, self._bz, precision=self.precision)
            self._packed_version = v
        return self._w, self._b

    def hoisted_map(self, features: torch.Tensor) -> torch.Tensor:
        self.packed()
        return self._hoist.get(features, self._packed_version, self._wz, self._bz, self.precision)

    # ---- reference API -----------------------------------------------------------------
    @torch.no_grad()
    def get_density(self, world_space_xyz: torch.Tensor, pixel_encoding: PixelEncoding) -> torch.Tensor:
        """[B,R,S,3] world points -> density [B,R,S,1]  (density_decoder.py:45-71)."""
        b, r, s = world_space_xyz.shape[:3]
        w, bias = self.packed()
        gmap, base = _map_of(self, pixel_encoding.features)
        out = torch.empty(b, r, s, 1, dtype=torch.float32, device=world_space_xyz.device)
        xyz = world_space_xyz.reshape(b, r * s, 3).contiguous()
        hip.points_forward(xyz, None, _cameras(pixel_encoding, False), hip.make_feature_map(gmap), base, base, 0, w, bias, density=out,
                           precision=self.precision)
        return out


# --------------------------------------------------------------------------------------
# action decoders (action_decoder.py:33-64, action_decoder_jacobian.py:86-337)
# --------------------------------------------------------------------------------------
class ActionDecoder(nn.Module, ABC):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    @abstractmethod
    def forward(self, world_space_xyz, world_space_dir, pixel_encoding: PixelEncoding) -> DecoderOutput: ...

    @abstractmethod
    def encode_image(self, world_space_xyz, pixel_encoding: PixelEncoding) -> DecoderFeatureOnlyOutput: ...

    @abstractmethod
    def freeze_non_action_parameters(self) -> int: ...


class ActionDecoderJacobian(ActionDecoder):
    """Shared machinery of the two Jacobian decoders (action_decoder_jacobian.py:86-258): density_head (d_out=16)
    + colour head + a Jacobian head, evaluated by one fused kernel.  Subclasses define the Jacobian head's
    parameters, its packed form and the width of its hoisted feature channels."""

    spatial_dim: int = 3
    # kind / packed sizes / hoisted channels of the subclass's OWN ("regular") Jacobian head; the properties below give those
    # of the head that is active (switch_mode): the arm head is a ResnetFC in both decoders (action_decoder_jacobian.py:
    # 306-313, 400-407), i.e. kind JACOBIAN_MLP with a 384-channel hoisted block
    REGULAR = (hip.JACOBIAN_NONE, 0, 0, 0)
    ARM = (hip.JACOBIAN_MLP, hip.RESNET_W_FLOATS, hip.RESNET_B_FLOATS, hip.ZDIM)
    GOFF_DENSITY, GOFF_JACOBIAN = 0, hip.ZDIM

    @property
    def _active(self):
        return self.ARM if self.mode == "arm" else self.REGULAR

    @property
    def JACOBIAN_KIND(self) -> int:
        return self._active[0]

    @property
    def J_W_FLOATS(self) -> int:
        return self._active[1]

    @property
    def J_B_FLOATS(self) -> int:
        return self._active[2]

    @property
    def J_HOIST(self) -> int:
        return self._active[3]

    @property
    def active_head_prefix(self) -> str:
        """Parameter-name prefix (relative to the decoder) of the ResnetFC Jacobian head in use; '' when the active head is
        not a single ResnetFC (the transformer decoder in regular mode)."""
        if self.mode == "arm":
            return "jacobian_head_arm."
        return "jacobian_head." if isinstance(getattr(self, "jacobian_head", None), ResnetFC) else ""

    def _init_common(self, cfg, action_dim: int, encoder_dim: int, max_action: int, max_arm_action: int = hip.MAX_ACTION_DIM):
        n_freq = cfg.num_frequncies if hasattr(cfg, "num_frequncies") else cfg.num_frequencies
        if n_freq != 10 or cfg.geometry_feature_dim != 15:
            raise ValueError("fused path supports num_frequencies=10 and geometry_feature_dim=15")
        arm = getattr(cfg, "use_arm_model", False)
        if arm and not (cfg.arm_action_dim is not None and 1 <= cfg.arm_action_dim <= max_arm_action):
            raise ValueError(f"use_arm_model needs arm_action_dim in [1, {max_arm_action}]")
        self.arm_action_dim = cfg.arm_action_dim if arm else None
        if not (1 <= action_dim <= max_action):
            raise ValueError(f"action_dim must be in [1, {max_action}] for {cfg.name}")
        self.action_dim = action_dim
        self.density_head = ResnetFC(cfg.mlp, d_in=63, d_latent=encoder_dim, d_out=cfg.geometry_feature_dim + 1)
        self.mode = "regular"
        self.precision = hip.DEFAULT_PRECISION  # "f32" | "f16x2" | "f16f6": density + colour networks
        self.jacobian_precision = None          # the Jacobian head's, when it differs (Model.set_precision)
        self._packed_version = None
        self._hoist = _HoistCache()
        # Reference checkpoints carry one key this module has no use for: SHEncoding(implementation="tcnn")
        # (action_decoder_jacobian.py:284) wraps a tinycudann Encoding, which always registers a `params` Parameter -- empty
        # for spherical harmonics.  It is accepted (and dropped) on load, so wrapper.load_state_dict(ckpt["state_dict"])
        # works with strict=True as well as with the reference's strict=False (train.py:58).
        self._register_load_state_dict_pre_hook(self._drop_tcnn_placeholder)

    @staticmethod
    def _drop_tcnn_placeholder(state_dict, prefix, *unused):
        key = prefix + "directional_encoding.tcnn_encoding.params"
        if key in state_dict and state_dict[key].numel() == 0:
            del state_dict[key]

    def _make_color_head(self, cfg):
        return nn.Sequential(nn.Linear(cfg.geometry_feature_dim + 16, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(),
                             nn.Linear(64, 3), nn.Sigmoid())

    def _make_arm_head(self, cfg, encoder_dim: int) -> None:
        """action_decoder_jacobian.py:306-313 / 400-407: the second Jacobian head, a ResnetFC(d_out = 3 * arm_action_dim),
        registered as ``jacobian_head_arm`` (state-dict keys ``decoder.jacobian_head_arm.*``) when cfg.use_arm_model."""
        if self.arm_action_dim is not None:
            self.jacobian_head_arm = ResnetFC(cfg.mlp, d_in=63, d_latent=encoder_dim, d_out=self.spatial_dim * self.arm_action_dim)
            self.jacobian_head_arm.apply(initialize_jacobian_weights)

    def switch_mode(self, mode: str):
        """action_decoder_jacobian.py:89-90.  ``"arm"`` routes compute_jacobian through ``jacobian_head_arm`` (:330-331,
        :438-446); everything downstream is unchanged -- in particular compute_flow contracts the head's output with the
        robot action viewed as (action_dim, 3) (:134-140), which is why the reference only runs in arm mode when
        arm_action_dim == action_dim.  That condition is checked HERE (the reference fails later, inside einops)."""
        if mode not in ("regular", "arm"):
            raise ValueError(f"mode must be 'regular' or 'arm', not {mode!r}")
        if mode == "arm":
            if self.arm_action_dim is None:
                raise AttributeError("switch_mode('arm'): this decoder was built without use_arm_model (no arm head)")
            if self.arm_action_dim != self.action_dim:
                raise ValueError(f"switch_mode('arm'): arm_action_dim = {self.arm_action_dim} but compute_flow contracts the head's "
                                 f"output with the {self.action_dim}-dimensional robot action (action_decoder_jacobian.py:134-140)")
        self.mode = mode

    # ---- packed state ----------------------------------------------------------------
    def _pack_regular_jacobian(self, params, w_j, b_j, wz, bz):  # pragma: no cover - abstract
        raise NotImplementedError

    def _jacobian_sub_version(self, params):
        heads = ("density_head.", "color_head.")
        return tuple((p.data_ptr(), p._version) for k, p in params.items()
                     if not k.startswith(heads) and k.startswith("jacobian_head_arm.") == (self.mode == "arm"))

    def _pack_jacobian(self, params, w_j, b_j, wz, bz):
        if self.mode == "arm":
            hip.pack_resnetfc(params, "jacobian_head_arm.", w_j, b_j, wz, hip.ZDIM, bz, precision=self.j_precision)
        else:
            self._pack_regular_jacobian(params, w_j, b_j, wz, bz)

    def packed(self):
        """Packed weights, rebuilt PER SUB-NETWORK: an action-mode optimiser step only touches the Jacobian head
        (freeze_non_action_parameters), so the density and colour packs of the previous step stay valid."""
        v = (self.mode,) + _version(self)
        if v != self._packed_version:
            dev = self.density_head.lin_in.weight.device
            n = hip.RESNET_W_FLOATS
            # (a mode switch changes which head the Jacobian block holds -- and, for the transformer decoder, its size)
            fresh = self._packed_version is None or self._w.device != dev or self._packed_version[0] != self.mode
            if fresh:
                f32 = dict(dtype=torch.float32, device=dev)
                self._w = torch.zeros(n + hip.COLOR_W_FLOATS + self.J_W_FLOATS, **f32)
                self._bd = torch.empty(hip.RESNET_B_FLOATS, **f32)
                self._bc = torch.empty(hip.COLOR_B_FLOATS, **f32)
                self._bj = torch.zeros(self.J_B_FLOATS, **f32)
                self._wz = torch.zeros(512, hip.ZDIM + self.J_HOIST, **f32)
                self._bz = torch.zeros(hip.ZDIM + self.J_HOIST, **f32)
                self._sub_versions = {}
            params = {k: p for k, p in self.named_parameters()}
            heads = ("density_head.", "color_head.")
            subs = {"density": (self.precision,) + tuple((p.data_ptr(), p._version) for k, p in params.items() if k.startswith(heads[0])),
                    "color": (self.precision,) + tuple((p.data_ptr(), p._version) for k, p in params.items() if k.startswith(heads[1])),
                    # the ACTIVE head's parameters only: training one head must not re-pack on the other's (frozen) values
                    "jacobian": (self.j_precision, self.mode) + self._jacobian_sub_version(params)}
            if subs["density"] != self._sub_versions.get("density"):
                hip.pack_resnetfc(params, heads[0], self._w[:n], self._bd, self._wz, 0, self._bz, precision=self.precision)
            if subs["color"] != self._sub_versions.get("color"):
                hip.pack_color_head(params, heads[1], self._w[n:n + hip.COLOR_W_FLOATS], self._bc, precision=self.precision)
            if subs["jacobian"] != self._sub_versions.get("jacobian"):
                self._pack_jacobian(params, self._w[n + hip.COLOR_W_FLOATS:], self._bj, self._wz, self._bz)
            self._sub_versions = subs
            self._packed_version = v
        return self._w, self._bd, self._bc, self._bj

    def hoisted_map(self, features: torch.Tensor, action: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Per-image hoisted map.  ``action`` ([B,A]) only matters to decoders whose head takes the action as a latent
        input (``flow_mlp``); the Jacobian decoders ignore it."""
        self.packed()
        return self._hoist.get(features, self._packed_version, self._wz, self._bz, self.precision)

    @property
    def j_precision(self) -> str:
        return self.precision if self.jacobian_precision is None else self.jacobian_precision

    # what the kernel contracts the head's 3*A' outputs with: the robot action for the Jacobian heads
    @property
    def kernel_action_dim(self) -> int:
        return self.action_dim

    def kernel_action(self, action: torch.Tensor) -> torch.Tensor:
        return action

    def _points(self, xyz_flat, dirs_flat, enc: PixelEncoding, with_jacobian: bool, want: Dict[str, bool]):
        b, n = xyz_flat.shape[:2]
        dev = xyz_flat.device
        w, bd, bc, bj = self.packed()
        gmap, base = _map_of(self, enc.features, enc.action)
        fmap = hip.make_feature_map(gmap)
        f32 = dict(dtype=torch.float32, device=dev)
        out = {"density": torch.empty(b, n, 1, **f32)}
        if want.get("color"):
            out["color"] = torch.empty(b, n, 3, **f32)
        if want.get("geo"):
            out["geo"] = torch.empty(b, n, 15, **f32)
        if with_jacobian:
            out["jacobian"] = torch.empty(b, n, 3 * self.kernel_action_dim, **f32)
            if want.get("flow"):
                out["flow"] = torch.empty(b, n, 3, **f32)
            if want.get("features"):   # ResnetFC.forward(compute_features=True): residual stream after each block, block-major
                out["features"] = torch.empty(5, b * n, 128, **f32)
        hip.points_forward(xyz_flat.contiguous(), None if dirs_flat is None else dirs_flat.contiguous(),
                           _cameras(enc, with_jacobian and want.get("flow", False), action_dim=self.kernel_action_dim,
                                    action=self.kernel_action(enc.action)), fmap,
                           base + self.GOFF_DENSITY, base + self.GOFF_JACOBIAN, 1, w, bd, bc, bj,
                           jacobian_kind=self.JACOBIAN_KIND if with_jacobian else hip.JACOBIAN_NONE,
                           precision=self.precision, jacobian_precision=self.j_precision, **out)
        return out

    # ---- reference API -----------------------------------------------------------------
    @torch.no_grad()
    def compute_density(self, world_space_xyz: torch.Tensor, pixel_encoding: PixelEncoding) -> DensityHeadOutput:
        """[B,N,3] -> DensityHeadOutput (action_decoder_jacobian.py:92-119)."""
        o = self._points(world_space_xyz, None, pixel_encoding, False, {"geo": True})
        return DensityHeadOutput(o["density"], o["geo"], None, None)

    @torch.no_grad()
    def forward(self, world_space_xyz, world_space_dir, pixel_encoding: PixelEncoding) -> DecoderOutput:
        """action_decoder_jacobian.py:147-215."""
        b, r, s = world_space_xyz.shape[:3]
        o = self._points(world_space_xyz.reshape(b, r * s, 3), world_space_dir.reshape(b, r * s, 3), pixel_encoding, True,
                         {"color": True, "flow": True})
        sh = lambda t: t.reshape(b, r, s, -1)
        return DecoderOutput(sh(o["density"]), sh(o["color"]), sh(o["flow"]), sh(o["jacobian"]))

    @torch.no_grad()
    def encode_image(self, world_space_xyz, pixel_encoding: PixelEncoding) -> DecoderFeatureOnlyOutput:
        """action_decoder_jacobian.py:217-249."""
        b, r, s = world_space_xyz.shape[:3]
        o = self._points(world_space_xyz.reshape(b, r * s, 3), None, pixel_encoding, True, {})
        return DecoderFeatureOnlyOutput(o["density"].reshape(b, r, s, 1), o["jacobian"].reshape(b, r, s, -1))

    @torch.no_grad()
    def compute_jacobian_at(self, world_space_xyz, pixel_encoding: PixelEncoding) -> torch.Tensor:
        """Jacobian head on [B,N,3] points (what Model.compute_density puts in extras, model.py:447-454)."""
        return self._points(world_space_xyz, None, pixel_encoding, True, {})["jacobian"]

    def freeze_non_action_parameters(self) -> int:
        """action_decoder_jacobian.py:251-258."""
        count = 0
        for name, p in self.named_parameters():
            if self.action_param_glob_pattern not in name:
                p.requires_grad = False
                count += 1
        return count


class ActionDecoderJacobianMLP(ActionDecoderJacobian):
    """action_decoder_jacobian.py:261-337: Jacobian head = ResnetFC(d_out=3A)."""

    action_param_glob_pattern = "jacobian_head"
    REGULAR = (hip.JACOBIAN_MLP, hip.RESNET_W_FLOATS, hip.RESNET_B_FLOATS, hip.ZDIM)

    def __init__(self, cfg: ActionDecoderJacobianMlpCfg, action_dim: int, encoder_dim: int):
        super().__init__(cfg)
        self._init_common(cfg, action_dim, encoder_dim, hip.MAX_ACTION_DIM)
        self.jacobian_head = ResnetFC(cfg.mlp, d_in=63, d_latent=encoder_dim, d_out=self.spatial_dim * action_dim)
        self.jacobian_head.apply(initialize_jacobian_weights)
        self._make_arm_head(cfg, encoder_dim)      # (registration order of the reference: head, arm head, colour head)
        self.color_head = self._make_color_head(cfg)

    def _pack_regular_jacobian(self, params, w_j, b_j, wz, bz):
        hip.pack_resnetfc(params, "jacobian_head.", w_j, b_j, wz, hip.ZDIM, bz, precision=self.j_precision)


def initialize_flow_weights(m: nn.Module) -> None:
    """action_decoder_flow.py:56-61 (same rule as the Jacobian heads)."""
    initialize_jacobian_weights(m)


class ActionDecoderFlowMlp(ActionDecoderJacobian):
    """action_decoder_flow.py:64-290 (``flow_mlp``, the reference's direct-flow ablation): density + colour heads as in
    the Jacobian decoders, and ``flow_head = ResnetFC(d_latent = encoder_dim + action_dim, d_out = 3)`` evaluated on
    ``cat[pixel_aligned_features, action]`` -- the scene flow itself, not a Jacobian.

    On the fused path the action never reaches the per-point kernel as an input: it is constant per batch element and
    enters only through ``lin_z``, so ``lin_z(cat[f, a]) = W_f f + (W_a a + b)`` and the bracket is a per-image bias of
    the hoisted map (``hoisted_map`` adds it to the flow head's 384 channels).  The kernel then runs the flow head as a
    "Jacobian head" with ONE action channel contracted with the constant 1.0, which returns its three outputs unchanged.

    ``use_arm_model`` registers ``flow_head_arm = ResnetFC(d_latent = encoder_dim + arm_action_dim)`` (:109-116);
    ``switch_mode("arm")`` routes compute_flow through it (:163-166).  compute_flow concatenates the robot action itself
    (:168-172), so the reference only runs in arm mode when arm_action_dim == action_dim -- checked in ``switch_mode``.
    The active flow head trains in the reference's action mode (``action_param_glob_pattern = "flow_head"``; training.py).

    ``encode_image`` mirrors the reference's as it is (a ``map`` object yielding the density, :246-279; ``Model.encode_image``
    cannot consume it on either side).  ``DecoderOutput.action_features`` / ``ModelVisOutput.action_features`` are the flow
    head's 640 hidden features as in the reference (:168-176, model.py:381-390): the point-query kernel stores the head's residual
    stream after each block (ABI v18), ``composited_features`` weights them along the rays.  (Not in the plain-fp16 mode, which has
    no such instantiation.)
    """

    action_param_glob_pattern = "flow_head"
    REGULAR = (hip.JACOBIAN_MLP, hip.RESNET_W_FLOATS, hip.RESNET_B_FLOATS, hip.ZDIM)

    def __init__(self, cfg: ActionDecoderFlowMlpCfg, action_dim: int, encoder_dim: int):
        super().__init__(cfg)
        self._init_common(cfg, action_dim, encoder_dim, 1 << 16, 1 << 16)  # the kernel sees one channel, any A works
        self.flow_head = ResnetFC(cfg.mlp, d_in=63, d_latent=encoder_dim, d_out=self.spatial_dim, extra_latent=action_dim)
        self.flow_head.apply(initialize_flow_weights)
        if self.arm_action_dim is not None:   # action_decoder_flow.py:109-116 (registered between flow_head and color_head)
            self.flow_head_arm = ResnetFC(cfg.mlp, d_in=63, d_latent=encoder_dim, d_out=self.spatial_dim,
                                          extra_latent=self.arm_action_dim)
            self.flow_head_arm.apply(initialize_flow_weights)
        self.color_head = self._make_color_head(cfg)
        self._ones = None

    @property
    def active_head_prefix(self) -> str:
        return "flow_head_arm." if self.mode == "arm" else "flow_head."

    @property
    def active_head(self) -> ResnetFC:
        return self.flow_head_arm if self.mode == "arm" else self.flow_head

    @property
    def kernel_action_dim(self) -> int:
        return 1

    def kernel_action(self, action: torch.Tensor) -> torch.Tensor:
        if self._ones is None or self._ones.shape[0] != action.shape[0] or self._ones.device != action.device:
            self._ones = torch.ones(action.shape[0], 1, dtype=torch.float32, device=action.device)
        return self._ones

    def _jacobian_sub_version(self, params):
        """The ACTIVE flow head's parameters (both heads' names start with "flow_head")."""
        prefix = self.active_head_prefix
        return tuple((p.data_ptr(), p._version) for k, p in params.items() if k.startswith(prefix))

    def _pack_jacobian(self, params, w_j, b_j, wz, bz):
        prefix = self.active_head_prefix
        enc_dim = self.active_head.d_latent - self.action_dim
        sliced = dict(params)
        for i in range(3):  # the kernels hoist the 512 feature columns; the action columns become a bias (hoisted_map)
            sliced[f"{prefix}lin_z.{i}.weight"] = params[f"{prefix}lin_z.{i}.weight"][:, :enc_dim].contiguous()
        hip.pack_resnetfc(sliced, prefix, w_j, b_j, wz, hip.ZDIM, bz, precision=self.j_precision)

    @torch.no_grad()
    def hoisted_map(self, features: torch.Tensor, action: Optional[torch.Tensor] = None) -> torch.Tensor:
        base = super().hoisted_map(features)
        if action is None:
            raise ValueError("flow_mlp: the hoisted map depends on the robot action (PixelEncoding.action)")
        head = self.active_head
        enc_dim = head.d_latent - self.action_dim
        w_a = torch.stack([lin.weight[:, enc_dim:] for lin in head.lin_z])                   # [3,128,A]
        delta = torch.einsum("lfa,ba->blf", w_a, action.to(w_a.dtype))                       # [B,3,128] logical order
        pos = hip.hoisted_channel_order(128, delta.device, self.j_precision)                 # njf_hoisted_channel
        permuted = torch.empty_like(delta)
        permuted[:, :, pos] = delta
        gmap = base.clone()
        gmap[..., self.GOFF_JACOBIAN:self.GOFF_JACOBIAN + hip.ZDIM] += permuted.reshape(-1, 1, 1, hip.ZDIM)
        return gmap

    @torch.no_grad()
    def forward(self, world_space_xyz, world_space_dir, pixel_encoding: PixelEncoding) -> DecoderOutput:
        """action_decoder_flow.py:185-244.  ``action_features`` are the flow head's 5 x 128 hidden features
        (``ResnetFC.forward(compute_features=True)``, resnet_fc.py:141-151: the residual stream after each block, concatenated),
        stored by the point-query kernel next to the flow."""
        b, r, s = world_space_xyz.shape[:3]
        with_features = self.j_precision not in hip.REDUCED_PRECISIONS   # (plain fp16: no such instantiation; None as before)
        o = self._points(world_space_xyz.reshape(b, r * s, 3), world_space_dir.reshape(b, r * s, 3), pixel_encoding, True,
                         {"color": True, "flow": True, "features": with_features})
        sh = lambda t: t.reshape(b, r, s, -1)
        return DecoderOutput(sh(o["density"]), sh(o["color"]), sh(o["flow"]),
                             sh(self.hidden_features(o["features"])) if with_features else None)

    @staticmethod
    def hidden_features(block_major: torch.Tensor) -> torch.Tensor:
        """[5, P, 128] (the kernel's block-major dump) -> [P, 640] = torch.cat(features, dim=-1) of resnet_fc.py:150-151."""
        return block_major.permute(1, 0, 2).reshape(block_major.shape[1], -1)

    @torch.no_grad()
    def composited_features(self, positions: torch.Tensor, weights: torch.Tensor, pixel_encoding: PixelEncoding,
                            max_points: int = 1 << 20) -> torch.Tensor:
        """Model.render_action_features (model.py:281-286) of this decoder's hidden features: sum_s w_s f_s -> [B,R,640] from sample
        positions [B,R,S,3] and weights [B,R,S].  Ray chunks of at most ``max_points`` points (the per-sample features are 2.5 KB
        per point: a 480 x 640 x 256 patch_render frame would be 200 GB in one piece)."""
        if self.j_precision in hip.REDUCED_PRECISIONS:
            raise RuntimeError("flow_mlp: the hidden action features are not produced in the plain-fp16 mode (no feature-storing "
                               "instantiation of its point-query kernel); call model.set_precision('f16f6') / ('f32') for them")
        b, r, s = positions.shape[:3]
        out = torch.empty(b, r, 640, dtype=torch.float32, device=positions.device)
        step = max(1, max_points // max(1, b * s))
        for lo in range(0, r, step):
            hi = min(r, lo + step)
            n = (hi - lo) * s
            o = self._points(positions[:, lo:hi].reshape(b, n, 3), None, pixel_encoding, True, {"features": True})
            feats = self.hidden_features(o["features"]).reshape(b, hi - lo, s, 640)
            out[:, lo:hi] = torch.einsum("brs,brsc->brc", weights[:, lo:hi], feats)
        return out

    @torch.no_grad()
    def encode_image(self, world_space_xyz, pixel_encoding: PixelEncoding):
        """action_decoder_flow.py:246-279, as it is: the reference computes the density head only (its feature line is commented
        out) and returns the ``map`` object of its reshape -- an iterator that yields ONE tensor, density [B,R,S,1] -- not a
        DecoderFeatureOnlyOutput; ``Model.encode_image`` cannot consume it there (model.py:487-492) and refuses here."""
        b, r, s = world_space_xyz.shape[:3]
        density = self._points(world_space_xyz.reshape(b, r * s, 3), None, pixel_encoding, False, {})["density"]
        return map(lambda x: x.reshape(b, r, s, 1), (density,))

    def compute_jacobian_at(self, world_space_xyz, pixel_encoding: PixelEncoding):
        raise NotImplementedError("flow_mlp predicts the scene flow directly; it has no Jacobian")


# ---- parameter tree of model_components/transformer.py (names only; arithmetic is folded + fused) ----
class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class _CrossAttention(nn.Module):
    def __init__(self, dim, heads, dim_head, kv_dim):
        super().__init__()
        inner = heads * dim_head
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(kv_dim, inner * 2, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0))


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), nn.Linear(hidden, dim), nn.Dropout(0.0))


class _TransformerParams(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, kv_dim):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([_PreNorm(dim, _CrossAttention(dim, heads, dim_head, kv_dim)), _PreNorm(dim, _FeedForward(dim, mlp_dim))])
            for _ in range(depth)])


class ActionDecoderJacobianTransformer(ActionDecoderJacobian):
    """action_decoder_jacobian.py:340-446: query MLP -> 3 x (cross-attention to A learned tokens, GELU FF) -> Linear.

    The fused kernel evaluates an algebraically folded form (exact algebra, fp64 folding, fp32 evaluation):
      * ``jacobian_query_mlp`` splits into a positional-encoding part (in-kernel) and a feature part that is
        hoisted into the per-image map exactly like ``lin_z``;
      * keys/values depend only on the learned ``jacobian_index_embedding`` (they are not layer-normed and shared
        by all points, transformer.py:63-70), so ``softmax(to_q(LN x) K^T / 8) V -> to_out`` becomes
        ``Nov . softmax_8(Mqk . norm(x) + bqk) + bo`` with two 64x64 matrices per layer (rows of ``Mqk`` ordered
        head*8 + key; LayerNorm affine and the 1/sqrt(64) scale folded in);
      * the feed-forward's LayerNorm affine is folded into its first Linear.
    That is ~55 kMAC/point instead of the reference's 284 kMAC/point.
    """

    action_param_glob_pattern = "jacobian"
    REGULAR = (hip.JACOBIAN_TRANSFORMER, hip.TRANSFORMER_W_FLOATS, hip.TRANSFORMER_B_FLOATS, hip.QDIM)

    def __init__(self, cfg: ActionDecoderJacobianTransformerCfg, action_dim: int, encoder_dim: int):
        super().__init__(cfg)
        t = cfg.transformer
        if (t.attn_feat_dim, t.num_attn_heads, t.attn_depth, t.attn_mlp_dim) != (64, 8, 3, 64):
            raise ValueError("fused transformer head supports attn_feat_dim=64, num_attn_heads=8, attn_depth=3, "
                             "attn_mlp_dim=64 (configurations/model/model_allegro.yaml:34-39)")
        self._init_common(cfg, action_dim, encoder_dim, 8)  # 8 key slots per head
        self.jacobian_index_embedding = nn.Parameter(torch.randn(1, action_dim, t.attn_feat_dim), requires_grad=True)
        self.jacobian_query_mlp = nn.Linear(encoder_dim + 63, t.attn_feat_dim)
        self.jacobian_attn_decoder = _TransformerParams(t.attn_feat_dim, t.attn_depth, t.num_attn_heads, t.attn_head_dim,
                                                        t.attn_mlp_dim, t.attn_feat_dim)
        self.jacobian_head = nn.Linear(t.attn_feat_dim, self.spatial_dim * action_dim)
        self.jacobian_head.apply(initialize_jacobian_weights)
        self._make_arm_head(cfg, encoder_dim)
        self.color_head = self._make_color_head(cfg)

    @torch.no_grad()
    def _pack_regular_jacobian(self, params, w_j, b_j, wz, bz):
        t = self.cfg.transformer
        heads = t.num_attn_heads
        f32 = lambda x: x.to(torch.float32).contiguous()
        half = lambda i: w_j[4096 * i: 4096 * (i + 1)]
        qw = self.jacobian_query_mlp.weight  # [64, 63 + 512], input = cat[xyz_features, pixel_aligned_features] (:421-427)
        hip.pack_linear(qw[:, :63].contiguous(), self.jacobian_query_mlp.bias, 1, half(0), precision=self.j_precision)
        # hoisted query channels use the same in-block order as lin_z (csrc: njf_hoist_position, MB=2)
        pos = hip.hoisted_channel_order(64, qw.device, self.j_precision)
        wz[:, hip.ZDIM + pos] = qw[:, 63:].t()
        bz[hip.ZDIM:] = 0.0
        # the fold itself (keys / values / LayerNorm affines -> two 64 x 64 matrices per attention layer, the affine of the second
        # LayerNorm into W1): training.folded_transformer -- ONE set of batched float64 ops for the three layers, the same graph the
        # head's backward pass differentiates (an action-mode run re-packs after every optimiser step; fp32 results bit-identical
        # to a loop over the layers)
        from . import training
        names = [n for n in params if n.startswith("jacobian") and not n.startswith("jacobian_head_arm.")]
        _, folded = training.transformer_fold(names, [params[n] for n in names], heads=heads)   # (kept for this step's backward pass)
        mats, biases = f32(folded["mats"].detach()), f32(folded["biases"].detach())     # [3,4,64,64] = (Mqk', Nov, W1', W2), [3,4,64] = (bqk', bo, b1', b2)
        for l in range(mats.shape[0]):
            bl = b_j[256 * l: 256 * (l + 1)]
            for i in range(4):
                hip.pack_linear(mats[l, i], biases[l, i], 0, half(1 + 4 * l + i), bl[64 * i:64 * (i + 1)], precision=self.j_precision)
        hip.pack_linear(self.jacobian_head.weight, self.jacobian_head.bias, 0, half(13)[:2048], b_j[768:800], precision=self.j_precision)


# --------------------------------------------------------------------------------------
# registries (models/decoder/__init__.py:11-44)
# --------------------------------------------------------------------------------------
DENSITY_DECODERS = {"density_mlp": DensityDecoderMlp}
ACTION_DECODERS = {"jacobian_mlp": ActionDecoderJacobianMLP, "jacobian_transformer": ActionDecoderJacobianTransformer,
                   "flow_mlp": ActionDecoderFlowMlp}


def get_density_decoder(cfg: DensityDecoderCfg, encoder_dim: int) -> DensityDecoderMlp:
    return DENSITY_DECODERS[cfg.name](cfg=cfg, encoder_dim=encoder_dim)


def get_action_decoder(cfg: ActionDecoderCfg, action_dim: int, encoder_dim: int) -> ActionDecoder:
    if cfg.name not in ACTION_DECODERS:
        raise KeyError(f"action decoder {cfg.name!r} is not available in the HIP path; known: {sorted(ACTION_DECODERS)}")
    return ACTION_DECODERS[cfg.name](cfg=cfg, action_dim=action_dim, encoder_dim=encoder_dim)
