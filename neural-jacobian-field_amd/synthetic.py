"""Deterministic synthetic weights and inputs for the rendering hot path.

The reference ships neither checkpoints nor datasets (``.MISSING_LARGE_BLOBS``), and its own
initialisation makes most of the network invisible to a parity test (``fc_1.weight = 0`` at
``model_components/resnet_fc.py:56``; Jacobian head std 1e-4 at
``models/decoder/action_decoder_jacobian.py:78-83``).  SURVEY.md section 8(d) therefore defines
seeded, non-degenerate synthetic weights and batches; this module is their single source, used
by ``bench.py``, ``__graft_entry__.smoke()``, the tests and the golden-vector generator.

Names and shapes follow the reference ``Model.state_dict()`` (pinned by
``tests/golden/state_dict_manifest.json``).
"""

from __future__ import annotations

import math
import zlib
from typing import Dict, Optional, Tuple

import torch

Shape = Tuple[int, ...]


# --------------------------------------------------------------------------------------
# state-dict manifests
# --------------------------------------------------------------------------------------
def _linear(out: Dict[str, Shape], name: str, d_out: int, d_in: int, bias: bool = True) -> None:
    out[name + ".weight"] = (d_out, d_in)
    if bias:
        out[name + ".bias"] = (d_out,)


def resnet_fc_shapes(prefix: str, d_in: int, d_latent: int, d_out: int, d_hidden: int = 128,
                     n_blocks: int = 5, combine_layer: int = 3) -> Dict[str, Shape]:
    """Parameter names of ``ResnetFC`` (model_components/resnet_fc.py:100-128)."""
    out: Dict[str, Shape] = {}
    _linear(out, prefix + "lin_in", d_hidden, d_in)
    _linear(out, prefix + "lin_out", d_out, d_hidden)
    for i in range(n_blocks):
        _linear(out, f"{prefix}blocks.{i}.fc_0", d_hidden, d_hidden)
        _linear(out, f"{prefix}blocks.{i}.fc_1", d_hidden, d_hidden)
    for i in range(min(combine_layer, n_blocks)):
        _linear(out, f"{prefix}lin_z.{i}", d_hidden, d_latent)
    return out


def color_head_shapes(prefix: str, geo_dim: int = 15) -> Dict[str, Shape]:
    out: Dict[str, Shape] = {}
    _linear(out, prefix + "color_head.0", 64, geo_dim + 16)
    _linear(out, prefix + "color_head.2", 64, 64)
    _linear(out, prefix + "color_head.4", 3, 64)
    return out


def decoder_shapes(kind: str, action_dim: int, encoder_dim: int = 512, pe_dim: int = 63, geo_dim: int = 15,
                   attn_feat_dim: int = 64, heads: int = 8, head_dim: int = 64, depth: int = 3,
                   mlp_dim: int = 64, prefix: str = "decoder.", arm_action_dim: Optional[int] = None) -> Dict[str, Shape]:
    """``ActionDecoderJacobianMLP`` / ``ActionDecoderJacobianTransformer`` parameters
    (models/decoder/action_decoder_jacobian.py:261-322, :340-416); ``arm_action_dim``: the second head of ``use_arm_model``
    (:306-313, :400-407)."""
    out = resnet_fc_shapes(prefix + "density_head.", pe_dim, encoder_dim, geo_dim + 1)
    if arm_action_dim is not None and kind == "flow_mlp":   # action_decoder_flow.py:109-116
        out.update(resnet_fc_shapes(prefix + "flow_head_arm.", pe_dim, encoder_dim + arm_action_dim, 3))
    elif arm_action_dim is not None:
        out.update(resnet_fc_shapes(prefix + "jacobian_head_arm.", pe_dim, encoder_dim, 3 * arm_action_dim))
    if kind == "jacobian_mlp":
        out.update(resnet_fc_shapes(prefix + "jacobian_head.", pe_dim, encoder_dim, 3 * action_dim))
    elif kind == "jacobian_transformer":
        inner = heads * head_dim
        out[prefix + "jacobian_index_embedding"] = (1, action_dim, attn_feat_dim)
        _linear(out, prefix + "jacobian_query_mlp", attn_feat_dim, encoder_dim + pe_dim)
        for l in range(depth):
            base = f"{prefix}jacobian_attn_decoder.layers.{l}."
            out[base + "0.norm.weight"] = (attn_feat_dim,)
            out[base + "0.norm.bias"] = (attn_feat_dim,)
            _linear(out, base + "0.fn.to_q", inner, attn_feat_dim, bias=False)
            _linear(out, base + "0.fn.to_kv", 2 * inner, attn_feat_dim, bias=False)
            _linear(out, base + "0.fn.to_out.0", attn_feat_dim, inner)
            out[base + "1.norm.weight"] = (attn_feat_dim,)
            out[base + "1.norm.bias"] = (attn_feat_dim,)
            _linear(out, base + "1.fn.net.0", mlp_dim, attn_feat_dim)
            _linear(out, base + "1.fn.net.3", attn_feat_dim, mlp_dim)
        _linear(out, prefix + "jacobian_head", 3 * action_dim, attn_feat_dim)
    elif kind == "flow_mlp":  # models/decoder/action_decoder_flow.py:100-106: the action joins the latent
        out.update(resnet_fc_shapes(prefix + "flow_head.", pe_dim, encoder_dim + action_dim, 3))
    else:
        raise ValueError(f"unknown action decoder {kind!r}")
    out.update(color_head_shapes(prefix, geo_dim))
    return out


def proposal_shapes(index: int = 0, encoder_dim: int = 512, pe_dim: int = 63) -> Dict[str, Shape]:
    """``DensityDecoderMlp`` (models/decoder/density_decoder.py:23-43) inside ``Model.proposal_networks``."""
    return resnet_fc_shapes(f"proposal_networks.{index}.density_head.", pe_dim, encoder_dim, 1)


def resnet34_shapes(prefix: str = "encoder.model.") -> Dict[str, Shape]:
    """torchvision resnet34 state dict (BasicBlock [3,4,6,3]); layer4/fc exist but are unused
    by the reference forward (models/encoder/encoder_resnet.py:66-75)."""
    out: Dict[str, Shape] = {}

    def bn(name: str, c: int) -> None:
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)
        out[name + ".running_mean"] = (c,)
        out[name + ".running_var"] = (c,)
        out[name + ".num_batches_tracked"] = ()

    out[prefix + "conv1.weight"] = (64, 3, 7, 7)
    bn(prefix + "bn1", 64)
    cin = 64
    for li, (planes, blocks) in enumerate(zip([64, 128, 256, 512], [3, 4, 6, 3]), start=1):
        for bi in range(blocks):
            base = f"{prefix}layer{li}.{bi}."
            out[base + "conv1.weight"] = (planes, cin, 3, 3)
            bn(base + "bn1", planes)
            out[base + "conv2.weight"] = (planes, planes, 3, 3)
            bn(base + "bn2", planes)
            if bi == 0 and li > 1:
                out[base + "downsample.0.weight"] = (planes, cin, 1, 1)
                bn(base + "downsample.1", planes)
            cin = planes
    out[prefix + "fc.weight"] = (1000, 512)
    out[prefix + "fc.bias"] = (1000,)
    return out


def model_shapes(decoder_kind: str = "jacobian_mlp", action_dim: int = 8, num_proposal_networks: int = 1,
                 with_encoder: bool = True, arm_action_dim: Optional[int] = None) -> Dict[str, Shape]:
    """Full ``Model.state_dict()`` manifest (models/model.py:147-199)."""
    out: Dict[str, Shape] = {}
    if with_encoder:
        out.update(resnet34_shapes())
    out.update(decoder_shapes(decoder_kind, action_dim, arm_action_dim=arm_action_dim))
    for i in range(num_proposal_networks):
        out.update(proposal_shapes(i))
    return out


# --------------------------------------------------------------------------------------
# seeded non-degenerate values
# --------------------------------------------------------------------------------------
def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
    return g


def seeded_tensor(name: str, shape: Shape, seed: int = 0, linear_std: float = 0.05) -> torch.Tensor:
    """One parameter/buffer.  Every value depends only on (seed, name, shape)."""
    g = _gen(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_var":
        return 1.0 + 0.1 * torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.05 * torch.randn(shape, generator=g)
    if len(shape) == 4:  # conv: fan-in scaled so activations keep O(1) scale through the trunk
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
    if ".norm." in name or ".bn" in name or "downsample.1" in name:
        if leaf == "weight":
            return 1.0 + 0.05 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)
    if name.endswith("jacobian_index_embedding"):
        return torch.randn(shape, generator=g)
    t = torch.randn(shape, generator=g) * linear_std
    # Density pre-activation: bias the last output of the density nets so sigma*delta spans
    # roughly [0.05, 5] and the hierarchical sampler sees a non-trivial PDF.
    if name.endswith("density_head.lin_out.weight"):
        t[-1] *= 3.0
    if name.endswith("density_head.lin_out.bias"):
        t[-1] += 2.0
    # flow_mlp predicts the scene flow itself: keep it at the ~0.1 m scale J.a has for the Jacobian decoders, so the
    # warped points stay in front of the target camera and the projected flow is well conditioned
    if name.endswith(("flow_head.lin_out.weight", "flow_head.lin_out.bias", "flow_head_arm.lin_out.weight",
                      "flow_head_arm.lin_out.bias")):
        t *= 0.1
    return t


def seeded_state_dict(shapes: Dict[str, Shape], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(k, v, seed) for k, v in shapes.items()}


# --------------------------------------------------------------------------------------
# synthetic batches
# --------------------------------------------------------------------------------------
def yaw_pose(deg: float = 10.0, x_offset: float = 0.1, batch: int = 1) -> torch.Tensor:
    a = math.radians(deg)
    m = torch.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
    m[0, 3] = x_offset
    return m[None].repeat(batch, 1, 1).contiguous()


def synthetic_cameras(batch: int = 1) -> Dict[str, torch.Tensor]:
    k = torch.tensor([[0.8, 0.0, 0.5], [0.0, 0.8, 0.5], [0.0, 0.0, 1.0]])
    return {
        "ctxt_c2w": torch.eye(4)[None].repeat(batch, 1, 1).contiguous(),
        "ctxt_k_norm": k[None].repeat(batch, 1, 1).contiguous(),
        "trgt_c2w": yaw_pose(batch=batch),
        "trgt_k_norm": k[None].repeat(batch, 1, 1).contiguous(),
        "z_near": torch.full((batch,), 0.5),
        "z_far": torch.full((batch,), 10.0),
    }


def synthetic_features(batch: int, height: int, width: int, channels: int = 512, seed: int = 1) -> torch.Tensor:
    """Feature map F ~ N(0,1) [B,C,H/2,W/2] standing in for the encoder output."""
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randn((batch, channels, height // 2, width // 2), generator=g)


def synthetic_action(batch: int, action_dim: int, seed: int = 2) -> torch.Tensor:
    g = torch.Generator()
    g.manual_seed(seed)
    return 0.1 * torch.randn((batch, action_dim), generator=g)


def synthetic_training_batch(batch: int, height: int, width: int, rays: int, action_dim: int, seed: int, device) -> Dict[str, torch.Tensor]:
    """One training batch of the reference's shape (``configurations/config.yaml:18-20``: 7 scenes x 256 random pixels,
    the SAME pixel set for every scene, ``models/model_wrapper.py:438-444``) on ``device``: context images, cameras, the
    rays of the selected target pixels (the HIP ray-generation kernel), a command per scene and random targets.  Every
    value depends only on the arguments -- rank r of a data-parallel job passes ``seed = r`` and so owns its scenes."""
    from . import geometry

    g = torch.Generator().manual_seed(seed * 7919 + 17)
    cams = {k: v.to(device) for k, v in synthetic_cameras(batch).items()}
    sel = torch.randperm(height * width, generator=g)[:rays]
    coords, _ = geometry.get_pixel_coordinates(height, width)
    xy = coords.reshape(1, -1, 2)[:, sel].repeat(batch, 1, 1).contiguous().to(device)
    origins, directions, _ = geometry.get_world_rays_with_z(xy, cams["trgt_k_norm"], cams["trgt_c2w"])
    return {
        "image": torch.rand(batch, 3, height, width, generator=g).to(device),
        "ctxt_c2w": cams["ctxt_c2w"], "ctxt_k_norm": cams["ctxt_k_norm"], "trgt_c2w": cams["trgt_c2w"],
        "trgt_k_pix": geometry.denormalize_intrinsics(cams["trgt_k_norm"], width, height),
        "z_near": cams["z_near"], "z_far": cams["z_far"], "origins": origins, "directions": directions,
        "action": synthetic_action(batch, action_dim, seed + 2).to(device),
        "target_rgb": torch.rand(batch, rays, 3, generator=g).to(device),
        "target_depth": (torch.rand(batch, rays, 1, generator=g) + 0.5).to(device),
        "target_flow": torch.randn(batch, rays, 2, generator=g).to(device),
    }


def general_pose(seed: int, batch: int, scale: float = 0.15) -> torch.Tensor:
    """A seeded non-identity camera pose [B,4,4] (rotation exp(scale * skew), translation 0.1 * N(0,1))."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(batch, 3, 3, generator=g)
    m = torch.eye(4)[None].repeat(batch, 1, 1)
    m[:, :3, :3] = torch.matrix_exp(scale * (a - a.transpose(1, 2)))
    m[:, :3, 3] = 0.1 * torch.randn(batch, 3, generator=g)
    return m.contiguous()


def synthetic_case(batch: int, height: int, width: int, rays: Optional[int], action_dim: int, seed: int = 0, device="cpu",
                   identity_context: bool = True, decoder: str = "jacobian_mlp") -> Dict[str, object]:
    """One seeded frame for tools and benches that need INPUTS only (the same recipe as the parity suite's cases,
    built from the package alone so such tools travel without oracle/): weights (no encoder), feature map, cameras, the rays of
    ``rays`` randomly chosen pixels (None = the whole frame) from the HIP ray-generation kernel, pixel intrinsics, command."""
    from . import geometry

    dev = torch.device(device)
    cams = synthetic_cameras(batch)
    if not identity_context:
        cams["ctxt_c2w"] = general_pose(seed + 5, batch)
    cams = {k: v.to(dev) for k, v in cams.items()}
    coords, _ = geometry.get_pixel_coordinates(height, width)
    xy = coords.reshape(1, -1, 2)
    if rays is not None and rays < height * width:
        xy = xy[:, torch.randperm(height * width, generator=torch.Generator().manual_seed(seed + 3))[:rays]]
    xy = xy.repeat(batch, 1, 1).contiguous().to(dev)
    origins, directions, _ = geometry.get_world_rays_with_z(xy, cams["trgt_k_norm"], cams["trgt_c2w"])
    return {"params": {k: v.to(dev) for k, v in seeded_state_dict(model_shapes(decoder, action_dim, with_encoder=False), seed).items()},
            "feats": synthetic_features(batch, height, width, seed=seed + 1).to(dev), "cams": cams,
            "origins": origins.contiguous(), "directions": directions.contiguous(),
            "k_pix": geometry.denormalize_intrinsics(cams["trgt_k_norm"], width, height),
            "action": synthetic_action(batch, action_dim, seed + 2).to(dev)}
