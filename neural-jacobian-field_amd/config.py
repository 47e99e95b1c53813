"""Typed configuration mirroring the reference dataclasses (field names and defaults identical),
without hydra/omegaconf/dacite.

Reference: ``model_components/resnet_fc.py:11-18`` (MlpCfg), ``models/decoder/action_decoder_jacobian.py:33-61``, ``action_decoder_flow.py:32-39``,
``models/decoder/density_decoder.py:16-20``, ``models/encoder/encoder_resnet.py:15-21``,
``models/model.py:35-54`` (RenderingCfg, ModelCfg); YAML defaults in ``configurations/model/*.yaml``.
"""

from __future__ import annotations

from dataclasses import dataclass, field, fields, is_dataclass
from typing import Any, Dict, Literal, Optional, Tuple, Union, get_origin, get_type_hints


@dataclass
class MlpCfg:
    n_blocks: int = 5
    d_hidden: int = 128
    combine_layer: int = 3
    combine_type: Literal["mean"] = "mean"
    beta: float = 0.0


@dataclass
class TransformerCfg:
    attn_feat_dim: int = 64
    attn_head_dim: int = 64
    num_attn_heads: int = 8
    attn_depth: int = 3
    attn_mlp_dim: int = 64


@dataclass
class EncoderResnetCfg:
    name: Literal["resnet"] = "resnet"
    upsample_interp: Literal["bilinear"] = "bilinear"
    num_layers: int = 4
    use_first_pool: bool = True
    norm_type: Literal["batch", "instance", "group", "none"] = "batch"


@dataclass
class EncoderPrecomputedCfg:
    """Not in the reference: an encoder entry whose forward returns a feature map the caller provides (cached encoder
    output, synthetic benchmark features) -- encoder.EncoderPrecomputed."""
    name: Literal["precomputed"] = "precomputed"
    dim: int = 512


@dataclass
class DensityDecoderMlpCfg:
    name: Literal["density_mlp"] = "density_mlp"
    mlp: MlpCfg = field(default_factory=MlpCfg)
    num_frequencies: int = 10


@dataclass
class ActionDecoderJacobianMlpCfg:
    name: Literal["jacobian_mlp"] = "jacobian_mlp"
    mlp: MlpCfg = field(default_factory=MlpCfg)
    num_frequencies: int = 10
    geometry_feature_dim: int = 15
    use_arm_model: bool = False
    arm_action_dim: Optional[int] = None


@dataclass
class ActionDecoderJacobianTransformerCfg:
    name: Literal["jacobian_transformer"] = "jacobian_transformer"
    mlp: MlpCfg = field(default_factory=MlpCfg)
    transformer: TransformerCfg = field(default_factory=TransformerCfg)
    num_frequencies: int = 10
    geometry_feature_dim: int = 15
    use_arm_model: bool = False
    arm_action_dim: Optional[int] = None


@dataclass
class ActionDecoderFlowMlpCfg:
    """models/decoder/action_decoder_flow.py:32-39 (the field really is spelled ``num_frequncies`` there)."""
    name: Literal["flow_mlp"] = "flow_mlp"
    mlp: MlpCfg = field(default_factory=MlpCfg)
    num_frequncies: int = 10
    geometry_feature_dim: int = 15
    use_arm_model: bool = False
    arm_action_dim: Optional[int] = None


EncoderCfg = Union[EncoderResnetCfg, EncoderPrecomputedCfg]
DensityDecoderCfg = DensityDecoderMlpCfg
ActionDecoderCfg = Union[ActionDecoderJacobianMlpCfg, ActionDecoderFlowMlpCfg, ActionDecoderJacobianTransformerCfg]


@dataclass
class RenderingCfg:
    num_proposal_samples: Tuple[int, ...] = (256,)
    num_nerf_samples: int = 256
    single_jitter: bool = False
    proposal_warmup: int = 5000
    proposal_update_every: int = 5
    use_proposal_weight_anneal: bool = True
    proposal_weights_anneal_max_num_iters: int = 1000
    proposal_weights_anneal_slope: float = 10.0


@dataclass
class ModelCfg:
    action_dim: int = 8
    rendering: RenderingCfg = field(default_factory=RenderingCfg)
    encoder: EncoderCfg = field(default_factory=EncoderResnetCfg)
    density_decoder: DensityDecoderCfg = field(default_factory=DensityDecoderMlpCfg)
    action_decoder: ActionDecoderCfg = field(default_factory=ActionDecoderJacobianMlpCfg)


_ACTION_DECODER_CFGS = {
    "jacobian_mlp": ActionDecoderJacobianMlpCfg,
    "jacobian_transformer": ActionDecoderJacobianTransformerCfg,
    "flow_mlp": ActionDecoderFlowMlpCfg,
}


def _build(cls, data: Dict[str, Any]):
    """Nested dict -> dataclass (what dacite.from_dict does for the reference, config/tools.py:16-25)."""
    hints = get_type_hints(cls)
    kwargs = {}
    known = {f.name for f in fields(cls)}
    for key, value in data.items():
        if key not in known:
            raise KeyError(f"{cls.__name__}: unknown field {key!r}")
        t = hints[key]
        if key == "encoder" and isinstance(value, dict):
            kwargs[key] = _build(EncoderPrecomputedCfg if value.get("name") == "precomputed" else EncoderResnetCfg, value)
        elif key == "action_decoder" and isinstance(value, dict):
            name = value.get("name")
            if name not in _ACTION_DECODER_CFGS:
                raise KeyError(f"unknown action decoder {name!r}; known: {sorted(_ACTION_DECODER_CFGS)}")
            kwargs[key] = _build(_ACTION_DECODER_CFGS[name], value)
        elif is_dataclass(t) and isinstance(value, dict):
            kwargs[key] = _build(t, value)
        elif get_origin(t) is tuple and isinstance(value, (list, tuple)):
            kwargs[key] = tuple(value)
        else:
            kwargs[key] = value
    return cls(**kwargs)


def model_cfg_from_dict(data: Dict[str, Any]) -> ModelCfg:
    """Accepts the content of ``configurations/model/model_*.yaml`` (with the encoder group merged in)."""
    data = dict(data)
    data.pop("defaults", None)
    return _build(ModelCfg, data)
