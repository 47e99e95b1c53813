"""Image encoder: ResNet34 trunk -> 512-channel feature map at H/2 x W/2.

Mirrors ``models/encoder/encoder_resnet.py:24-89`` and its registry (``models/encoder/__init__.py:7-16``).
The convolutions run on stock PyTorch-ROCm (MIOpen): the encoder executes once per image and is
outside the per-ray hot path.  torchvision is not available, so the
BasicBlock [3,4,6,3] trunk is restated here with torchvision's state-dict names
(``model.conv1.weight``, ``model.layer1.0.bn1.running_mean`` ...; ``layer4``/``fc`` are kept so
reference checkpoints load with ``strict=True`` even though the forward never uses them).
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Callable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import EncoderCfg, EncoderResnetCfg  # noqa: F401


def _norm_factory(norm_type: str) -> Optional[Callable[[int], nn.Module]]:
    # model_components/get_norm_layer.py:6-30
    if norm_type == "batch":
        return lambda c: nn.BatchNorm2d(c, affine=True, track_running_stats=True)
    if norm_type == "instance":
        return lambda c: nn.InstanceNorm2d(c, affine=False, track_running_stats=False)
    if norm_type == "group":
        return lambda c: nn.GroupNorm(32, c)
    if norm_type == "none":
        return None
    raise NotImplementedError(f"normalization layer [{norm_type}] is not found")


class _Block(nn.Module):
    def __init__(self, c_in: int, c_out: int, stride: int, norm):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, c_out, 3, stride, 1, bias=False)
        self.bn1 = norm(c_out)
        self.conv2 = nn.Conv2d(c_out, c_out, 3, 1, 1, bias=False)
        self.bn2 = norm(c_out)
        self.downsample = None
        if stride != 1 or c_in != c_out:
            self.downsample = nn.Sequential(nn.Conv2d(c_in, c_out, 1, stride, bias=False), norm(c_out))

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + skip)


class _ResNet34Trunk(nn.Module):
    def __init__(self, norm):
        super().__init__()
        norm = norm or (lambda c: nn.BatchNorm2d(c))
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        c_in = 64
        for i, (c, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
            blocks = []
            for b in range(n):
                blocks.append(_Block(c_in, c, 2 if (b == 0 and i > 1) else 1, norm))
                c_in = c
            setattr(self, f"layer{i}", nn.Sequential(*blocks))
        self.fc = nn.Linear(512, 1000)


class Encoder(nn.Module, ABC):
    """models/encoder/encoder_base.py:10-27."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    @abstractmethod
    def forward(self, rgb: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def get_output_dim(self) -> int: ...


class FeaturePyramid:
    """The encoder's latents BEFORE up-sampling and concatenation (conv1, layer1..3 outputs).  Stands in for the
    [B,512,Hf,Wf] feature tensor on the inference path: the hoisted map is produced from the levels directly
    (njf_project_pyramid), so the concatenated map and its three up-sampled copies are never written."""

    def __init__(self, levels):
        self.levels = list(levels)

    @property
    def shape(self):
        b, _, h, w = self.levels[0].shape
        return torch.Size((b, sum(lv.shape[1] for lv in self.levels), h, w))

    @property
    def device(self):
        return self.levels[0].device

    @property
    def _version(self):
        return sum(lv._version for lv in self.levels)


class EncoderResnet(Encoder):
    def __init__(self, cfg: EncoderResnetCfg):
        super().__init__(cfg)
        self.pretrained = False
        self.use_first_pool = cfg.use_first_pool
        self.model = _ResNet34Trunk(_norm_factory(cfg.norm_type))
        self.num_layers = cfg.num_layers
        self.upsample_interp = cfg.upsample_interp
        for m in self.modules():  # encoder_resnet.py:43-51
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _latents(self, rgb: torch.Tensor):
        m = self.model
        x = m.relu(m.bn1(m.conv1(rgb)))
        pyramid = [x]
        if self.num_layers > 1:
            if self.use_first_pool:
                x = m.maxpool(x)
            for i in range(1, min(self.num_layers, 5)):
                x = getattr(m, f"layer{i}")(x)
                pyramid.append(x)
        return pyramid

    def forward_pyramid(self, rgb: torch.Tensor):
        """Inference path of the fused renderer: the latents without the up-sampling + concatenation of
        encoder_resnet.py:78-86 (a FeaturePyramid), or the plain feature tensor when the configuration is not the
        bilinear 512-channel one the pyramid producer implements."""
        pyramid = self._latents(rgb)
        if self.upsample_interp != "bilinear" or sum(p.shape[1] for p in pyramid) != 512:
            return self.forward(rgb)
        return FeaturePyramid(pyramid)

    def forward(self, rgb: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] -> [B,512,H/2,W/2]: conv1..layer3 outputs, bilinearly upsampled (align_corners=False)
        to the conv1 resolution and concatenated (encoder_resnet.py:53-86)."""
        pyramid = self._latents(rgb)
        size = pyramid[0].shape[-2:]
        pyramid = [F.interpolate(p, size, mode=self.upsample_interp, align_corners=False) for p in pyramid]
        return torch.cat(pyramid, dim=1)

    def get_output_dim(self) -> int:
        return 512


class EncoderPrecomputed(Encoder):
    """``forward`` returns the feature map set with ``set_features`` ([B,dim,Hf,Wf], or a FeaturePyramid): for callers
    that already hold the encoder output -- the control loop re-rendering one image, the benchmark's synthetic feature
    maps, the parity harness.  Has no parameters; not part of the reference's registry."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.dim = cfg.dim
        self.features = None

    def set_features(self, features) -> None:
        self.features = features

    def forward(self, rgb=None):
        if self.features is None:
            raise RuntimeError("EncoderPrecomputed: call set_features(feature_map) before the forward pass")
        return self.features

    def get_output_dim(self) -> int:
        return self.dim


ENCODERS = {"resnet": EncoderResnet, "precomputed": EncoderPrecomputed}


def get_encoder(cfg: EncoderCfg) -> Encoder:
    return ENCODERS[cfg.name](cfg)
