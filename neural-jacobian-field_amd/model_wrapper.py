"""Training-step contract of ``models/model_wrapper.py`` without Lightning/wandb: ray subsampling,
packing of ``ModelInput``/``ModelTarget`` and the loss expressions (what gets all-reduced under DP).

Reference: ``model_wrapper.py:437-444`` (random_sample_ray_yx_indices), ``:446-551``
(prepare_training_input_output), ``:117-163`` (losses), ``utils/loss_utils.py:9-35`` (ds-nerf depth loss).
Everything here is host-side indexing / scalar reductions on the device the batch lives on.
"""

from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .geometry import denormalize_intrinsics, get_world_rays_with_z
from .model import CameraInput, Model, ModelInput, ModelOutput, ModelTarget, RenderingInput, RobotInput


def random_sample_ray_yx_indices(image_height: int, image_width: int, num_samples: int):
    """model_wrapper.py:437-444 -- one (y, x) set shared by every batch element."""
    idx = torch.floor(torch.rand((num_samples, 2)) * torch.tensor([image_height, image_width])).long()
    return idx[:, 0], idx[:, 1]


def prepare_training_input_output(batch: Dict, mode: str, rays_per_batch: int) -> Tuple[ModelInput, ModelTarget]:
    """model_wrapper.py:446-551.  ``batch`` follows the dataset schema (data/dataset/dataset.py:391-459)."""
    coordinates = batch["scene"]["coordinates"]
    trgt_rgb, trgt_depth = batch["target"]["rgb"], batch["target"]["depth"]
    h, w = coordinates.shape[1:3]
    trgt_flow = trgt_mask = None
    if mode == "perception" or "pixel_motion" not in batch["target"]:
        y, x = random_sample_ray_yx_indices(h, w, rays_per_batch)
        y, x = y.to(coordinates.device), x.to(coordinates.device)
        trgt_rgb = trgt_rgb[:, :, y, x].transpose(1, 2)
        trgt_depth = trgt_depth[:, :, y, x].transpose(1, 2)
        coordinates = coordinates[:, y, x, :]
        if mode != "perception":
            trgt_flow = batch["target"]["flow"][:, :, y, x].transpose(1, 2)
    else:  # tracked pixels (:479-507)
        sel = batch["target"]["pixel_selector"]
        trgt_flow = batch["target"]["pixel_motion"]
        trgt_mask = batch["target"]["pixel_visible_mask"]
        gather = lambda t: torch.gather(t, 1, sel.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
        trgt_rgb = gather(trgt_rgb.flatten(2).transpose(1, 2))
        trgt_depth = gather(trgt_depth.flatten(2).transpose(1, 2))
        coordinates = gather(coordinates.flatten(1, 2))
    origins, directions, z = get_world_rays_with_z(coordinates.contiguous(), batch["target"]["intrinsics"],
                                                   batch["target"]["extrinsics"])
    trgt_depth = trgt_depth / z  # :513-516
    batch["target"]["depth"] = trgt_depth
    model_input = ModelInput(
        camera_input=CameraInput(
            input_image=batch["context"]["rgb"], ctxt_extrinsics=batch["context"]["extrinsics"],
            ctxt_intrinsics=batch["context"]["intrinsics"], trgt_extrinsics=batch["target"]["extrinsics"],
            trgt_intrinsics=denormalize_intrinsics(batch["target"]["intrinsics"], width=w, height=h)),
        rendering_input=RenderingInput(origins=origins, directions=directions, z_near=batch["scene"]["near"],
                                       z_far=batch["scene"]["far"]),
        robot_input=RobotInput(robot_action=batch["context"]["robot_action"]))
    return model_input, ModelTarget(rgb=trgt_rgb, depth=trgt_depth, optical_flow=trgt_flow, visible_mask=trgt_mask)


def ds_nerf_depth_loss(weights, termination_depth, steps, lengths, sigma):
    """utils/loss_utils.py:9-35 (divides by 2*sigma, as the reference does)."""
    mask = termination_depth > 0
    loss = -torch.log(weights + 1.0e-7) * torch.exp(-((steps - termination_depth[..., None, :]) ** 2) / (2 * sigma)) * lengths
    return torch.mean(loss.sum(-2) * mask)


def rgb_loss(output: ModelOutput, target: ModelTarget) -> torch.Tensor:
    """model_wrapper.py:119-121."""
    return F.mse_loss(output.standard_output.rgb, target.rgb)


def flow_loss(output: ModelOutput, target: ModelTarget) -> torch.Tensor:
    """model_wrapper.py:148-160."""
    err = 0.01 * F.mse_loss(output.standard_output.optical_flow, target.optical_flow, reduction="none")
    if target.visible_mask is not None:
        return (err * target.visible_mask.unsqueeze(-1)).sum() / target.visible_mask.sum()
    return err.mean()


def depth_loss(output: ModelOutput, target: ModelTarget, sigma: float = 0.001) -> torch.Tensor:
    """model_wrapper.py:123-136: 0.08 x mean over levels of the ds-nerf loss."""
    wl, sl = output.training_output.weights_list, output.training_output.ray_samples_list
    sig = torch.tensor([sigma], device=target.depth.device)
    total = 0.0
    for wts, smp in zip(wl, sl):
        total = total + ds_nerf_depth_loss(wts, target.depth, (smp.starts + smp.ends) / 2, smp.ends - smp.starts, sig) / len(wl)
    return 0.08 * total


class ModelWrapper(torch.nn.Module):
    """State-dict-compatible shell (``model.*`` prefix, ``depth_sigma`` buffer) around ``Model``:
    ``wrapper.load_state_dict(ckpt["state_dict"])`` works as in the notebooks."""

    def __init__(self, mode: str, rays_per_batch: int, model: Model):
        super().__init__()
        self.model = model
        self.mode = mode
        self.rays_per_batch = rays_per_batch
        self.register_buffer("depth_sigma", torch.tensor([0.001]))
        if mode == "action":  # model_wrapper.py:75-85
            self.model.decoder.freeze_non_action_parameters()
            for name, p in self.model.named_parameters():
                if "decoder" not in name:
                    p.requires_grad = False

    def evaluate_losses(self, batch: Dict) -> Dict[str, torch.Tensor]:
        """Forward + loss values of model_wrapper.py:107-163 (values only: no autograd graph in round 1)."""
        model_input, target = prepare_training_input_output(batch, self.mode, self.rays_per_batch)
        out = self.model.forward(model_input.camera_input, model_input.rendering_input, model_input.robot_input)
        if self.mode == "perception":
            losses = {"loss/rgb": rgb_loss(out, target)}
            if out.training_output is not None:
                losses["loss/depth"] = depth_loss(out, target, float(self.depth_sigma))
        else:
            losses = {"loss/flow_loss": flow_loss(out, target)}
        return losses
