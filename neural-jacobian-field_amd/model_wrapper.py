"""Training-step contract of ``models/model_wrapper.py`` without Lightning/wandb: ray subsampling,
packing of ``ModelInput``/``ModelTarget`` and the loss expressions (what gets all-reduced under DP).

Reference: ``model_wrapper.py:437-444`` (random_sample_ray_yx_indices), ``:446-551``
(prepare_training_input_output), ``:117-163`` (losses), ``utils/loss_utils.py:9-35`` (ds-nerf depth loss).
Everything here is host-side indexing / scalar reductions on the device the batch lives on.
"""

from __future__ import annotations

from typing import Dict, Tuple

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


_SIGMA: Dict[tuple, torch.Tensor] = {}


def _sigma_on(sigma: float, device) -> torch.Tensor:
    """The loss's sigma as a one-element device tensor, uploaded once per (value, device): a per-step torch.tensor(...) is a
    pageable host-to-device copy on the critical path, and illegal while a step is being recorded into a HIP graph."""
    from . import hip
    return hip.cached_device_constant(_SIGMA, (sigma, hip.device_key(device)), device, lambda: torch.tensor([sigma]))


def depth_loss(output: ModelOutput, target: ModelTarget, sigma: float = 0.001) -> torch.Tensor:
    """model_wrapper.py:123-136: 0.08 x mean over levels of the ds-nerf loss."""
    wl, sl = output.training_output.weights_list, output.training_output.ray_samples_list
    sig = _sigma_on(float(sigma), target.depth.device)
    total = 0.0
    for wts, smp in zip(wl, sl):
        total = total + ds_nerf_depth_loss(wts, target.depth, (smp.starts + smp.ends) / 2, smp.ends - smp.starts, sig) / len(wl)
    return 0.08 * total


_EPS = 1.0e-7


def _sdist(ray_samples) -> torch.Tensor:
    """Spacing-domain bin edges [..., S+1] of a RaySamples."""
    return torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], dim=-1)


def _outer_measure(t: torch.Tensor, t_env: torch.Tensor, w_env: torch.Tensor) -> torch.Tensor:
    """For every interval [t_i, t_i+1): the summed envelope weight of all envelope intervals that can overlap it
    (mip-NeRF 360, eq. 13's bound)."""
    cw = torch.cat([torch.zeros_like(w_env[..., :1]), torch.cumsum(w_env, dim=-1)], dim=-1)
    last = w_env.shape[-1] - 1
    lo = (torch.searchsorted(t_env[..., :-1].contiguous(), t[..., :-1].contiguous(), right=True) - 1).clamp(0, last)
    hi = torch.searchsorted(t_env[..., 1:].contiguous(), t[..., 1:].contiguous(), right=True).clamp(0, last)
    return torch.take_along_dim(cw[..., 1:], hi, dim=-1) - torch.take_along_dim(cw[..., :-1], lo, dim=-1)


def interlevel_loss(weights_list, ray_samples_list) -> torch.Tensor:
    """nerfstudio.model_components.losses.interlevel_loss (imported at model_wrapper.py:12; nerfstudio is absent
    and un-pinned, so this restates the published algorithm -- mip-NeRF 360 eq. 13: the proposal histogram must
    upper-bound the final one): sum over proposal levels of mean(max(0, w - bound)^2 / (w + eps)), with the final
    level's weights and edges detached."""
    c = _sdist(ray_samples_list[-1]).detach()
    w = weights_list[-1][..., 0].detach()
    total = 0.0
    for smp, wts in zip(ray_samples_list[:-1], weights_list[:-1]):
        bound = _outer_measure(c, _sdist(smp), wts[..., 0])
        total = total + torch.mean(torch.clip(w - bound, min=0) ** 2 / (w + _EPS))
    return total


def distortion_loss(weights_list, ray_samples_list) -> torch.Tensor:
    """nerfstudio.model_components.losses.distortion_loss (mip-NeRF 360 eq. 15) on the final level, in the spacing
    domain: sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 (t_i+1 - t_i), averaged over rays.

    The pairwise term is evaluated in O(S log S) per ray instead of the O(S^2) outer difference nerfstudio forms: with the bin
    mid-points of a ray in non-decreasing order (they are sorted here; the samplers produce them sorted anyway)
        sum_ij w_i w_j |m_i - m_j| = 2 sum_i w_i (m_i W_i - M_i),   W_i = sum_{j<i} w_j,  M_i = sum_{j<i} w_j m_j
    -- two exclusive prefix sums.  At the shipped sampling (256 final samples, 1,792 rays) the outer form is four passes over a
    470 MB [R,S,S] tensor forwards and as many backwards: 0.9 ms of a 16 ms perception step.  The prefix sums run in float64 (the
    difference m_i W_i - M_i cancels when the weights concentrate at a surface; [R,S] doubles cost nothing), so the value and its
    gradient d/dw_k = 2 sum_j w_j |m_k - m_j| are closer to the exact ones than the fp32 outer form's."""
    t = _sdist(ray_samples_list[-1])
    w = weights_list[-1][..., 0]
    mid = ((t[..., 1:] + t[..., :-1]) / 2).double()
    w64 = w.double()
    # (the samplers hand over sorted bins; the pairwise sum does not depend on the order of the samples, so sorting here keeps the
    #  identity exact for ANY input at the price of one [R,S] sort)
    mid, order = torch.sort(mid, dim=-1)
    w64 = torch.gather(w64, -1, order)
    wm = w64 * mid
    below_w = torch.cumsum(w64, dim=-1) - w64          # exclusive prefix sums
    below_wm = torch.cumsum(wm, dim=-1) - wm
    inter = (2.0 * torch.sum(w64 * (mid * below_w - below_wm), dim=-1)).to(w.dtype)
    intra = torch.sum(w ** 2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return torch.mean(inter + intra)


def distortion_loss_outer(weights_list, ray_samples_list) -> torch.Tensor:
    """The same loss in nerfstudio's literal O(S^2) form (the comparator of tests/test_host_cpu.py; not used by the training path)."""
    t = _sdist(ray_samples_list[-1])
    w = weights_list[-1][..., 0]
    mid = (t[..., 1:] + t[..., :-1]) / 2
    inter = torch.sum(w * torch.sum(w[..., None, :] * torch.abs(mid[..., :, None] - mid[..., None, :]), dim=-1), dim=-1)
    intra = torch.sum(w ** 2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return torch.mean(inter + intra)


class ModelWrapper(torch.nn.Module):
    """State-dict-compatible shell (``model.*`` prefix, ``depth_sigma`` buffer) around ``Model``:
    ``wrapper.load_state_dict(ckpt["state_dict"])`` works as in the notebooks."""

    def __init__(self, mode: str, rays_per_batch: int, model: Model):
        super().__init__()
        self.model = model
        self.mode = mode
        self.rays_per_batch = rays_per_batch
        self.register_buffer("depth_sigma", torch.tensor([0.001]))
        self.global_step = 0  # optimiser steps taken (LightningModule.global_step)
        if mode == "action":  # model_wrapper.py:75-85
            self.model.decoder.freeze_non_action_parameters()
            for name, p in self.model.named_parameters():
                if "decoder" not in name:
                    p.requires_grad = False

    def evaluate_losses(self, batch: Dict) -> Dict[str, torch.Tensor]:
        """Forward + the individually logged loss terms of model_wrapper.py:107-163 (``loss/*`` names as logged there).
        With gradients enabled the terms carry autograd graphs (Model.forward's action / perception training paths)."""
        model_input, target = prepare_training_input_output(batch, self.mode, self.rays_per_batch)
        out = self.model.forward(model_input.camera_input, model_input.rendering_input, model_input.robot_input)
        if self.mode == "perception":
            losses = {"loss/rgb": rgb_loss(out, target)}
            if out.training_output is not None:
                tr = out.training_output
                losses["loss/depth"] = depth_loss(out, target, float(self.depth_sigma))
                losses["loss/interlevel"] = 1.0 * interlevel_loss(tr.weights_list, tr.ray_samples_list)
                losses["loss/distortion"] = 0.01 * distortion_loss(tr.weights_list, tr.ray_samples_list)
        else:
            losses = {"loss/flow_loss": flow_loss(out, target)}
        return losses

    def training_step(self, batch: Dict, batch_idx: int = 0) -> torch.Tensor:
        """model_wrapper.py:107-146: the scalar that is back-propagated (and, under data parallelism, whose gradient
        bucket is all-reduced: parallel.allreduce_gradients).  Lightning brackets every batch with
        on_train_batch_start / on_train_batch_end (model_wrapper.py:575-581), which drive the proposal-weight anneal and
        the sampler's update schedule from ``global_step``; without Lightning this method does the bracketing itself
        (``global_step`` counts optimiser steps: call ``optimizer_stepped()`` -- or set the attribute -- after each one)."""
        self.on_train_batch_start(batch, batch_idx)
        loss = sum(self.evaluate_losses(batch).values())
        self.on_train_batch_end(None, batch, batch_idx)
        return loss

    def on_train_batch_start(self, batch=None, batch_idx: int = 0) -> None:
        self.model.step_before_iter(self.global_step)

    def on_train_batch_end(self, outputs=None, batch=None, batch_idx: int = 0) -> None:
        self.model.step_after_iter(self.global_step)

    def optimizer_stepped(self) -> None:
        """Lightning increments ``global_step`` per optimiser step; loops without it call this after ``optimizer.step()``."""
        self.global_step += 1

    def configure_optimizers(self, lr: float, warm_up_steps: int):
        """model_wrapper.py:87-105: Adam(weight_decay=1e-5) + linear warm-up."""
        optimizer = torch.optim.Adam([p for p in self.parameters() if p.requires_grad], lr=lr, weight_decay=1e-5)
        warm_up = torch.optim.lr_scheduler.LinearLR(optimizer, 1 / warm_up_steps, 1, total_iters=warm_up_steps)
        return optimizer, warm_up
