"""Data-parallel ray sharding over one process per GPU (``torch.distributed``; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

The reference's only parallelism is Lightning DDP over scenes (``train.py:67-79``).  Rays are
independent units, so a frame shards by ray with replicated weights and feature map and
NO data-path collective; what crosses xGMI is
  * the scalar photometric/flow loss (sum + count -> all_reduce SUM),
  * the two scalars of render_depth's tensor-global clip (model.py:277 -> all_reduce MIN/MAX),
  * optionally the rendered pixels (all_gather of [R/G, C] shards) to assemble a frame.
Messages are tiny, so the design point is latency: one flattened buffer per collective.
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_rays: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced ray ranges (first ``num_rays % world_size`` ranks get one extra ray)."""
    q, r = divmod(num_rays, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_rays(origins: torch.Tensor, directions: torch.Tensor, world_size: Optional[int] = None,
               rank: Optional[int] = None):
    """Slice [B,R,3] ray tensors to this rank's contiguous shard."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(origins.shape[1], world_size, rank)
    return origins[:, lo:hi].contiguous(), directions[:, lo:hi].contiguous(), (lo, hi)


def allreduce_mean_loss(local_sq_err_sum: torch.Tensor, local_count: torch.Tensor) -> torch.Tensor:
    """Global mean of a per-element loss from per-rank (sum, count): identical to the unsharded
    ``mse_loss`` because the loss is a mean over rays (model_wrapper.py:119-121,148-160)."""
    buf = torch.stack([local_sq_err_sum.reshape(()).to(torch.float32), local_count.reshape(()).to(torch.float32)])
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf[0] / buf[1]


def global_depth_clip(depth: torch.Tensor, step_minmax: torch.Tensor) -> torch.Tensor:
    """render_depth's clip bounds are min/max over the WHOLE [B,R,S] step tensor (model.py:277); under ray
    sharding they need an all-reduce(MIN/MAX) of two scalars."""
    lo = step_minmax[..., 0].min().reshape(1)
    hi = step_minmax[..., 1].max().reshape(1)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return torch.clamp(depth, min=lo[0], max=hi[0])


def gather_frame(shard: torch.Tensor, num_rays: int) -> torch.Tensor:
    """all_gather ragged [B,R_k,C] shards into the full [B,R,C] tensor on every rank."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return shard
    world = dist.get_world_size()
    sizes = [shard_bounds(num_rays, world, k) for k in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    b, rk, c = shard.shape
    buf = torch.zeros(b, pad, c, dtype=shard.dtype, device=shard.device)
    buf[:, :rk] = shard
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:, : hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=1)


def sharded_losses(rgb: torch.Tensor, trgt_rgb: torch.Tensor, flow: Optional[torch.Tensor] = None,
                   trgt_flow: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Photometric (+ 0.01 x flow) loss of a ray-sharded step, reduced over all ranks."""
    out = {"loss/rgb": allreduce_mean_loss(((rgb - trgt_rgb) ** 2).sum(), torch.tensor(float(rgb.numel()), device=rgb.device))}
    if flow is not None:
        out["loss/flow_loss"] = 0.01 * allreduce_mean_loss(((flow - trgt_flow) ** 2).sum(),
                                                           torch.tensor(float(flow.numel()), device=flow.device))
    return out


def allreduce_gradients(parameters, average: bool = True) -> None:
    """DDP semantics for the trainable parameters (reference: Lightning `ddp_find_unused_parameters_true`,
    train.py:67-79): ONE flattened bucket per step -- action mode is 0.37 M parameters = 1.5 MB, far below where
    bucketing matters, so the design point is a single latency-bound RCCL all-reduce over xGMI."""
    params = [p for p in parameters if p.grad is not None]
    if not params or not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def data_parallel_step(loss_fn, parameters: Sequence[torch.nn.Parameter], optimizer: torch.optim.Optimizer) -> torch.Tensor:
    """One optimiser step under data parallelism (BASELINE config 4; reference: Lightning DDP, train.py:67-79): every
    rank evaluates ``loss_fn()`` on ITS scenes / ray shard (e.g. ``lambda: wrapper.training_step(batch)``), gradients
    are averaged in one flattened RCCL all-reduce, every rank applies the identical update.  Returns the loss
    averaged over ranks (what rank 0 would log)."""
    optimizer.zero_grad(set_to_none=True)
    loss = loss_fn()
    loss.backward()
    parameters = list(parameters)
    allreduce_gradients(parameters, average=True)
    optimizer.step()
    mean = loss.detach().clone().reshape(1)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(mean, op=dist.ReduceOp.SUM)
        mean /= dist.get_world_size()
    return mean[0]
