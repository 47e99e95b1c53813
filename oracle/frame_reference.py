"""Tensor-op restatements of the two frame-level kernels of a ray-sharded step (TEST INFRASTRUCTURE, same status as
njf_oracle.py / lm_reference.py: imported by tests/ only -- the checker of njf_reduce_frame_partials / njf_assemble_frame
on the GPU, and their stand-ins in the world-size-2 gloo test of parallel.ShardedFrameStep on CPU tensors).

Reference semantics: render_depth's clip uses the min / max over the WHOLE step tensor (models/model.py:277); rgb loss =
mse over all rays (models/model_wrapper.py:119-121); flow loss = 0.01 * mse (:148-160)."""
import torch


def reduce_frame_partials(partials: torch.Tensor, out4: torch.Tensor) -> None:
    out4[0] = partials[:, 0].min()
    out4[1] = partials[:, 1].max()
    out4[2] = partials[:, 2].double().sum().float()
    out4[3] = partials[:, 3].double().sum().float()


def shard_bounds(num_rays, world, rank):
    q, r = divmod(num_rays, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def assemble_frame(packets, batch, rays, frame, scalars6, rgb_scale=0.0, flow_scale=0.0) -> None:
    world, plen = packets.shape
    cap = -(-rays // world)
    rec = packets[:, plen - 4:]
    mn, mx = rec[:, 0].min(), rec[:, 1].max()
    s0, s1 = rec[:, 2].double().sum().float(), rec[:, 3].double().sum().float()
    for k in range(world):
        lo, hi = shard_bounds(rays, world, k)
        n = hi - lo
        if n == 0:
            continue
        pk = packets[k]
        frame[:, lo:hi, 0:3] = pk[: 3 * batch * n].view(batch, n, 3)
        frame[:, lo:hi, 3] = torch.clamp(pk[3 * batch * cap: 3 * batch * cap + batch * n].view(batch, n), min=mn, max=mx)
        frame[:, lo:hi, 4:6] = pk[4 * batch * cap: 4 * batch * cap + 2 * batch * n].view(batch, n, 2)
    scalars6[0], scalars6[1], scalars6[2], scalars6[3] = mn, mx, s0, s1
    scalars6[4], scalars6[5] = s0 * rgb_scale, s1 * flow_scale
