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
    sharding they need an all-reduce of two scalars -- ONE collective: MAX over (-min, max)."""
    flat = step_minmax.reshape(-1, 2)
    if flat.shape[0] == 0:   # a rank without rays (more ranks than rays): neutral element of the MAX all-reduce
        buf = torch.full((2,), float("-inf"), dtype=step_minmax.dtype, device=step_minmax.device)
    else:
        buf = torch.cat([-flat[:, 0], flat[:, 1]]).reshape(2, -1).amax(dim=1)   # (-min, max) in one reduction
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.MAX)
    return torch.clamp(depth, min=-buf[0], max=buf[1])


def enable_ray_sharding(model) -> None:
    """Make ``Model.forward`` correct on a ray shard: its depth clip takes the all-reduced bounds."""
    model.depth_clip = global_depth_clip


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


class ShardedFrameStep:
    """One ray-sharded step of a frame -- every rank renders ITS contiguous ray shard with ``Model.forward`` -- with ONE
    collective per step (VERDICT r02 "next" #5).

    Per rank and step:  Model.forward on the shard (the render kernel writes rgb / depth / flow straight into this rank's
    *packet* and, in its epilogue, per-workgroup partials of the frame-level reductions: the bounds of render_depth's
    tensor-global clip, model.py:277, and the squared-error sums of the photometric / flow loss, model_wrapper.py:117-163)
    -> njf_reduce_frame_partials (partials -> the packet's 16-byte tail) -> ONE all_gather of the packets (pixels AND the four
    scalars) -> njf_assemble_frame (frame [B,R,6] = rgb | depth clipped with the GLOBAL bounds | flow, global loss sums and
    the two losses).  Nothing is concatenated, padded or reduced by ATen ops; buffers are allocated once.

    ``capture()`` records the per-rank compute (forward + reduce) into a HIP graph; the collective and the assemble launch
    stay eager, so the step is: graph launch, all_gather, one kernel.

    The layout logic is plain torch + torch.distributed (covered by the world-size-2 gloo tests with injected stand-ins for
    the two kernels); on a GPU the kernels are the C-ABI entry points -- there is no CPU fallback."""

    def __init__(self, model, batch: int, frame_rays: int, device, world_size: Optional[int] = None, rank: Optional[int] = None,
                 reduce_fn=None, assemble_fn=None, collective: bool = True):
        live = dist.is_initialized() and dist.get_world_size() > 1
        self.world = (dist.get_world_size() if live else 1) if world_size is None else world_size
        self.rank = (dist.get_rank() if live else 0) if rank is None else rank
        self.model, self.batch, self.frame_rays, self.device = model, batch, frame_rays, torch.device(device)
        self.lo, self.hi = shard_bounds(frame_rays, self.world, self.rank)
        n, cap = self.hi - self.lo, -(-frame_rays // self.world)
        self.cap = cap
        self.packet_floats = 6 * batch * cap + 4
        f32 = dict(dtype=torch.float32, device=self.device)
        # every rank's packet has the same length (cap rays per batch element; a shorter shard leaves its tail unused)
        # (``collective=False``: a dry run of one rank of a larger world on a single process -- no exchange, the other ranks'
        # packets stay zero; bench.py --simulate-world)
        # The exchange runs whenever a process group of this world size is up -- also with ONE rank (bench.py --force-dist:
        # the RCCL call path is exercised on a single GPU) -- and is skipped without a group (single process, no RCCL).
        self.collective = bool(collective and dist.is_initialized() and dist.get_world_size() == self.world)
        self.packets = torch.zeros(self.world, self.packet_floats, **f32)
        self.packet = torch.zeros(self.packet_floats, **f32) if self.collective else self.packets[self.rank]
        self.rgb = self.packet[: 3 * batch * n].view(batch, n, 3)
        self.depth = self.packet[3 * batch * cap: 3 * batch * cap + batch * n].view(batch, n, 1)
        self.flow = self.packet[4 * batch * cap: 4 * batch * cap + 2 * batch * n].view(batch, n, 2)
        self.record = self.packet[self.packet_floats - 4:]
        if reduce_fn is None or assemble_fn is None:
            from . import hip
            reduce_fn, assemble_fn = reduce_fn or hip.reduce_frame_partials, assemble_fn or hip.assemble_frame
            groups = hip.frame_partial_groups(batch * n)
        else:
            groups = (batch * n + 3) // 4
        self._reduce, self._assemble = reduce_fn, assemble_fn
        self.partials = torch.empty(max(groups, 1), 4, **f32)
        self.frame = torch.empty(batch, frame_rays, 6, **f32)
        self.scalars = torch.zeros(6, **f32)
        self.trgt_rgb = torch.zeros(batch, n, 3, **f32)
        self.trgt_flow = torch.zeros(batch, n, 2, **f32)
        self.io = {"rgb": self.rgb, "depth": self.depth, "flow": self.flow, "frame_partials": self.partials,
                   "trgt_rgb": self.trgt_rgb, "trgt_flow": self.trgt_flow}
        self.rgb_scale = 1.0 / float(batch * frame_rays * 3)
        self.flow_scale = 0.01 / float(batch * frame_rays * 2)
        self._neutral = torch.tensor([3.0e38, -3.0e38, 0.0, 0.0], **f32)   # the record of a rank without rays, uploaded once
        self.partials[:, 0], self.partials[:, 1] = 3.0e38, -3.0e38         # neutral until the first forward pass writes them
        self.partials[:, 2:] = 0.0
        self._graph = None
        self._static = None
        # True: every step renders a NEW image -- the per-image lin_z projection runs (and is captured) in every step instead
        # of being served from the model's hoisted-map cache (bench.py; a control loop that re-renders one image leaves it False)
        self.new_image_each_step = False

    # ---- pieces ------------------------------------------------------------------------------------------------------
    def set_targets(self, trgt_rgb: Optional[torch.Tensor] = None, trgt_flow: Optional[torch.Tensor] = None) -> None:
        """This rank's slice of the frame's targets ([B,n,3] / [B,n,2]); copied into the step's static buffers."""
        if trgt_rgb is not None:
            self.trgt_rgb.copy_(trgt_rgb)
        if trgt_flow is not None:
            self.trgt_flow.copy_(trgt_flow)

    def local(self, camera_input, rendering_input, robot_input):
        """The rank-local part: Model.forward on the shard + the fold of its partials into the packet's record."""
        if self.hi == self.lo:   # a rank without rays (more ranks than rays): neutral record, nothing to render
            self.record.copy_(self._neutral)    # device-to-device: no host copy per step, capturable
            return None
        m = self.model
        if self.new_image_each_step:
            m.reset_image_cache()
        prev = m.frame_io
        m.frame_io = self.io
        try:
            # the packet views are only written by the INFERENCE path of Model.forward (the training paths return autograd
            # outputs and ignore frame_io): a model with trainable parameters would otherwise leave the packet unwritten
            with torch.no_grad():
                out = m.forward(camera_input, rendering_input, robot_input)
        finally:
            m.frame_io = prev
        self._reduce(self.partials, self.record)
        return out

    def exchange(self):
        """ONE collective (pixels + the 16-byte record of every rank), then the assemble kernel."""
        if self.collective:
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(self.packets.view(-1), self.packet)
            else:
                dist.all_gather([self.packets[k] for k in range(self.world)], self.packet)
        self._assemble(self.packets, self.batch, self.frame_rays, self.frame, self.scalars, self.rgb_scale, self.flow_scale)
        return self.frame, self.scalars

    # ---- the step ----------------------------------------------------------------------------------------------------
    def capture(self, camera_input, rendering_input, robot_input, warmup: int = 2) -> None:
        """Record ``local`` into a HIP graph (the C-ABI launches are plain stream work).  The input tensors of this call
        become the step's static inputs: refill them in place (``.copy_``) between steps -- rays, cameras, command, feature
        map alike.  Everything derived from them is recomputed by the replay: the camera inverses are launched INSIDE the
        graph (the model's per-tensor inverse cache is switched off for warm-up and capture, and emptied), and the per-image
        projection is part of the graph when ``new_image_each_step`` is set."""
        m = self.model
        cached = getattr(m, "inverse_cache_enabled", None)
        if cached is not None:
            m.inverse_cache_enabled = False
            m._inverse_cache.clear()
        try:
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self.local(camera_input, rendering_input, robot_input)
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._static = (camera_input, rendering_input, robot_input)
            self._graph = torch.cuda.CUDAGraph()
            # with a process group alive, its watchdog thread polls events while this thread captures: only calls of the
            # capturing thread itself may invalidate the capture
            mode = "thread_local" if dist.is_initialized() else "global"
            with torch.cuda.graph(self._graph, capture_error_mode=mode):
                self._out = self.local(camera_input, rendering_input, robot_input)
        finally:
            if cached is not None:
                m.inverse_cache_enabled = cached

    def __call__(self, camera_input=None, rendering_input=None, robot_input=None):
        """-> (frame [B,R,6], scalars [6] = (t_min, t_max, S_rgb, S_flow, loss/rgb, loss/flow_loss), the rank's ModelOutput
        whose depth is NOT clipped -- the clipped depth of the whole frame is frame[..., 3]).

        After ``capture()`` the step replays the recorded launches on the STATIC inputs of the capture call; new inputs are
        given by refilling those tensors in place.  Arguments are then accepted only if they ARE the static objects (or
        omitted) -- anything else would be silently ignored, so it raises."""
        if self._graph is not None:
            for given, static in zip((camera_input, rendering_input, robot_input), self._static):
                if given is not None and given is not static:
                    raise ValueError("ShardedFrameStep was captured: refill the static inputs of capture() in place (.copy_) "
                                     "instead of passing new input objects")
            self._graph.replay()
            out = self._out
        else:
            out = self.local(camera_input, rendering_input, robot_input)
        frame, scalars = self.exchange()
        return frame, scalars, out


def sharded_losses(rgb: torch.Tensor, trgt_rgb: torch.Tensor, flow: Optional[torch.Tensor] = None,
                   trgt_flow: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Photometric (+ 0.01 x flow) loss of a ray-sharded step, reduced over all ranks in ONE all-reduce of
    (sum, count) pairs -- the RCCL all-reduce of the image/flow loss of BASELINE.json's north_star."""
    mse_sum = torch.nn.functional.mse_loss
    sums = [mse_sum(rgb, trgt_rgb, reduction="sum").reshape(1)]
    counts = [float(rgb.numel())]
    if flow is not None:
        sums.append(mse_sum(flow, trgt_flow, reduction="sum").reshape(1))
        counts.append(float(flow.numel()))
    buf = torch.cat(sums + [_device_constant(tuple(counts), rgb.device)])
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    n = len(sums)
    out = {"loss/rgb": buf[0] / buf[n]}
    if flow is not None:
        out["loss/flow_loss"] = 0.01 * buf[1] / buf[n + 1]
    return out


_constants: Dict[tuple, torch.Tensor] = {}


def _device_constant(values: Tuple[float, ...], device) -> torch.Tensor:
    """A small constant vector on the device, uploaded once (per-step host-to-device copies of element counts are
    launch latency on the critical path of a 2 ms step)."""
    from . import hip
    return hip.cached_device_constant(_constants, (values, hip.device_key(device)), device,
                                      lambda: torch.tensor(values, dtype=torch.float32))


def allreduce_gradients(parameters, average: bool = True) -> None:
    """DDP semantics for the trainable parameters (reference: Lightning `ddp_find_unused_parameters_true`,
    train.py:67-79): ONE flattened bucket per step -- action mode is 0.37 M parameters = 1.5 MB, far below where
    bucketing matters, so the design point is a single latency-bound RCCL all-reduce over xGMI.

    The bucket covers EVERY parameter that requires grad, in the given order, with zeros where this rank produced no
    gradient -- ranks may disagree on which gradients exist (the proposal nets' `updated` schedule under
    ``set_to_none``, a head the local graph did not reach), and buckets of different sizes would hang or corrupt the
    collective.  A trailing flag per parameter tells whether ANY rank had a gradient; parameters nobody touched keep
    ``grad = None`` (what the optimiser would have seen without data parallelism)."""
    params = [p for p in parameters if p.requires_grad]
    if not params or not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    dev, dt = params[0].device, torch.float32
    pieces = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in params]
    pattern = tuple(0.0 if p.grad is None else 1.0 for p in params)
    flags = _device_constant(pattern, dev)   # uploaded once per pattern: no per-step host-to-device copy
    flat = torch.cat(pieces + [flags])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    # the flags are only READ (a host synchronisation) when this rank is missing a gradient some other rank may have;
    # with every local gradient present -- every step of the reference's two training modes -- nothing is read back
    seen = flat[-len(params):].tolist() if 0.0 in pattern else pattern
    if average:
        flat /= dist.get_world_size()
    off = 0
    for p, any_rank in zip(params, seen):
        n = p.numel()
        if any_rank > 0:
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
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
    owner = getattr(loss_fn, "__self__", None)   # a bound ModelWrapper.training_step: keep its step counter current
    if hasattr(owner, "optimizer_stepped"):
        owner.optimizer_stepped()
    mean = loss.detach().clone().reshape(1)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(mean, op=dist.ReduceOp.SUM)
        mean /= dist.get_world_size()
    return mean[0]
