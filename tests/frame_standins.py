"""Stand-ins for the two frame-level HIP kernels and for ``Model`` on CPU tensors (TEST INFRASTRUCTURE).

Used by the world-size-2 ``gloo`` tests of ``parallel.ShardedFrameStep`` (tests/test_host_cpu.py) and, through
``bench.py --dry-launch tests/frame_standins.py``, by the launcher test.  Self-contained on purpose -- nothing under oracle/
is imported, so the bench's dry-launch mode never executes oracle code: the two stand-in kernels below are plain tensor
ops, and tests/test_host_cpu.py::test_frame_standins_equal_the_oracle_restatements ties them to oracle/frame_reference.py
(which the GPU tests check the real kernels against)."""
import torch


def _shard_bounds(num_rays, world, rank):
    q, r = divmod(num_rays, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def reduce_frame_partials(partials: torch.Tensor, out4: torch.Tensor) -> None:
    """njf_reduce_frame_partials on CPU tensors: per-group (min t, max t, sum err_rgb^2, sum err_flow^2) -> one record."""
    out4[0], out4[1] = partials[:, 0].min(), partials[:, 1].max()
    out4[2], out4[3] = partials[:, 2].double().sum().float(), partials[:, 3].double().sum().float()


def assemble_frame(packets, batch, rays, frame, scalars6, rgb_scale=0.0, flow_scale=0.0) -> None:
    """njf_assemble_frame on CPU tensors: packets [world, 6 B cap + 4] -> frame [B,R,6] with the GLOBAL depth clip + losses."""
    world, plen = packets.shape
    cap = -(-rays // world)
    rec = packets[:, plen - 4:]
    mn, mx = rec[:, 0].min(), rec[:, 1].max()
    s0, s1 = rec[:, 2].double().sum().float(), rec[:, 3].double().sum().float()
    for k in range(world):
        lo, hi = _shard_bounds(rays, world, k)
        n = hi - lo
        if n == 0:
            continue
        pk = packets[k]
        frame[:, lo:hi, 0:3] = pk[: 3 * batch * n].view(batch, n, 3)
        frame[:, lo:hi, 3] = torch.clamp(pk[3 * batch * cap: 3 * batch * cap + batch * n].view(batch, n), min=mn, max=mx)
        frame[:, lo:hi, 4:6] = pk[4 * batch * cap: 4 * batch * cap + 2 * batch * n].view(batch, n, 2)
    scalars6[0], scalars6[1], scalars6[2], scalars6[3] = mn, mx, s0, s1
    scalars6[4], scalars6[5] = s0 * rgb_scale, s1 * flow_scale


class FakeShardModel:
    """Stands in for Model in the layout test of parallel.ShardedFrameStep: `forward` fills the buffers the step handed over
    (frame_io) with this rank's slice of a known frame and with the per-group partials the render kernel's epilogue writes."""

    def __init__(self, frame_rgb, frame_depth, frame_flow, tmin, tmax, lo, hi):
        self.frame_io = None
        self.args = (frame_rgb, frame_depth, frame_flow, tmin, tmax, lo, hi)

    def reset_image_cache(self):
        return self

    def forward(self, cam, rin, rob):
        rgb, depth, flow, tmin, tmax, lo, hi = self.args
        io = self.frame_io
        io["rgb"].copy_(rgb[:, lo:hi])
        io["depth"].copy_(depth[:, lo:hi])
        io["flow"].copy_(flow[:, lo:hi])
        b, n = rgb.shape[0], hi - lo
        flat = lambda t: t[:, lo:hi].reshape(b * n, -1)
        se_rgb = ((flat(rgb) - io["trgt_rgb"].reshape(b * n, 3)) ** 2).sum(-1)
        se_flow = ((flat(flow) - io["trgt_flow"].reshape(b * n, 2)) ** 2).sum(-1)
        groups = io["frame_partials"].shape[0]
        pad = groups * 4 - b * n
        grp = lambda v, fill: torch.cat([v, torch.full((pad,), fill)]).view(groups, 4)
        io["frame_partials"][:, 0] = grp(flat(tmin)[:, 0], 3.0e38).min(-1).values
        io["frame_partials"][:, 1] = grp(flat(tmax)[:, 0], -3.0e38).max(-1).values
        io["frame_partials"][:, 2] = grp(se_rgb, 0.0).sum(-1)
        io["frame_partials"][:, 3] = grp(se_flow, 0.0).sum(-1)
        return "out"


def known_frame(batch: int = 2, rays: int = 101, seed: int = 1):
    g = torch.Generator().manual_seed(seed)
    rgb, trg = torch.rand(batch, rays, 3, generator=g), torch.rand(batch, rays, 3, generator=g)
    flow, tflow = torch.randn(batch, rays, 2, generator=g), torch.randn(batch, rays, 2, generator=g)
    depth = torch.rand(batch, rays, 1, generator=g) * 12
    tmin, tmax = torch.rand(batch, rays, 1, generator=g) + 0.5, torch.rand(batch, rays, 1, generator=g) + 9
    return dict(rgb=rgb, trg=trg, flow=flow, tflow=tflow, depth=depth, tmin=tmin, tmax=tmax)


def make_frame_step(parallel, world: int, rank: int, batch: int = 2, rays: int = 101):
    """-> (ShardedFrameStep over this rank's ragged shard of a known frame, check(frame, scalars) -> bool)."""
    f = known_frame(batch, rays)
    lo, hi = parallel.shard_bounds(rays, world, rank)
    step = parallel.ShardedFrameStep(FakeShardModel(f["rgb"], f["depth"], f["flow"], f["tmin"], f["tmax"], lo, hi), batch, rays, "cpu",
                                     world_size=world, rank=rank, reduce_fn=reduce_frame_partials, assemble_fn=assemble_frame)
    assert (step.lo, step.hi) == (lo, hi) and step.world == world
    step.set_targets(f["trg"][:, lo:hi], f["tflow"][:, lo:hi])
    mse = torch.nn.functional.mse_loss

    def check(frame, scalars) -> bool:
        ref_depth = torch.clip(f["depth"], f["tmin"].min(), f["tmax"].max())
        return bool(torch.equal(frame[..., 0:3], f["rgb"]) and torch.equal(frame[..., 3:4], ref_depth)
                    and torch.equal(frame[..., 4:6], f["flow"])
                    and abs(scalars[4] - mse(f["rgb"], f["trg"])) < 1e-6
                    and abs(scalars[5] - 0.01 * mse(f["flow"], f["tflow"])) < 1e-6
                    and scalars[0] == f["tmin"].min() and scalars[1] == f["tmax"].max())

    return step, check
