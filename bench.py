#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s of the fused hot path (BASELINE.json metric).

A *step* is one full eval-mode rendering pass of ``Model.forward`` (reference
``models/model.py:316-396``, encoder excluded -- it is per image, not per ray) over one synthetic
256x256 frame, config C2 of SURVEY.md 8(d): B=1, 65,536 rays, 64 proposal + 64 final samples per
ray, ``jacobian_mlp`` decoder, A=8, fp32.  Inside the timed region, per step and per rank:

    njf_project_features (lin_z hoist of the 512-ch feature map)  ->  njf_proposal_forward
    ->  njf_render_forward  ->  photometric + flow loss against synthetic targets
    ->  (N>1) RCCL all-reduce of the loss

Inputs are resident in HBM when the clock starts.  N>1 is weak scaling: every rank renders its own
frame (rays shard data-parallel, no data-path collective), value = total rays / max-over-ranks time.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W]      (torchrun launches N>1)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 256
S_PROP, S_FINAL, ACTION_DIM = 64, 64, 8

# Algorithmic work (hoisted-lin_z formulation), MACs per point
MAC_PROPOSAL = 172_032
MAC_DENSITY, MAC_JACOBIAN, MAC_COLOR = 173_952, 174_976, 6_272
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_TFLOPS = {"f32": 157.3, "f16x2": 2500.0}
# MFMA instructions issued per algorithmic product block: the f16x2 path evaluates hi*hi + hi*lo + lo*hi
ISSUE_FACTOR = {"f32": 1.0, "f16x2": 3.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rays", type=int, default=8192)
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank")
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--samples", type=int, default=S_FINAL, help="proposal and final samples per ray")
    ap.add_argument("--precision", choices=["f16x2", "f32"], default=None,
                    help="MFMA precision of the fused MLPs (default: package default, f16x2 split with fp32 accumulate)")
    return ap.parse_args()


def cpu_baseline(case, sample_rays: int):
    # the ONLY place bench.py touches oracle/: the reported CPU baseline (never the thing measured as `value`)
    """Time the CPU oracle (a port of the reference's PyTorch path) on a bounded sample of the SAME
    workload: `sample_rays` rays of the 256x256 frame, 64+64 samples, chunked at 2048 rays exactly as
    Model.patch_render does (models/model.py:533)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import parity_harness as ph

    # 32 threads is the fastest setting on the GPU box's 256-core host (tools/cpu_threads_probe.py: 8/16/32/64/128
    # threads -> 760/762/808/707/337 rays/s; all 256 threads oversubscribe these small ops and drop to ~20 rays/s)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sub = dict(case)
    sub["origins"] = case["origins"][:, :sample_rays].contiguous()
    sub["directions"] = case["directions"][:, :sample_rays].contiguous()
    warm = dict(sub)
    warm["origins"], warm["directions"] = sub["origins"][:, :128], sub["directions"][:, :128]
    ph.oracle_forward(warm, S_PROP, S_FINAL)
    t0 = time.perf_counter()
    for lo in range(0, sample_rays, 2048):
        chunk = dict(sub)
        chunk["origins"] = sub["origins"][:, lo:lo + 2048]
        chunk["directions"] = sub["directions"][:, lo:lo + 2048]
        ph.oracle_forward(chunk, S_PROP, S_FINAL)
    dt = time.perf_counter() - t0
    return {"value": round(sample_rays / dt, 1), "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{sample_rays} rays of the same 256x256 frame (64+64 samples), fp32, torch {torch.__version__} "
                      f"CPU, {dt:.1f} s, chunked at 2048 rays like patch_render"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the hot path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ["NCCL_DEBUG"] = "WARN"  # keep RCCL's version banner off stdout: rank 0 prints exactly one JSON line
        if "MASTER_ADDR" not in os.environ:  # --force-dist without torchrun
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()
    from neural_jacobian_field_amd import geometry, hip, synthetic
    from neural_jacobian_field_amd.renderer import FusedRenderer

    # ---- synthetic frame: seeded weights/cameras replicated on every rank (data parallel), a per-rank
    # feature map (each rank renders its own image); rays come from the HIP ray-generation kernel -------------------
    HH, WW, BB, SS = args.height, args.width, args.batch, args.samples
    dev = lambda t: t.to(device)
    params = synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", ACTION_DIM, with_encoder=False), seed=0)
    cams = synthetic.synthetic_cameras(BB)
    feats_cpu = synthetic.synthetic_features(BB, HH, WW, seed=1 + rank)
    action_cpu = synthetic.synthetic_action(BB, ACTION_DIM, seed=2)
    ctxt_c2w, ctxt_k, trgt_c2w = dev(cams["ctxt_c2w"]), dev(cams["ctxt_k_norm"]), dev(cams["trgt_c2w"])
    z_near, z_far, action = dev(cams["z_near"]), dev(cams["z_far"]), dev(action_cpu)
    origins, directions, _ = geometry.full_frame_rays(HH, WW, dev(cams["trgt_k_norm"]), trgt_c2w)
    k_pix = geometry.denormalize_intrinsics(dev(cams["trgt_k_norm"]), WW, HH)
    ctxt_w2c, trgt_w2c = torch.linalg.inv(ctxt_c2w), torch.linalg.inv(trgt_c2w)
    feats = dev(feats_cpu)
    precision = args.precision or hip.DEFAULT_PRECISION
    dev_params = {k: dev(v) for k, v in params.items()}
    renderers = {}
    for prec in ("f16x2", "f32"):
        renderers[prec] = FusedRenderer(device, 1, ACTION_DIM, precision=prec)
        renderers[prec].load_weights(dev_params)
    fr = renderers[precision]
    g = torch.Generator().manual_seed(100 + rank)
    trgt_rgb = dev(torch.rand(BB, HH * WW, 3, generator=g))
    trgt_flow = dev(torch.randn(BB, HH * WW, 2, generator=g))
    rays = BB * HH * WW

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(args.steps)]
    loss_buf = torch.zeros(2, device=device)

    def step(events=None, fr=fr):
        if events:
            events[0].record()
        gmap = fr.project(feats)
        if events:
            events[1].record()
        res = fr.render(gmap, origins, directions, ctxt_c2w, ctxt_k, z_near, z_far, [SS], SS,
                        trgt_c2w=trgt_c2w, trgt_k_pix=k_pix, action=action, ctxt_w2c=ctxt_w2c, trgt_w2c=trgt_w2c,
                        _events=events[2:5] if events else None)
        # photometric + flow loss (model_wrapper.py:117-163), summed locally then all-reduced
        loss_buf[0] = torch.nn.functional.mse_loss(res.rgb, trgt_rgb)
        loss_buf[1] = 0.01 * torch.nn.functional.mse_loss(res.optical_flow, trgt_flow)
        if dist is not None:
            dist.all_reduce(loss_buf)
        if events:
            events[5].record()
        return res

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ev[i])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # the other MFMA precision, measured briefly in the same process (not part of `value`)
    alt = "f32" if precision == "f16x2" else "f16x2"
    alt_steps = max(2, args.steps // 4)
    alt_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(alt_steps)]
    step(fr=renderers[alt])
    torch.cuda.synchronize()
    for i in range(alt_steps):
        step(alt_ev[i], fr=renderers[alt])
    torch.cuda.synchronize()

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * rays * args.steps / elapsed
        k_ms = {"project": 0.0, "proposal": 0.0, "render": 0.0}
        for e in ev:
            k_ms["project"] += e[0].elapsed_time(e[1])
            k_ms["proposal"] += e[2].elapsed_time(e[3])
            k_ms["render"] += e[3].elapsed_time(e[4])
        k_ms = {k: v / args.steps for k, v in k_ms.items()}
        render_flop = 2.0 * rays * SS * (MAC_DENSITY + MAC_JACOBIAN + MAC_COLOR)
        achieved = render_flop / (k_ms["render"] * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", f"r01_render_kernel_hbm_bytes_{precision}.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "rendered rays/s (64 samples/ray, 256^2 image)",
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if precision == "f32" else "f32 (matrix products as an error-compensated 2 x f16 split of the fp32 "
                     "operands, 3 f16 MFMAs per block, fp32 accumulate; same parity bound as the f32-MFMA path)",
            "data": "synthetic",
            "config": {"workload": ("C2: " if (BB, HH, WW, SS) == (1, 256, 256, 64) else "") +
                                   f"Allegro single-view PixelNeRF, B={BB}, {HH}x{WW} rays, {SS} proposal + {SS} final "
                                   "samples/ray, jacobian_mlp, A=8, eval-mode Model.forward (encoder excluded), "
                                   "+ rgb/flow loss" + (" + RCCL all-reduce" if dist is not None else ""),
                       "rays_per_gpu": rays, "parallelism": f"dp{world} (ray-sharded, replicated weights)"},
            "kernel_ms": {k: round(v, 3) for k, v in k_ms.items()},
            "roofline": {"kernel": f"render_kernel<jacobian_mlp, {precision}> (density+colour+Jacobian MLPs + compositing)",
                         "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_TFLOPS[precision], 4), "traffic": traffic,
                         "algorithmic_flop_per_launch": render_flop,
                         "mfma_issue_factor": ISSUE_FACTOR[precision],
                         "frac_of_fp32_mfma_peak": round(achieved / PEAK_TFLOPS["f32"], 4)},
        }
        alt_ms = sum(e[0].elapsed_time(e[5]) for e in alt_ev) / alt_steps
        alt_render = sum(e[3].elapsed_time(e[4]) for e in alt_ev) / alt_steps
        alt_ach = render_flop / (alt_render * 1e-3) / 1e12
        out["other_precision"] = {"precision": alt, "ms_per_step": round(alt_ms, 3), "rays_per_s_per_gpu": round(rays / (alt_ms * 1e-3), 1),
                                  "render_kernel_ms": round(alt_render, 3), "roofline_achieved_tflops": round(alt_ach, 2),
                                  "roofline_peak_tflops": PEAK_TFLOPS[alt], "roofline_frac": round(alt_ach / PEAK_TFLOPS[alt], 4)}
        if world == 1 and not args.no_cpu_baseline and (BB, HH, WW, SS) == (1, 256, 256, 64):
            case = {"params": params, "feats": feats_cpu, "cams": cams, "origins": origins.cpu(), "directions": directions.cpu(),
                    "k_pix": k_pix.cpu(), "action": action_cpu}
            out["cpu_baseline"] = cpu_baseline(case, args.cpu_sample_rays)
        import ctypes
        ctypes.CDLL(None).fflush(None)  # anything native libraries buffered on stdout goes out BEFORE the JSON line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
