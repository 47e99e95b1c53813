#!/usr/bin/env python3
"""BASELINE config 4: the data-parallel TRAINING step (not the headline metric).

Reference batch shape (7 scenes x 256 rays per rank, configurations/config.yaml:18-20) with the benchmark's 64+64 samples,
encoder included, forward + backward + ONE flattened gradient all-reduce (RCCL) + Adam -- what Lightning DDP does for the
reference (train.py:67-79, models/model_wrapper.py:117-163).  ``--mode action``: only the Jacobian head trains (flow
loss); ``--mode perception``: everything trains (rgb + ds-nerf depth + interlevel + distortion losses).

    python tools/bench_train.py --gpus N [--mode action|perception] [--steps K] [--warmup W]

Started as a plain process it spawns its N ranks itself (neural_jacobian_field_amd.launch, one per GPU, rendezvous on
127.0.0.1); under a launcher it checks WORLD_SIZE == N.  Rank 0 prints ONE JSON line with the rank evidence
(``rccl``: backend, world size, one device record per rank, per-rank step times); a line whose n_gpus differs from
--gpus is refused.  Inputs come from ``synthetic.synthetic_training_batch`` (rank r owns the scenes of seed r) -- nothing
under oracle/ is imported, so the tool travels with the package alone."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd import launch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode_pos", nargs="?", choices=["action", "perception"], default=None, help="(old spelling of --mode)")
    ap.add_argument("--mode", choices=["action", "perception"], default=None)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even with one rank")
    ap.add_argument("--scenes", type=int, default=7, help="scenes per rank (reference: 7, configurations/config.yaml:18-20)")
    ap.add_argument("--rays", type=int, default=256, help="rays per scene (reference: 256)")
    ap.add_argument("--decoder", choices=["jacobian_mlp", "jacobian_transformer", "flow_mlp"], default="jacobian_mlp",
                    help="action decoder (configurations/model/model_allegro.yaml:26 ships jacobian_transformer)")
    ap.add_argument("--samples", type=int, default=64, help="proposal = final samples per ray (model_allegro.yaml: 256)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1234, help="torch seed of every rank (stratified jitter): the same step twice gives the same loss")
    ap.add_argument("--start-step", type=int, default=0,
                    help="global step of the first iteration (20000: steady state -- anneal exponent 1, proposal nets updated every sixth step)")
    ap.add_argument("--matmul-precision", choices=["highest", "high"], default="highest",
                    help="torch.set_float32_matmul_precision of the run.  'high' is what the reference's train.py sets (train.py:64-65: "
                         "TF32 GEMMs in every training step); with the 'auto' settings below it selects the TF32-class forms of this "
                         "backward pass (f16x2 chain + 16-bit storage) for the networks whose forward runs in a split precision")
    ap.add_argument("--backward-precision", choices=["auto", "f32", "f16x2"], default="auto",
                    help="product form of the fused backward chain (training.set_backward_precision): auto (follows --matmul-precision), "
                         "exact fp32, or split fp16 on scaled gradients (fp32-class; the reference trains on TF32 products)")
    ap.add_argument("--storage", choices=["auto", "f32", "f16"], default="auto",
                    help="what the weight-gradient GEMMs read (training.set_storage_precision): auto (follows --matmul-precision), fp32 "
                         "activations / deltas, or fp16 ones with fp32 accumulation (the reference trains on TF32 products)")
    args = ap.parse_args()
    mode = args.mode or args.mode_pos or "action"
    launch.ensure_world(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    launch.reserve_stdout()   # only the JSON line reaches stdout (RCCL's banner, MIOpen's notes go to stderr)
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "the training step has no CPU fallback"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = launch.init_process_group("nccl", dev) if (world > 1 or args.force_dist) else None
    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()

    from neural_jacobian_field_amd import model_wrapper as mw, synthetic, training
    torch.set_float32_matmul_precision(args.matmul_precision)
    training.set_backward_precision(args.backward_precision)
    training.set_storage_precision(args.storage)
    chain_form, storage_form = training.backward_precision("f16f6"), training.storage_precision("f16f6")   # what a split-precision net gets
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import CameraInput, Model, ModelTarget, RenderingInput, RobotInput
    from neural_jacobian_field_amd.parallel import data_parallel_step

    if os.environ.get("NJF_MIOPEN_FIND"):
        torch.backends.cudnn.benchmark = True
    B, H, W, R, S, A = args.scenes, args.height, args.width, args.rays, args.samples, 8
    torch.manual_seed(args.seed)
    model = Model(model_cfg_from_dict({"action_dim": A, "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                                       "action_decoder": {"name": args.decoder}}))
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes(args.decoder, A), seed=0))   # replicated weights
    model.to(dev).train()
    if os.environ.get("NJF_CHANNELS_LAST"):
        model.encoder.to(memory_format=torch.channels_last)
    if mode == "action":
        model.encoder.eval()
        model.decoder.freeze_non_action_parameters()
        for n, p in model.named_parameters():
            if "decoder" not in n:
                p.requires_grad = False
    trainable = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(trainable, lr=1e-4, weight_decay=1e-5)
    b = synthetic.synthetic_training_batch(B, H, W, R, A, seed=rank, device=dev)      # this rank's scenes
    cam = CameraInput(b["image"], b["ctxt_c2w"], b["ctxt_k_norm"], b["trgt_c2w"], b["trgt_k_pix"])
    rin = RenderingInput(b["origins"], b["directions"], b["z_near"], b["z_far"])
    rob = RobotInput(b["action"])
    ptarget = ModelTarget(rgb=b["target_rgb"], depth=b["target_depth"], optical_flow=None, visible_mask=None)

    def compute_loss():
        out = model.forward(cam, rin, rob)
        if mode == "action":
            return 0.01 * torch.nn.functional.mse_loss(out.standard_output.optical_flow, b["target_flow"])
        tr = out.training_output
        return (mw.rgb_loss(out, ptarget) + mw.depth_loss(out, ptarget) + mw.interlevel_loss(tr.weights_list, tr.ray_samples_list)
                + 0.01 * mw.distortion_loss(tr.weights_list, tr.ray_samples_list))

    counter = [args.start_step]

    def step():
        i = counter[0]
        counter[0] += 1
        model.step_before_iter(i)
        loss = data_parallel_step(compute_loss, trainable, opt)     # fwd + bwd + ONE gradient all-reduce + Adam
        model.step_after_iter(i)
        return loss

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(dev)
    local_s = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if os.environ.get("NJF_PROFILE"):  # steady-state kernel breakdown of 5 steps (after warm-up)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(5):
                step()
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:18]
        tot = sum(e.device_time_total for e in prof.key_averages())
        print(f"steady-state device time per step: {tot / 5e3:.2f} ms", file=sys.stderr)
        for e in rows:
            print(f"  {e.device_time_total / 5e3:7.3f} ms/step  x{e.count / 5:6.1f}  {e.key[:90]}", file=sys.stderr)
    evidence = launch.rank_evidence(dist, dev, 1e3 * local_s / args.steps)
    bucket = sum(p.numel() for p in trainable) * 4
    if rank == 0:
        dt = elapsed / args.steps
        line = {"metric": f"training rays/s (config 4: {B} scene(s) x {R} rays per rank, {S}+{S} samples, fwd + bwd + gradient all-reduce + Adam)",
                "value": round(world * B * R / dt, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * dt, 3), "training_step_ms": round(1e3 * dt, 2), "higher_is_better": True, "scaling": "weak",
                "rays_per_step": world * B * R, "samples": f"{S}+{S}", "train_rays_per_s": round(world * B * R / dt, 1),
                "final_loss": float(loss), "gradient_bucket_bytes": bucket, "start_step": args.start_step,
                "decoder": args.decoder,
                "mode": {"action": "action (Jacobian head only), encoder fwd included",
                         "perception": "perception (all parameters), encoder fwd+bwd included"}[mode],
                "dtype": "forward: package default precision; backward chain: "
                         + ("exact fp32 MFMA" if chain_form == "f32" else "f16x2 (split fp16 products on power-of-two-scaled gradients)")
                         + ("; weight-gradient GEMMs: fp32 operands" if storage_form == "f32" else
                            "; weight-gradient GEMMs: fp16 activations x fp16 scaled deltas, fp32 accumulation (16-bit training storage)"),
                "matmul_precision": {"torch.get_float32_matmul_precision": torch.get_float32_matmul_precision(),
                                     "backward_precision": f"{args.backward_precision} -> {chain_form}",
                                     "storage": f"{args.storage} -> {storage_form}",
                                     "note": "the reference's train.py sets 'high' (TF32 GEMMs, train.py:64-65)"},
                "data": "synthetic",
                "config": {"workload": "C4: Allegro training step, ray-batch DP, one flattened gradient all-reduce per step",
                           "parallelism": f"dp{world}"},
                "rccl": evidence}
        launch.check_line(line, args.gpus)
        launch.print_line(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
