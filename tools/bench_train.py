#!/usr/bin/env python3
"""Training-step timing (not the headline metric): reference batch shape (7 scenes x 256 rays,
configurations/config.yaml:18-20) with the benchmark's 64+64 samples, encoder included, forward + backward + Adam.
``python tools/bench_train.py [action|perception]``: action = only the Jacobian head trains (flow loss); perception =
everything trains (rgb + ds-nerf depth + interlevel + distortion losses, model_wrapper.py:117-146)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import parity_harness as ph
from neural_jacobian_field_amd import synthetic
from neural_jacobian_field_amd.config import model_cfg_from_dict
from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput

# BASELINE config 4: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py <mode>`
# runs one rank per GPU, each on its own scenes, gradients averaged in one RCCL all-reduce per step
WORLD, RANK, LOCAL = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
torch.cuda.set_device(LOCAL)
dev = torch.device("cuda", LOCAL)
if WORLD > 1:
    import torch.distributed as dist
    os.environ["NCCL_DEBUG"] = "WARN"
    dist.init_process_group("nccl", device_id=dev)
if os.environ.get("NJF_MIOPEN_FIND"):
    torch.backends.cudnn.benchmark = True
B, H, W, R, S = 7, 256, 256, 256, 64
case = ph.make_case(B, H, W, R, 8, seed=RANK)
model = Model(model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                                   "action_decoder": {"name": "jacobian_mlp"}}))
sd = synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", 8), seed=0)
model.load_state_dict(sd)
MODE = sys.argv[1] if len(sys.argv) > 1 else "action"
model.to(dev).train()
if os.environ.get("NJF_CHANNELS_LAST"):
    model.encoder.to(memory_format=torch.channels_last)
if MODE == "action":
    model.encoder.eval()
    model.decoder.freeze_non_action_parameters()
    for n, p in model.named_parameters():
        if "decoder" not in n:
            p.requires_grad = False
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-5)
c = case["cams"]; d = lambda t: t.to(dev)
cam = CameraInput(d(torch.rand(B, 3, H, W)), d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(c["trgt_c2w"]), d(case["k_pix"]))
rin = RenderingInput(d(case["origins"]), d(case["directions"]), d(c["z_near"]), d(c["z_far"]))
rob = RobotInput(d(case["action"]))
target = d(torch.randn(B, R, 2))
from neural_jacobian_field_amd import model_wrapper as mw
from neural_jacobian_field_amd.model import ModelTarget
ptarget = ModelTarget(rgb=d(torch.rand(B, R, 3)), depth=d(torch.rand(B, R, 1) + 0.5), optical_flow=None, visible_mask=None)

from neural_jacobian_field_amd.parallel import data_parallel_step
trainable = [p for p in model.parameters() if p.requires_grad]


def compute_loss():
    out = model.forward(cam, rin, rob)
    if MODE == "action":
        loss = 0.01 * torch.nn.functional.mse_loss(out.standard_output.optical_flow, target)
    else:
        tr = out.training_output
        loss = (mw.rgb_loss(out, ptarget) + mw.depth_loss(out, ptarget) + mw.interlevel_loss(tr.weights_list, tr.ray_samples_list)
                + 0.01 * mw.distortion_loss(tr.weights_list, tr.ray_samples_list))
    return loss


def step(i):
    model.step_before_iter(i)
    loss = data_parallel_step(compute_loss, trainable, opt)
    model.step_after_iter(i)
    return loss

for i in range(3):
    step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 10
for i in range(N):
    step(3 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
if os.environ.get("NJF_PROFILE"):  # steady-state kernel breakdown of 5 steps (after warm-up)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(5):
            step(20 + i)
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:18]
    tot = sum(e.device_time_total for e in prof.key_averages())
    print(f"steady-state device time per step: {tot / 5e3:.2f} ms")
    for e in rows:
        print(f"  {e.device_time_total / 5e3:7.3f} ms/step  x{e.count / 5:6.1f}  {e.key[:90]}")
if WORLD > 1:
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
if RANK == 0:
  print(json.dumps({"training_step_ms": round(1e3 * dt, 2), "n_gpus": WORLD, "rays_per_step": WORLD * B * R, "samples": f"{S}+{S}",
                  "train_rays_per_s": round(WORLD * B * R / dt, 1), "mode": {"action": "action (Jacobian head only), encoder fwd included",
                           "perception": "perception (all parameters), encoder fwd+bwd included"}[MODE]}))
