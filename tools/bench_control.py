#!/usr/bin/env python3
"""Control-loop latency (not the headline metric): one context image, R tracked rays, 64+64 samples.
Times (a) the encoder, (b) linearize_flow = one fused render of the tracked rays, eager and as a replayed HIP graph,
(c) the Levenberg-Marquardt solve, next to (d) the notebook's route: encode_image once + 100 Adam steps through
infer_optical_flow (notebooks/real_world/2_inverse_dynamics.ipynb cells 26-29)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd import inverse_dynamics as idyn, synthetic
from neural_jacobian_field_amd.config import model_cfg_from_dict
from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput

dev = torch.device("cuda:0")
torch.manual_seed(0)  # the flow residual of the solve depends on the drawn image / command
B, H, W, R, S, A = 1, 256, 256, int(os.environ.get("RAYS", 256)), 64, 8
case = synthetic.synthetic_case(B, H, W, R, A, seed=0, device=dev)   # package-only inputs: the tool travels without oracle/
model = Model(model_cfg_from_dict({"action_dim": A, "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                                   "action_decoder": {"name": "jacobian_mlp"}}))
sd = synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", A), seed=0)
for k in sd:  # a Jacobian field of realistic magnitude (the seeded head is ~1e2 too strong for a well-posed solve)
    if k.startswith("decoder.jacobian_head.lin_out"):
        sd[k] = sd[k] * 0.01
model.load_state_dict(sd)
model.to(dev).eval().requires_grad_(False)
c = case["cams"]; d = lambda t: t.to(dev)
cam = CameraInput(d(torch.rand(B, 3, H, W)), d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(c["trgt_c2w"]), d(case["k_pix"]))
rin = RenderingInput(d(case["origins"]), d(case["directions"]), d(c["z_near"]), d(c["z_far"]))
truth = torch.randn(B, A, device=dev) * 0.1


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, out


with torch.no_grad():
    t_enc, feats = timed(lambda: model.encoder(cam.input_image))
    t_lin, lin = timed(lambda: idyn.linearize_flow(model, cam, rin))
    target = lin.optical_flow(truth)
    t_solve, got = timed(lambda: idyn.solve_action(lin, target, iterations=8))
    err = (lin.optical_flow(got) - target).abs().max().item()
    res = {"rays": R, "samples": f"{S}+{S}", "encoder_ms": round(t_enc, 3), "linearize_ms_incl_encoder": round(t_lin, 3),
           "lm_solve_ms_8_iters": round(t_solve, 3), "flow_residual_px": err}
    try:
        graphed = idyn.GraphedLinearizer(model, cam, rin)
        t_graph, lin_g = timed(lambda: graphed(cam.input_image))
        res["linearize_graph_ms_incl_encoder"] = round(t_graph, 3)
        res["graph_vs_eager_max_abs"] = (lin_g.jacobian - lin.jacobian).abs().max().item()
        ctrl = idyn.GraphedInverseDynamics(model, cam, rin, iterations=8)
        t_ctrl, act_g = timed(lambda: ctrl(cam.input_image, target))
        res["control_step_graph_ms"] = round(t_ctrl, 3)
        res["graph_action_vs_eager_max_abs"] = (act_g - got).abs().max().item()
    except Exception as e:  # noqa: BLE001
        res["graph_error"] = repr(e)[:300]

# the notebook's route
enc = model.encode_image(cam, rin, RobotInput(torch.zeros(B, A, device=dev)))
act = torch.zeros(B, A, device=dev, requires_grad=True)
opt = torch.optim.Adam([act], lr=1e-2)


def adam_100():
    for _ in range(100):
        opt.zero_grad()
        loss = (model.infer_optical_flow(enc, cam, RobotInput(act)) - target).square().mean()
        loss.backward()
        opt.step()


t_adam, _ = timed(adam_100, n=2, warm=1)
res["notebook_route_100_adam_steps_ms"] = round(t_adam, 1)
print(json.dumps(res))
