#!/usr/bin/env python3
"""Diagnosis (GPU box): where do the perception-mode gradients of the proposal net / the colour head leave the oracle?
Replays the setup of tests/test_training_gpu.py::test_perception_mode_gradients_match_oracle_autograd and compares, stage by
stage, HIP against the oracle: per-level densities and weights, the upstream gradients of the weights (from the losses), the
density gradients the compositing backward returns, and the colour-head dumps between MFMA precisions."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import __graft_entry__ as g_
g_.build()
import njf_oracle as orc
import parity_harness as ph
from neural_jacobian_field_amd import synthetic, training
from neural_jacobian_field_amd.config import model_cfg_from_dict
from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput

rel = lambda a, b: ((a.detach().double().cpu() - b.detach().double().cpu()).abs().max() / (b.detach().double().cpu().abs().max() + 1e-30)).item()
dev = torch.device("cuda:0")
B, H, W, R, S = 2, 16, 16, 40, 32
case = ph.make_case(B, H, W, R, 8, seed=4)
full = synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", 8), seed=4)
full.update(case["params"])
model = Model(model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                                   "action_decoder": {"name": "jacobian_mlp"}}))
model.load_state_dict(full, strict=True)
model.to(dev).train()
model.encoder.eval()
for smp in (model.proposal_sampler.initial_sampler, model.proposal_sampler.pdf_sampler):
    smp.train_stratified = False
g2 = torch.Generator().manual_seed(9)
image = torch.rand(B, 3, H, W, generator=g2)
c = case["cams"]; d = lambda t: t.to(dev)
cam = CameraInput(d(image), d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(c["trgt_c2w"]), d(case["k_pix"]))
rin = RenderingInput(d(case["origins"]), d(case["directions"]), d(c["z_near"]), d(c["z_far"]))
rob = RobotInput(d(case["action"]))
g3 = torch.Generator().manual_seed(21)
t_rgb = torch.rand(B, R, 3, generator=g3); t_depth = torch.rand(B, R, 1, generator=g3) * 0.5 + 0.6
sig = torch.tensor([0.05])

def loss_fn(rgb, depth, wl, se, to):
    loss = torch.nn.functional.mse_loss(rgb, to(t_rgb)) + 0.1 * (depth - to(t_depth)).abs().mean()
    for w, (st, en) in zip(wl, se):
        loss = loss + 0.08 * orc.ds_nerf_depth_loss(w, to(t_depth), (st + en) / 2, en - st, to(sig)) / len(wl)
    return loss

# ---- oracle with retained intermediate gradients
params = {k: v.clone() for k, v in full.items()}
for k, v in params.items():
    if v.is_floating_point() and "running_" not in k:
        v.requires_grad_(True)
ref = orc.model_forward(params, input_image=image, ctxt_c2w=c["ctxt_c2w"], ctxt_k_norm=c["ctxt_k_norm"], trgt_c2w=c["trgt_c2w"],
                        trgt_k_pix=case["k_pix"], origins=case["origins"], directions=case["directions"], z_near=c["z_near"],
                        z_far=c["z_far"], action=case["action"], num_proposal_samples=[S], num_nerf_samples=S, decoder_kind="jacobian_mlp")
for w in ref.weights_list:
    w.retain_grad()
loss_ref = loss_fn(ref.rgb, ref.depth, ref.weights_list, [(x.starts, x.ends) for x in ref.samples_list], lambda t: t)
loss_ref.backward()

for prec in ("f32", "f16f6"):
    model.set_precision(prec)
    model.zero_grad(set_to_none=True)
    stash = {}
    orig = training.CompositeFunction.backward
    def spy(ctx, *gs, _orig=orig):
        out = _orig(ctx, *gs)
        stash.setdefault("calls", []).append({"g_w": gs[0], "g_sigma": out[2], "sigma": ctx.saved_tensors[1], "deltas": ctx.saved_tensors[0]})
        return out
    training.CompositeFunction.backward = staticmethod(spy)
    out = model.forward(cam, rin, rob)
    tr = out.training_output
    for w in tr.weights_list:
        w.retain_grad()
    loss = loss_fn(out.standard_output.rgb, out.standard_output.depth, tr.weights_list, [(x.starts, x.ends) for x in tr.ray_samples_list], d)
    loss.backward()
    training.CompositeFunction.backward = staticmethod(orig)
    print(f"== {prec}: loss {rel(loss.reshape(1), loss_ref.reshape(1)):.2e}")
    for lvl in range(2):
        print(f"  level {lvl}: weights {rel(tr.weights_list[lvl], ref.weights_list[lvl]):.2e}  g_weights {rel(tr.weights_list[lvl].grad, ref.weights_list[lvl].grad):.2e}"
              f"  starts {rel(tr.ray_samples_list[lvl].starts, ref.samples_list[lvl].starts):.2e}")
    if prec == "f32" and os.environ.get("NJF_DIAG_DETAIL"):
        # round 4 (ADVICE r03 / VERDICT r03 "weak" #2): WHERE does the proposal level's upstream gradient leave the oracle?
        # element by element at level 0 (uniform samples: identical positions on both sides)
        wh, wr = tr.weights_list[0].detach().cpu().double().flatten(), ref.weights_list[0].detach().double().flatten()
        gh, gr = tr.weights_list[0].grad.cpu().double().flatten(), ref.weights_list[0].grad.double().flatten()
        call0 = [c_ for c_ in stash["calls"] if c_["sigma"].shape[-2] == S and c_["g_w"] is not None]
        print(f"  level-0 detail: max|w| {wr.abs().max():.3e}  max|g_w| {gr.abs().max():.3e}")
        idx = (gh - gr).abs().argsort(descending=True)[:8]
        for i in idx.tolist():
            print(f"    elem {i:5d}: w_hip {wh[i]:.6e} w_ref {wr[i]:.6e} (abs diff {abs(wh[i]-wr[i]):.2e}, rel {abs(wh[i]-wr[i])/max(abs(wr[i]),1e-300):.2e})"
                  f"  g_hip {gh[i]:.5e} g_ref {gr[i]:.5e} (rel {abs(gh[i]-gr[i])/max(abs(gr[i]),1e-300):.2e})")
        # relative error of w as a function of its size
        for lo_, hi_ in ((0, 1e-9), (1e-9, 1e-7), (1e-7, 1e-5), (1e-5, 1e-3), (1e-3, 1)):
            m_ = (wr >= lo_) & (wr < hi_)
            if m_.any():
                print(f"    w in [{lo_:.0e},{hi_:.0e}): n {int(m_.sum()):5d}  max abs dw {(wh-wr)[m_].abs().max():.2e}  max rel dw {((wh-wr).abs()/wr.clamp_min(1e-300))[m_].max():.2e}"
                      f"  share of |g_w|_1 {(gr[m_].abs().sum()/gr.abs().sum()):.3f}  max rel dg {((gh-gr).abs()/gr.abs().clamp_min(1e-300))[m_].max():.2e}")
        # the same level-0 densities: HIP's sigma (what the backward saved) against the oracle's
        sig_h = [c_["sigma"] for c_ in stash["calls"]][-1].detach().cpu().double().flatten() if stash.get("calls") else None
        # weights recomputed in float64 from HIP's OWN sigma: is w_hip what its sigma implies?
        for c_ in stash["calls"]:
            if c_["g_w"] is not None and c_["sigma"].shape[-2] == S:
                w64 = Model._weights_from_density(c_["deltas"].cpu().double(), c_["sigma"].detach().cpu().double()).flatten()
                cand = tr.weights_list[0].detach().cpu().double().flatten()
                if w64.shape == cand.shape:
                    print(f"    a composite call: |w_kernel - w64(sigma_kernel)| max abs {(cand - w64).abs().max():.2e}, at small w (<1e-5): max abs {(cand - w64)[w64 < 1e-5].abs().max() if (w64 < 1e-5).any() else 0:.2e} max rel {((cand - w64).abs()/w64.clamp_min(1e-300))[w64 < 1e-5].max() if (w64 < 1e-5).any() else 0:.2e}")
    # the gradient w.r.t. the weights, evaluated by the ORACLE's formula on HIP's weights: is the difference in g_w explained by the weights?
    for lvl in range(2):
        w = tr.weights_list[lvl].detach().cpu().clone().requires_grad_(True)
        smp = ref.samples_list[lvl]
        l = 0.08 * orc.ds_nerf_depth_loss(w, t_depth, (smp.starts + smp.ends) / 2, smp.ends - smp.starts, sig) / 2
        l.backward()
        gw_hip = tr.weights_list[lvl].grad.cpu()
        if lvl == 0:
            print(f"  level 0: ds-nerf g_w recomputed on the CPU from HIP's weights vs HIP autograd {rel(gw_hip, w.grad):.2e}; vs oracle's g_w {rel(w.grad, ref.weights_list[0].grad):.2e}")
    for call in stash.get("calls", []):
        s_ = call["sigma"].detach().cpu().double().requires_grad_(True)
        w = Model._weights_from_density(call["deltas"].cpu().double(), s_)
        (w * call["g_w"].cpu().double()).sum().backward() if call["g_w"] is not None else None
        if call["g_w"] is not None and call["sigma"].shape[-2] == S:
            print(f"  composite backward (S={call['sigma'].shape[-2]}): kernel g_sigma vs float64 autograd of the weights term only {rel(call['g_sigma'], s_.grad):.2e} (final level also carries rgb/depth terms)")
    pn = dict(model.proposal_networks[0].density_head.named_parameters())
    for k in ("lin_out.weight", "lin_out.bias", "blocks.0.fc_0.weight"):
        print(f"  proposal grad {k}: {rel(pn[k].grad, params['proposal_networks.0.density_head.' + k].grad):.2e}")
    ch = dict(model.decoder.color_head.named_parameters())
    for k in ("0.weight", "2.weight", "4.weight"):
        print(f"  colour grad {k}: {rel(ch[k].grad, params['decoder.color_head.' + k].grad):.2e}")

# ---- colour-head dumps between precisions (same weights, same rays, eval forward with dumps)
model.eval()
dumps = {}
with torch.no_grad():
    feats = model.encoder.forward(cam.input_image)
    for prec in ("f32", "f16x2", "f16f6"):
        model.set_precision(prec)
        outs, bins, *_ = model._fused_render(cam, rin, rob, feats, want_lists=True, want_vis=False, want_samples=False, dump_perception=True,
                                             final_bins=dumps.get("bins"))
        dumps.setdefault("bins", bins)
        dumps[prec] = {k: outs[k].clone() for k in ("col_in", "col_act", "den_act", "color", "density", "jac_pe")}
for prec in ("f16x2", "f16f6"):
    print(f"dumps {prec} vs f32 at identical bins: " + "  ".join(f"{k} {rel(dumps[prec][k], dumps['f32'][k]):.2e}" for k in dumps["f32"]))
a, b_ = dumps["f32"]["col_act"], dumps["f16x2"]["col_act"]
print("col_act[0] / col_act[1] f32-vs-f16x2:", rel(b_[0], a[0]), rel(b_[1], a[1]))
