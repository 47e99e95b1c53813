#!/usr/bin/env python3
"""Round-6 golden vectors of the ``flow_mlp`` decoder beyond inference, produced by importing the *reference* (read-only,
/root/reference) like make_golden.py / make_golden_r02.py:

  model_flow_train.npz   ``ActionDecoderFlowMlp`` built with ``use_arm_model`` (``flow_head_arm``, action_decoder_flow.py:109-116),
                         arm_action_dim == action_dim (compute_flow concatenates the robot action itself, :168-172, so no other
                         arm width runs in the reference).  Per mode ("regular", "arm" = ``switch_mode``, :122-123, :163-166):
                           * ``Model.forward`` end to end (rgb, depth, optical_flow, and with compute_vis_features the composited 640
                             hidden features of the flow head) and the decoder on the fp32 run's final sample positions (density,
                             colour, scene flow, the per-sample hidden features of the first 2 rays), in fp32 and in float64;
                           * the reference's ACTION-MODE training gradient: parameters frozen exactly as
                             ``ModelWrapper.freeze_parameters`` does (models/model_wrapper.py:75-85 ->
                             ``freeze_non_action_parameters``, action_decoder_flow.py:281-288: every decoder parameter whose name does
                             not contain "flow_head"; every non-decoder parameter), loss = 0.01 * mse(optical_flow, target)
                             (model_wrapper.py:148-160 without a visibility mask), autograd through the whole reference model;
                             gradients of the ACTIVE flow head in fp32, and per parameter their max-norm distance from the float64
                             run's (``<mode>.floor64.<name>``: the floor of the comparison); the inactive head receives none
                             (asserted here).

Usage:  python tests/golden/make_golden_r06_flow.py     (build container only: the reference never travels)
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (shims + helpers; also puts the repo root and oracle/ on sys.path)

save, rigid, randn, rand, k_norm, load_seeded = mg.save, mg.rigid, mg.randn, mg.rand, mg.k_norm, mg.load_seeded

FEATURE_RAYS = 2
RESNET_ORDER = (["lin_in.weight", "lin_in.bias"]
                + [f"blocks.{b}.{fc}.{wb}" for b in range(5) for fc in ("fc_0", "fc_1") for wb in ("weight", "bias")]
                + [f"lin_z.{i}.{wb}" for i in range(3) for wb in ("weight", "bias")] + ["lin_out.weight", "lin_out.bias"])


def main():
    mg.install_shims()
    torch.set_num_threads(1)
    from neural_jacobian_field.rendering import geometry
    from neural_jacobian_field.model_components.resnet_fc import MlpCfg
    from neural_jacobian_field.models import model as ref_model
    from neural_jacobian_field.models.decoder import ActionDecoderFlowMlpCfg, DensityDecoderMlpCfg
    from neural_jacobian_field.models.decoder.action_decoder import PixelEncoding
    from neural_jacobian_field.models.encoder import EncoderResnetCfg
    from neural_jacobian_field.utils import convention

    A = 5
    mlp_cfg = MlpCfg(n_blocks=5, d_hidden=128, combine_layer=3, combine_type="mean", beta=0.0)
    enc_cfg = EncoderResnetCfg(name="resnet", upsample_interp="bilinear", num_layers=4, use_first_pool=True, norm_type="batch")
    dens_cfg = DensityDecoderMlpCfg(name="density_mlp", mlp=mlp_cfg)
    dec_cfg = ActionDecoderFlowMlpCfg(name="flow_mlp", mlp=mlp_cfg, use_arm_model=True, arm_action_dim=A)
    rcfg = ref_model.RenderingCfg(num_proposal_samples=(16,), num_nerf_samples=12, single_jitter=False, proposal_warmup=5000,
                                  proposal_update_every=5, use_proposal_weight_anneal=True,
                                  proposal_weights_anneal_max_num_iters=1000, proposal_weights_anneal_slope=10.0)
    model = ref_model.Model(ref_model.ModelCfg(action_dim=A, rendering=rcfg, encoder=enc_cfg, density_decoder=dens_cfg,
                                               action_decoder=dec_cfg))
    load_seeded(model, "", seed=0)
    model.eval()
    assert any(k.startswith("decoder.flow_head_arm.") for k in model.state_dict())

    # ---- the scene of make_golden.py / make_golden_r02.py (same seeds) -------------------------------------------------
    B, H, W = 2, 16, 16
    coords16, _ = geometry.get_pixel_coordinates(H, W)
    ctx_c2w = torch.eye(4)[None].repeat(B, 1, 1) + 0.0
    ctx_c2w[1] = rigid(21, 1)[0]
    trg_c2w = rigid(22, B)
    trg_c2w[:, :3, 3] *= 0.5
    Kn = k_norm(B)
    image = rand(23, B, 3, H, W)
    sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(24))[:20]
    xy16 = coords16.reshape(1, -1, 2)[:, sel].repeat(B, 1, 1)
    ro, rd, _ = geometry.get_world_rays_with_z(xy16, Kn, trg_c2w)
    kpix = convention.denormalize_intrinsics(Kn, width=W, height=H)
    z_near, z_far = torch.tensor([0.5, 0.4]), torch.tensor([10.0, 6.0])
    action = 0.5 * randn(25, B, A)
    target = 3.0 * randn(27, B, ro.shape[1], 2)

    def inputs(dtype):
        c = lambda t: t.to(dtype)
        cam = ref_model.CameraInput(input_image=c(image), ctxt_extrinsics=c(ctx_c2w), ctxt_intrinsics=c(Kn),
                                    trgt_extrinsics=c(trg_c2w), trgt_intrinsics=c(kpix))
        rin = ref_model.RenderingInput(origins=c(ro), directions=c(rd), z_near=c(z_near), z_far=c(z_far))
        return cam, rin, ref_model.RobotInput(robot_action=c(action))

    def freeze_like_the_wrapper(m):
        """models/model_wrapper.py:75-85 in dataset.mode == "action"."""
        for p in m.parameters():
            p.requires_grad = True
        m.decoder.freeze_non_action_parameters()
        for name, p in m.named_parameters():
            if "decoder" not in name:
                p.requires_grad = False

    def evaluate(m, dtype, fixed_positions=None):
        cam, rin, rob = inputs(dtype)
        out = {}
        with torch.no_grad():
            feats = m.encoder.forward(cam.input_image)
            res = m.forward(cam, rin, rob, compute_vis_features=True)
            penc = PixelEncoding(features=feats, extrinsics=cam.ctxt_extrinsics, intrinsics=cam.ctxt_intrinsics, action=rob.robot_action)
            rb = m.compute_ray_bundle(rin)
            samples, pos, dirs, wl, sl = m.compute_proposal(rb, penc)
            out.update(features=feats, rgb=res.standard_output.rgb, depth=res.standard_output.depth,
                       optical_flow=res.standard_output.optical_flow, final_positions=pos)
            p_fin = pos if fixed_positions is None else fixed_positions.to(dtype)
            dec = m.decoder.forward(world_space_xyz=p_fin, world_space_dir=rin.directions[..., None, :].expand(p_fin.shape),
                                    pixel_encoding=penc)
            out.update(dec_density=dec.density, dec_color=dec.color, dec_flow=dec.flow)
            # the flow head's 640 hidden features (action_decoder_flow.py:168-176 = ResnetFC.forward(compute_features=True)): per sample
            # on FEATURE_RAYS rays of every batch element (the whole tensor is 1.2 MB per mode and precision), and composited along
            # every ray by Model.forward (model.py:381-390)
            out.update(dec_action_features=dec.action_features[:, :FEATURE_RAYS], vis_action_features=res.vis_output.action_features,
                       vis_weights=res.vis_output.weights, vis_ray_positions_warped=res.vis_output.ray_positions_warped)
        # the action-mode training gradient (autograd through the whole reference model, frozen like the wrapper)
        freeze_like_the_wrapper(m)
        m.zero_grad(set_to_none=True)
        res = m.forward(cam, rin, rob, compute_vis_features=False)
        loss = 0.01 * torch.nn.functional.mse_loss(res.standard_output.optical_flow, target.to(dtype))
        loss.backward()
        out["loss"] = loss.detach().reshape(1)
        active = "flow_head_arm." if m.decoder.mode == "arm" else "flow_head."
        named = dict(m.decoder.named_parameters())
        for n in RESNET_ORDER:
            g = named[active + n].grad
            assert g is not None and torch.isfinite(g).all(), (active, n)
            out["grad." + n] = g.clone()
        idle = "flow_head." if m.decoder.mode == "arm" else "flow_head_arm."
        assert all(named[idle + n].grad is None for n in RESNET_ORDER), "the inactive head received a gradient"
        assert all(p.grad is None for n, p in m.named_parameters() if "flow_head" not in n)
        return out

    def rel(a, b):
        a, b = a.double(), b.double()
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    arrays = dict(image=image, ctxt_c2w=ctx_c2w, ctxt_k_norm=Kn, trgt_c2w=trg_c2w, trgt_k_pix=kpix, origins=ro, directions=rd,
                  z_near=z_near, z_far=z_far, action=action, target=target)
    runs = {}
    for mode in ("regular", "arm"):
        model.decoder.switch_mode(mode)
        r32 = evaluate(model, torch.float32)
        m64 = copy.deepcopy(model).double()
        m64.decoder.switch_mode(mode)
        r64 = evaluate(m64, torch.float64, fixed_positions=r32["final_positions"])
        runs[mode] = r32
        if mode == "regular":
            arrays["features"] = r32["features"]
        for k, v in r32.items():
            if k == "features":
                continue
            arrays[f"{mode}.{k}"] = v
            if k.startswith("grad."):   # (1.5 MB of gradients per mode and precision: the float64 run is kept as its distance)
                arrays[f"{mode}.floor64.{k[5:]}"] = np.float64(rel(v, r64[k]))
            else:
                arrays[f"{mode}.{k}_f64"] = r64[k]
        print(f"  {mode}: loss {float(r32['loss']):.6g}; |flow| max {float(r32['optical_flow'].abs().max()):.4g}; "
              f"grad fp32-vs-fp64 max rel {max(rel(r32['grad.' + n], r64['grad.' + n]) for n in RESNET_ORDER):.3g}")
    assert rel(runs["regular"]["optical_flow"], runs["arm"]["optical_flow"]) > 1e-2, "arm and regular heads give the same flow"
    arrays["arm_keys"] = np.array(sorted(k for k in model.state_dict() if "flow_head_arm" in k))
    save("model_flow_train", **arrays)


if __name__ == "__main__":
    main()
