#!/usr/bin/env python3
"""Round-2 golden vectors, produced by importing the *reference* (read-only, /root/reference) like make_golden.py:

  model_{mlp,transformer,flow}_f64.npz   the reference Model evaluated in FLOAT64 (model.double(), same seeds, same inputs)
                                         next to the fp32 fixtures: |ref32 - ref64| is the reference's own fp32 rounding
                                         noise per output, the floor every parity bound looser than 1e-4 must cite
  model_mlp2.npz                         two proposal levels (num_proposal_samples = [16, 12]): the level loop of
                                         ProposalNetworkSampler.generate_ray_samples (rendering/ray_samplers.py:497-552)
  model_transformer8.npz                 jacobian_transformer head with A = 8 (all eight key slots of a head in use)
  wrapper.npz                            the reference's OWN ModelWrapper (models/model_wrapper.py imported with shims
                                         for Lightning / wandb / cv2 / the config package): prepare_training_input_output
                                         on both branches (random pixels :458-477, tracked pixels :479-507) and
                                         training_step's individual loss terms (:117-163) on the reference's model output

Usage:  python tests/golden/make_golden_r02.py     (build container only: the reference never travels)
"""
import copy
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (shims + helpers; also puts the repo root and oracle/ on sys.path)

import njf_oracle as orc  # noqa: E402

save, rigid, randn, rand, k_norm, load_seeded = mg.save, mg.rigid, mg.randn, mg.rand, mg.k_norm, mg.load_seeded


ENC_NOISE = 1e-6


def f64(t):
    return t.double() if isinstance(t, torch.Tensor) and t.is_floating_point() else t


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--harness-cases", type=lambda t: [int(x) for x in t.split(",")], default=None,
                    help="only (re)generate these PARITY_CASES entries of harness_reference.npz, keeping every other array of "
                         "the committed file (round 3: the full-size C2 / C3 / C5 rows, 9,10,11)")
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--arm-only", action="store_true", help="only (re)generate model_arm.npz (round 4)")
    args = ap.parse_args()
    mg.install_shims()
    torch.set_num_threads(args.threads)
    from neural_jacobian_field.rendering import geometry
    from neural_jacobian_field.model_components.resnet_fc import MlpCfg
    from neural_jacobian_field.models import model as ref_model
    from neural_jacobian_field.models.decoder import (ActionDecoderFlowMlpCfg, ActionDecoderJacobianMlpCfg,
                                                       ActionDecoderJacobianTransformerCfg, DensityDecoderMlpCfg)
    from neural_jacobian_field.models.decoder.action_decoder import PixelEncoding
    from neural_jacobian_field.models.decoder.action_decoder_jacobian import TransformerCfg
    from neural_jacobian_field.models.encoder import EncoderResnetCfg
    from neural_jacobian_field.utils import convention

    mlp_cfg = MlpCfg(n_blocks=5, d_hidden=128, combine_layer=3, combine_type="mean", beta=0.0)
    enc_cfg = EncoderResnetCfg(name="resnet", upsample_interp="bilinear", num_layers=4, use_first_pool=True, norm_type="batch")
    dens_cfg = DensityDecoderMlpCfg(name="density_mlp", mlp=mlp_cfg)
    mlp_dec = ActionDecoderJacobianMlpCfg(name="jacobian_mlp", mlp=mlp_cfg)
    tr_dec = ActionDecoderJacobianTransformerCfg(
        name="jacobian_transformer", mlp=mlp_cfg,
        transformer=TransformerCfg(attn_feat_dim=64, attn_head_dim=64, num_attn_heads=8, attn_depth=3, attn_mlp_dim=64))
    flow_dec = ActionDecoderFlowMlpCfg(name="flow_mlp", mlp=mlp_cfg)

    def build(dec_cfg, action_dim, n_prop, n_nerf):
        rcfg = ref_model.RenderingCfg(num_proposal_samples=tuple(n_prop), num_nerf_samples=n_nerf, single_jitter=False,
                                      proposal_warmup=5000, proposal_update_every=5, use_proposal_weight_anneal=True,
                                      proposal_weights_anneal_max_num_iters=1000, proposal_weights_anneal_slope=10.0)
        cfg = ref_model.ModelCfg(action_dim=action_dim, rendering=rcfg, encoder=enc_cfg, density_decoder=dens_cfg,
                                 action_decoder=dec_cfg)
        m = ref_model.Model(cfg)
        load_seeded(m, "", seed=0)
        return m.eval()

    if args.harness_cases is not None:
        harness_reference(build, mlp_dec, ref_model, PixelEncoding, only=args.harness_cases)
        return

    # ---- the scene of make_golden.py (same seeds) ----------------------------------------------------------------------
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

    def inputs(action, dtype=torch.float32):
        c = lambda t: t.to(dtype)
        cam = ref_model.CameraInput(input_image=c(image), ctxt_extrinsics=c(ctx_c2w), ctxt_intrinsics=c(Kn),
                                    trgt_extrinsics=c(trg_c2w), trgt_intrinsics=c(kpix))
        rin = ref_model.RenderingInput(origins=c(ro), directions=c(rd), z_near=c(z_near), z_far=c(z_far))
        return cam, rin, ref_model.RobotInput(robot_action=c(action))

    def evaluate(model, action, dtype, fixed_positions=None, with_inference=True, perturb=None):
        """Every output the GPU tests compare, in `dtype`.  `fixed_positions` = (final_positions, prop_positions) of the
        fp32 run: the per-sample decoder outputs are then evaluated at IDENTICAL sample locations in both precisions.
        `perturb` = ("ulp", seed): ray origins / directions moved by one ulp at random; ("enc", seed): encoder features
        scaled by (1 + ENC_NOISE N(0,1)), ENC_NOISE = 1e-6 -- MIOpen's convolutions measure 6.6e-7 norm-wise against the
        reference's encoder output (margins row model_mlp.encoder / features); round 2 assumed 1e-5 here."""
        cam, rin, rob = inputs(action, dtype)
        enc_forward = model.encoder.forward
        if perturb is not None:
            gen = torch.Generator().manual_seed(perturb[1])
            if perturb[0] == "ulp":
                nudge = lambda t: torch.where(torch.rand(t.shape, generator=gen) < 0.5, torch.nextafter(t, t + 1), torch.nextafter(t, t - 1))
                rin = ref_model.RenderingInput(origins=nudge(rin.origins), directions=nudge(rin.directions), z_near=rin.z_near, z_far=rin.z_far)
            else:
                model.encoder.forward = lambda img: (lambda f: f * (1 + ENC_NOISE * torch.randn(f.shape, generator=gen)))(enc_forward(img))
        try:
            return _evaluate(model, cam, rin, rob, dtype, fixed_positions, with_inference)
        finally:
            model.encoder.forward = enc_forward

    def self_noise(model, action, base, keys):
        """How far the reference's OWN fp32 outputs move under perturbations no fp32 implementation can avoid: one ulp on
        the rays (sample placement feeds a positional encoding with a 2*pi*512 gain) and 1e-5 on the encoder features."""
        out = {}
        for kind, seeds in (("ulp", (1, 2, 3, 4)), ("enc", (5, 6))):
            runs = [evaluate(model, action, torch.float32, with_inference=False, perturb=(kind, sd)) for sd in seeds]
            for k in keys:
                if k in base:
                    out[f"floor_{kind}.{k}"] = np.float64(max(rel(r[k], base[k]) for r in runs))
        return out

    def rel(a, b):
        a, b = a.double(), b.double()
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    END_TO_END = ("rgb", "depth", "optical_flow", "prop_weights", "prop_weights0", "prop_weights1", "vis_action_features", "vis_steps",
                  "vis_weights", "vis_ray_positions", "vis_ray_positions_warped", "final_starts", "final_ends", "prop_starts1")

    def _evaluate(model, cam, rin, rob, dtype, fixed_positions, with_inference):
        out = {}
        with torch.no_grad():
            feats = model.encoder.forward(cam.input_image)
            res = model.forward(cam, rin, rob, compute_vis_features=model.cfg.action_decoder.name != "flow_mlp")
            penc = PixelEncoding(features=feats, extrinsics=cam.ctxt_extrinsics, intrinsics=cam.ctxt_intrinsics, action=rob.robot_action)
            rb = model.compute_ray_bundle(rin)
            samples, pos, dirs, wl, sl = model.compute_proposal(rb, penc)
            out.update(features=feats, rgb=res.standard_output.rgb, depth=res.standard_output.depth,
                       optical_flow=res.standard_output.optical_flow, final_positions=pos,
                       final_starts=samples.starts, final_ends=samples.ends)
            for i, (wts, smp) in enumerate(zip(wl, sl)):
                out[f"prop_weights{i}"] = wts
                out[f"prop_starts{i}"] = smp.starts
                out[f"prop_ends{i}"] = smp.ends
            out["prop_weights"] = wl[0]
            if res.vis_output is not None:
                out.update(vis_action_features=res.vis_output.action_features, vis_steps=res.vis_output.steps,
                           vis_weights=res.vis_output.weights, vis_ray_positions=res.vis_output.ray_positions,
                           vis_ray_positions_warped=res.vis_output.ray_positions_warped)
            p_fin = pos if fixed_positions is None else fixed_positions[0].to(dtype)
            p_prop = sl[0].get_positions() if fixed_positions is None else fixed_positions[1].to(dtype)
            d_fin = rin.directions[..., None, :].expand(p_fin.shape)
            dec = model.decoder.forward(world_space_xyz=p_fin, world_space_dir=d_fin, pixel_encoding=penc)
            out.update(dec_density=dec.density, dec_color=dec.color, dec_flow=dec.flow)
            if dec.action_features is not None:
                out["dec_action_features"] = dec.action_features
            out["prop_density"] = model.proposal_networks[0].get_density(p_prop, penc)
            out["prop_positions"] = p_prop
            if with_inference and "jacobian" in model.cfg.action_decoder.name:
                enc_out = model.encode_image(cam, rin, rob)
                out.update(enc_density=enc_out.density, enc_action_features=enc_out.action_features, enc_weights=enc_out.weights)
                if fixed_positions is not None:  # decoder.encode_image at the fp32 run's positions
                    fo = model.decoder.encode_image(p_fin, penc)
                    out.update(encpos_density=fo.density, encpos_action_features=fo.action_features)
                out["infer_flow"] = model.infer_optical_flow(enc_out, cam, ref_model.RobotInput(robot_action=rob.robot_action * 2 + 0.05))
        return out

    def arm_fixture():
        """model_arm.npz (round 4): the second Jacobian head (use_arm_model, action_decoder_jacobian.py:306-313 / 400-407) of BOTH
        decoders in arm mode (switch_mode("arm"), :89-90, :330-331, :438-446), with arm_action_dim == action_dim -- the only
        case the reference's compute_flow (:134-140) supports.  Per decoder: Model.forward end to end, the decoder on the fp32
        run's sample positions, both in fp32 and float64; plus the regular-mode flow of the same weights (the two modes must
        differ: the test would otherwise pass on a head that ignores the switch)."""
        import dataclasses
        arrays = dict(image=image, ctxt_c2w=ctx_c2w, ctxt_k_norm=Kn, trgt_c2w=trg_c2w, trgt_k_pix=kpix, origins=ro, directions=rd,
                      z_near=z_near, z_far=z_far)
        for tag, cfg0, A in (("mlp", mlp_dec, 8), ("transformer", tr_dec, 6)):
            dec_cfg = dataclasses.replace(cfg0, use_arm_model=True, arm_action_dim=A)
            model = build(dec_cfg, A, [16], 12)
            assert any(k.startswith("decoder.jacobian_head_arm.") for k in model.state_dict())
            action = 0.1 * randn(26, B, A)
            regular = evaluate(model, action, torch.float32, with_inference=False)
            model.decoder.switch_mode("arm")
            r32 = evaluate(model, action, torch.float32, with_inference=False)
            m64 = copy.deepcopy(model).double()
            m64.decoder.switch_mode("arm")
            r64 = evaluate(m64, action, torch.float64, fixed_positions=(r32["final_positions"], r32["prop_positions"]), with_inference=False)
            assert rel(regular["optical_flow"], r32["optical_flow"]) > 1e-2, "arm and regular heads give the same flow"
            arrays[f"{tag}.action"] = action
            arrays[f"{tag}.final_positions"] = r32["final_positions"]
            arrays[f"{tag}.regular_optical_flow"] = regular["optical_flow"]
            if tag == "mlp":
                arrays["features"] = r32["features"]
            for k in ("rgb", "depth", "optical_flow", "dec_action_features", "dec_flow", "dec_density", "vis_action_features"):
                arrays[f"{tag}.{k}"] = r32[k]
                arrays[f"{tag}.{k}_f64"] = r64[k]
            for k, v in self_noise(model, action, r32, ("rgb", "depth", "optical_flow", "vis_action_features")).items():
                arrays[f"{tag}.{k}"] = v
            arrays[f"{tag}.arm_keys"] = np.array(sorted(k for k in model.state_dict() if "jacobian_head_arm" in k))
        save("model_arm", **arrays)

    if args.arm_only:
        arm_fixture()
        return

    print("writing round-2 fixtures to", HERE)
    arm_fixture()
    committed = {t: dict(np.load(os.path.join(HERE, f"model_{t}.npz"))) for t in ("mlp", "transformer", "flow")}
    for tag, dec_cfg, A in (("mlp", mlp_dec, 8), ("transformer", tr_dec, 6), ("flow", flow_dec, 5)):
        model = build(dec_cfg, A, [16], 12)
        action = 0.1 * randn(25, B, A) * (5.0 if tag == "flow" else 1.0)
        r32 = evaluate(model, action, torch.float32)
        # the committed fp32 fixture and this script must describe the same run
        assert np.array_equal(r32["rgb"].numpy(), committed[tag]["rgb"]), tag
        assert np.array_equal(r32["final_positions"].numpy(), committed[tag]["final_positions"]), tag
        m64 = copy.deepcopy(model).double()
        r64 = evaluate(m64, action, torch.float64, fixed_positions=(r32["final_positions"], r32["prop_positions"]))
        keep = ["features", "rgb", "depth", "optical_flow", "prop_weights", "prop_density", "dec_density", "dec_color", "dec_flow",
                "dec_action_features", "vis_action_features", "vis_steps", "vis_ray_positions", "vis_ray_positions_warped",
                "vis_weights", "enc_weights", "encpos_density", "encpos_action_features", "infer_flow", "final_starts", "final_ends"]
        arrays = {k: r64[k] for k in keep if k in r64}
        arrays.update(self_noise(model, action, r32, END_TO_END))
        if tag == "mlp":   # enc_weights: per-sample weights at independently placed samples (encode_image)
            pert = [evaluate(model, action, torch.float32, perturb=("ulp", sd)) for sd in (1, 2)]
            arrays["floor_ulp.enc_weights"] = np.float64(max(rel(q["enc_weights"], r32["enc_weights"]) for q in pert))
        if tag != "mlp":
            arrays.pop("features", None)  # the encoder is shared: its fp64 output is kept once (model_mlp_f64)
        if tag == "flow":  # the zero-action run of the flow_mlp fixture; its 640 hidden "action features" are read by nothing
            arrays.pop("dec_action_features", None)
            z64 = evaluate(m64, torch.zeros_like(action), torch.float64, with_inference=False)
            arrays["optical_flow_zero_action"] = z64["optical_flow"]
        save(f"model_{tag}_f64", **arrays)

    # ---- two proposal levels ------------------------------------------------------------------------------------------
    model = build(mlp_dec, 8, [16, 12], 10)
    action = 0.1 * randn(25, B, 8)
    r32 = evaluate(model, action, torch.float32, with_inference=False)
    r64 = evaluate(copy.deepcopy(model).double(), action, torch.float64, with_inference=False)
    arrays = dict(image=image, ctxt_c2w=ctx_c2w, ctxt_k_norm=Kn, trgt_c2w=trg_c2w, trgt_k_pix=kpix, origins=ro, directions=rd,
                  z_near=z_near, z_far=z_far, action=action)
    for k in ("features", "rgb", "depth", "optical_flow", "prop_weights0", "prop_weights1", "prop_starts0", "prop_ends0",
              "prop_starts1", "prop_ends1", "final_starts", "final_ends", "vis_action_features", "vis_weights"):
        arrays[k] = r32[k]
        arrays[k + "_f64"] = r64[k]
    arrays.update(self_noise(model, action, r32, END_TO_END))
    save("model_mlp2", **arrays)

    # ---- transformer head with all eight key slots ----------------------------------------------------------------------
    model = build(tr_dec, 8, [16], 12)
    action = 0.1 * randn(25, B, 8)
    r32 = evaluate(model, action, torch.float32, with_inference=False)
    r64 = evaluate(copy.deepcopy(model).double(), action, torch.float64,
                   fixed_positions=(r32["final_positions"], r32["prop_positions"]), with_inference=False)
    arrays = dict(image=image, ctxt_c2w=ctx_c2w, ctxt_k_norm=Kn, trgt_c2w=trg_c2w, trgt_k_pix=kpix, origins=ro, directions=rd,
                  z_near=z_near, z_far=z_far, action=action, final_positions=r32["final_positions"])
    for k in ("features", "rgb", "depth", "optical_flow", "dec_action_features", "dec_flow", "dec_density", "vis_action_features"):
        arrays[k] = r32[k]
        arrays[k + "_f64"] = r64[k]
    arrays.update(self_noise(model, action, r32, END_TO_END))
    save("model_transformer8", **arrays)

    # ---- Model.patch_render + the Jacobian-field colouring on its output (models/model.py:527-628) ----------------------
    mg._module("cv2")
    mg._module("matplotlib")
    mg._module("matplotlib.pyplot")
    from neural_jacobian_field.inference import jacobian_color_map as ref_cm
    model = build(mlp_dec, 8, [16], 12)
    action = 0.1 * randn(25, B, 8)
    arrays = {}
    for dtype, suf in ((torch.float32, ""), (torch.float64, "_f64")):
        mdl = model if dtype == torch.float32 else copy.deepcopy(model).double()
        cam, rin, rob = inputs(action, dtype)
        frame = mdl.patch_render(cam, rin, rob, patch_size=8, render_height=4, render_width=5)   # 20 rays in 3 patches
        sens0 = ref_cm.compute_joint_sensitivity(frame.action_features, None, mode=0)
        sens1 = ref_cm.compute_joint_sensitivity(frame.action_features, cam.trgt_extrinsics[:, None, None, None], mode=1)
        cmap = torch.tensor(ref_cm.JACOBIAN_COLORMAP["model_allegro"]).t().contiguous().to(dtype)
        for k in ("rgb", "depth_raw", "flow_raw", "ray_positions", "ray_positions_warped", "action_features", "steps", "weights"):
            arrays[k + suf] = getattr(frame, k)
        arrays["sensitivity_mode0" + suf] = sens0
        arrays["sensitivity_mode1" + suf] = sens1
        if dtype == torch.float32:
            arrays["sensitivity_image"] = ref_cm.visualize_joint_sensitivity(sens0, cmap)
            arrays["color_map"] = cmap
    save("patch_render", **arrays)

    wrapper_fixture(build, mlp_dec, ref_model)
    # the small parity cases; the full-size rows (C2 / C3 / C5, minutes each) are produced by
    #   python tests/golden/make_golden_r02.py --harness-cases 9,10,11 --threads 8
    import parity_harness as ph
    harness_reference(build, mlp_dec, ref_model, PixelEncoding, only=[i for i in range(len(ph.PARITY_CASES)) if i not in ph.FULL_SIZE_CASES])


def harness_reference(build, mlp_dec, ref_model, PixelEncoding, only=None):
    """The reference itself on every parity-suite case (oracle/parity_harness.py: PARITY_CASES): its fp32 end-to-end
    outputs, its final spacing bins, and per compared quantity the floor max|ref32 - ref64| / max|ref64| -- end to end
    for rgb / depth / optical_flow / proposal weights / bins, and for the per-sample quantities with the float64 decoder
    evaluated AT THE fp32 RUN'S SAMPLE LOCATIONS (what the harness's "s_*" comparisons do on the HIP side)."""
    import dataclasses
    import parity_harness as ph

    def rel(a, b):
        a, b = a.double(), b.double()
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    def run(model, case, dtype, anneal, samples_from=None):
        c = lambda t: t.to(dtype)
        cams = case["cams"]
        feats = c(case["feats"])
        model.encoder.forward = lambda img: feats
        model.proposal_sampler.set_anneal(anneal)
        b = feats.shape[0]
        cam = ref_model.CameraInput(input_image=torch.zeros(b, 3, 8, 8, dtype=dtype), ctxt_extrinsics=c(cams["ctxt_c2w"]),
                                    ctxt_intrinsics=c(cams["ctxt_k_norm"]), trgt_extrinsics=c(cams["trgt_c2w"]),
                                    trgt_intrinsics=c(case["k_pix"]))
        rin = ref_model.RenderingInput(origins=c(case["origins"]), directions=c(case["directions"]), z_near=c(cams["z_near"]),
                                       z_far=c(cams["z_far"]))
        rob = ref_model.RobotInput(robot_action=c(case["action"]))
        with torch.no_grad():
            penc = PixelEncoding(features=feats, extrinsics=cam.ctxt_extrinsics, intrinsics=cam.ctxt_intrinsics, action=rob.robot_action)
            rb = model.compute_ray_bundle(rin)
            if samples_from is None:
                samples, pos, dirs, wl, sl = model.compute_proposal(rb, penc)
            else:  # the fp32 run's final samples, field by field in this dtype
                samples = dataclasses.replace(samples_from, **{f.name: c(getattr(samples_from, f.name)) for f in
                                                               dataclasses.fields(samples_from)
                                                               if torch.is_tensor(getattr(samples_from, f.name))})
                pos = samples.get_positions()
                dirs = rin.directions[..., None, :].expand(pos.shape)
                wl = None
            dec = model.decoder.forward(world_space_xyz=pos, world_space_dir=dirs, pixel_encoding=penc)
            w = samples.get_weights(dec.density)
            rgb = model.render_rgb(rgb=dec.color, weights=w, bg_color=None)
            depth, _ = model.render_depth(weights=w, ray_samples=samples)
            flow, rp, rpw = model.render_optical_flow(weights=w, ray_positions=pos, scene_flow=dec.flow[..., :3],
                                                      trgt_extrinsics=cam.trgt_extrinsics, trgt_intrinsics=cam.trgt_intrinsics)
            af = model.render_action_features(dec.action_features, w)
        bins = torch.cat([samples.spacing_starts[..., 0], samples.spacing_ends[..., -1:, 0]], -1)
        return dict(rgb=rgb, depth=depth, optical_flow=flow, prop_weights=None if wl is None else wl[0], bins=bins, samples=samples,
                    weights=w, density=dec.density, color=dec.color, sample_flow=dec.flow[..., :3], jacobian=dec.action_features,
                    action_features=af, pos=rp, pos_warped=rpw)

    arrays = {}
    if only is not None:
        with np.load(os.path.join(HERE, "harness_reference.npz")) as f:
            arrays = {k: f[k] for k in f.files if not any(k.startswith(f"c{i}.") for i in only)}
    for i, cfg in enumerate(ph.PARITY_CASES):
        if only is not None and i not in only:
            continue
        cfg = {**ph.CASE_DEFAULTS, **cfg}
        case = ph.make_case(cfg["batch"], cfg["height"], cfg["width"], cfg["rays"], cfg["action_dim"], cfg["seed"], cfg["identity_context"])
        model = build(mlp_dec, cfg["action_dim"], [cfg["s_prop"]], cfg["s_final"])
        sd = model.state_dict()
        assert all(k in sd for k in case["params"])
        sd.update(case["params"])
        model.load_state_dict(sd, strict=True)
        r32 = run(model, case, torch.float32, cfg["anneal"])
        m64 = copy.deepcopy(model).double()
        r64 = run(m64, case, torch.float64, cfg["anneal"])
        s64 = run(m64, case, torch.float64, cfg["anneal"], samples_from=r32["samples"])
        # the oracle must agree with the reference on these cases too (it is the harness's per-sample comparator)
        ora = ph.oracle_forward(case, cfg["s_prop"], cfg["s_final"], cfg["anneal"])
        assert rel(ora.rgb, r32["rgb"]) < 1e-5 and rel(ora.depth, r32["depth"]) < 1e-5, (i, rel(ora.rgb, r32["rgb"]))
        pre = f"c{i}."
        for k in ("rgb", "depth", "optical_flow", "bins"):
            arrays[pre + k] = r32[k]
        # round 4 (VERDICT r03 "next" #3): the float64 run's end-to-end outputs THEMSELVES, so that the GPU tests can hold the
        # HIP path to the truth element by element: e_hip = |hip - ref64| against e_ref = |ref32 - ref64|
        for k in ("rgb", "depth", "optical_flow", "bins"):
            arrays[pre + k + "64"] = r64[k]
        # the exact inputs of this run: derived quantities (normalised directions, matrix exponentials, inverses) can differ
        # by an ulp between CPUs, and the positional encoding turns one ulp of a ray direction into ~1e-4 of depth -- the
        # harness feeds THESE tensors to the oracle and to the HIP path on the GPU box
        cams = case["cams"]
        for k, v in (("in.origins", case["origins"]), ("in.directions", case["directions"]), ("in.k_pix", case["k_pix"]),
                     ("in.action", case["action"]), ("in.ctxt_c2w", cams["ctxt_c2w"]), ("in.trgt_c2w", cams["trgt_c2w"]),
                     ("in.ctxt_k_norm", cams["ctxt_k_norm"]), ("in.z_near", cams["z_near"]), ("in.z_far", cams["z_far"]),
                     ("in.ctxt_w2c", torch.inverse(cams["ctxt_c2w"])), ("in.trgt_w2c", torch.inverse(cams["trgt_c2w"]))):
            arrays[pre + k] = v
        # seeded tensors regenerated on the box (too large to commit): checksums guard the regeneration
        arrays[pre + "sum.feats"] = np.float64(case["feats"].double().sum().item())
        arrays[pre + "sum.params"] = np.float64(sum(v.double().abs().sum().item() for v in case["params"].values()))
        floor = {k: rel(r32[k], r64[k]) for k in ("rgb", "depth", "optical_flow", "prop_weights")}
        floor["final_bins"] = rel(r32["bins"], r64["bins"])
        for hk, rk in (("s_rgb", "rgb"), ("s_depth", "depth"), ("s_optical_flow", "optical_flow"), ("s_weights", "weights"),
                       ("s_density", "density"), ("s_color", "color"), ("s_sample_flow", "sample_flow"), ("s_jacobian", "jacobian"),
                       ("s_action_features", "action_features"), ("s_pos", "pos"), ("s_pos_warped", "pos_warped")):
            floor[hk] = rel(r32[rk], s64[rk])
        for k in ph.FLOOR_KEYS:
            arrays[pre + "floor." + k] = np.float64(floor[k])
        if i in ph.FULL_SIZE_CASES:
            # self-noise of the reference's end-to-end outputs (same definition as the floor_ulp.* arrays of the model_*
            # fixtures): its fp32 outputs under a one-ulp perturbation of the rays, two seeds
            for sd in (1, 2):
                gen = torch.Generator().manual_seed(sd)
                nudge = lambda t: torch.where(torch.rand(t.shape, generator=gen) < 0.5, torch.nextafter(t, t + 1), torch.nextafter(t, t - 1))
                moved = run(model, dict(case, origins=nudge(case["origins"]), directions=nudge(case["directions"])), torch.float32,
                            cfg["anneal"])
                for k, rk in (("rgb", "rgb"), ("depth", "depth"), ("optical_flow", "optical_flow"), ("prop_weights", "prop_weights"),
                              ("final_bins", "bins")):
                    key = pre + "floor_ulp." + k
                    arrays[key] = np.float64(max(float(arrays.get(key, 0.0)), rel(moved[rk], r32[rk])))
            print("           one-ulp rays: " + " ".join(f"{k}={float(arrays[pre + 'floor_ulp.' + k]):.1e}"
                                                         for k in ("rgb", "depth", "optical_flow", "prop_weights", "final_bins")))
        print(f"  case {i}: floors " + " ".join(f"{k}={floor[k]:.1e}" for k in ("rgb", "depth", "optical_flow", "s_density", "s_jacobian")))
    save("harness_reference", **arrays)


def wrapper_fixture(build, mlp_dec, ref_model):
    """The reference's own ModelWrapper.prepare_training_input_output / training_step, imported for real."""
    m = mg._module
    m("cv2")
    m("wandb", Image=lambda *a, **k: None, log=lambda *a, **k: None)
    m("lightning_fabric")
    m("lightning_fabric.utilities")
    m("lightning_fabric.utilities.apply_func", apply_to_collection=lambda *a, **k: None)
    m("nerfstudio.model_components")
    # third-party, absent and un-pinned: the oracle's restatement stands in (parity unpinned for these two terms)
    sdist = lambda s: torch.cat([s.spacing_starts[..., 0], s.spacing_ends[..., -1:, 0]], dim=-1).flatten(0, -2)
    flat = lambda w: w[..., 0].flatten(0, -2)
    m("nerfstudio.model_components.losses",
      interlevel_loss=lambda wl, sl: orc.interlevel_loss([flat(w) for w in wl], [sdist(s) for s in sl]),
      distortion_loss=lambda wl, sl: orc.distortion_loss(flat(wl[-1]), sdist(sl[-1])))
    m("pytorch_lightning", LightningModule=nn.Module)
    m("pytorch_lightning.utilities")
    m("pytorch_lightning.utilities.rank_zero", rank_zero_only=lambda f: f)

    class _Blank:
        def __getattr__(self, k):
            return ""

    m("colorama", Fore=_Blank(), Style=_Blank())
    import neural_jacobian_field
    cfgpkg = types.ModuleType("neural_jacobian_field.config")
    cfgpkg.__path__ = []
    sys.modules["neural_jacobian_field.config"] = cfgpkg
    neural_jacobian_field.config = cfgpkg
    common = types.ModuleType("neural_jacobian_field.config.common")
    common.PipelineCfg = object
    sys.modules["neural_jacobian_field.config.common"] = common
    cfgpkg.common = common
    from neural_jacobian_field.models import model_wrapper as mw

    NS = types.SimpleNamespace
    B, H, W, A, RAYS = 2, 12, 16, 8, 24

    def make_batch(seed, tracked):
        g = lambda k, *shape: rand(seed + k, *shape)
        from neural_jacobian_field.rendering import geometry
        coords, _ = geometry.get_pixel_coordinates(H, W)
        batch = {
            "context": {"rgb": g(1, B, 3, H, W), "extrinsics": torch.eye(4)[None].repeat(B, 1, 1),
                        "intrinsics": k_norm(B), "robot_action": 0.1 * randn(seed + 2, B, A)},
            "target": {"rgb": g(3, B, 3, H, W), "depth": g(4, B, 1, H, W) * 4 + 0.5, "extrinsics": rigid(seed + 5, B),
                       "intrinsics": k_norm(B)},
            "scene": {"near": torch.tensor([0.5, 0.4]), "far": torch.tensor([10.0, 6.0]),
                      "coordinates": coords[None].repeat(B, 1, 1, 1)},
        }
        batch["target"]["depth"][0, 0, 3, 5] = 0.0  # a pixel without depth (masked by ds_nerf_depth_loss)
        if tracked:
            n = 20
            batch["target"]["pixel_selector"] = torch.randint(0, H * W, (B, n), generator=torch.Generator().manual_seed(seed + 6))
            batch["target"]["pixel_motion"] = randn(seed + 7, B, n, 2) * 2
            batch["target"]["pixel_visible_mask"] = (rand(seed + 8, B, n) > 0.3).float()
        else:
            batch["target"]["flow"] = randn(seed + 9, B, 2, H, W) * 2
        return batch

    def clone(batch):
        return {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in batch.items()}

    arrays = {}
    for case, mode, tracked in (("perception", "perception", False), ("action_dense", "action", False), ("action_tracked", "action", True)):
        model = build(mlp_dec, A, [16], 12)
        cfg = NS(dataset=NS(mode=mode), training=NS(data=NS(rays_per_batch=RAYS)), wandb=NS(mode="disabled"))
        wrapper = mw.ModelWrapper(cfg, model)
        logs = {}
        wrapper.log = lambda k, v, logs=logs: logs.__setitem__(k, v.detach().clone() if torch.is_tensor(v) else torch.tensor(float(v)))
        batch = make_batch({"perception": 300, "action_dense": 400, "action_tracked": 500}[case], tracked)
        for part in ("context", "target", "scene"):
            for k, v in batch[part].items():
                arrays[f"{case}.batch.{part}.{k}"] = v.clone()
        # (1) the packing alone
        torch.manual_seed(77)
        mi, mt = wrapper.prepare_training_input_output(clone(batch))
        arrays.update({f"{case}.origins": mi.rendering_input.origins, f"{case}.directions": mi.rendering_input.directions,
                       f"{case}.z_near": mi.rendering_input.z_near, f"{case}.z_far": mi.rendering_input.z_far,
                       f"{case}.trgt_intrinsics": mi.camera_input.trgt_intrinsics, f"{case}.target_rgb": mt.rgb,
                       f"{case}.target_depth": mt.depth})
        if mt.optical_flow is not None:
            arrays[f"{case}.target_flow"] = mt.optical_flow
        if mt.visible_mask is not None:
            arrays[f"{case}.target_mask"] = mt.visible_mask
        # (2) the loss terms of training_step on the reference's own model output (train mode: stratified jitter from the
        # global RNG; the GPU test feeds THESE outputs to the product's loss functions, so the jitter needs no replay)
        captured = {}
        fwd = model.forward
        model.forward = lambda *a, **k: captured.setdefault("out", fwd(*a, **k))
        wrapper.train()
        model.encoder.eval()
        model.step_before_iter(300)
        torch.manual_seed(77)
        with torch.no_grad():
            total = wrapper.training_step(clone(batch), 0)
        out = captured["out"]
        arrays[f"{case}.loss_total"] = total
        for k, v in logs.items():
            if k.startswith("loss/"):
                arrays[f"{case}.{k}"] = v
        arrays.update({f"{case}.out_rgb": out.standard_output.rgb, f"{case}.out_depth": out.standard_output.depth,
                       f"{case}.out_flow": out.standard_output.optical_flow})
        for i, (wts, smp) in enumerate(zip(out.training_output.weights_list, out.training_output.ray_samples_list)):
            arrays.update({f"{case}.w{i}": wts, f"{case}.starts{i}": smp.starts, f"{case}.ends{i}": smp.ends,
                           f"{case}.sp0_{i}": smp.spacing_starts, f"{case}.sp1_{i}": smp.spacing_ends})
    save("wrapper", **arrays)


if __name__ == "__main__":
    main()
