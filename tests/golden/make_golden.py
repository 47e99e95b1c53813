#!/usr/bin/env python3
"""Generate golden vectors by importing the *reference* (read-only, /root/reference).

Runs only in the build container (the reference never travels to the GPU box); its outputs --
small ``.npz``/``.json`` fixtures next to this file -- are committed and are what pins
``oracle/njf_oracle.py`` (tests/test_oracle_golden.py) and, through it, the HIP path.

Usage:  python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The reference depends on packages that are not installed here (jaxtyping, omegaconf, nerfstudio,
tinycudann, torchvision, lightning ...).  They are replaced by the minimal shims below.  Three of
the shims contain arithmetic (NeRFEncoding, SHEncoding, resnet34): those are the un-vendored,
un-pinned third-party pieces, restated in oracle/njf_oracle.py -- fixtures that pass through them
pin only the reference's *use* of them ("parity unpinned").

Weights are never stored: they are regenerated from (seed, parameter name) by
``neural_jacobian_field_amd.synthetic.seeded_state_dict`` on both sides.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/project"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import njf_oracle as orc  # noqa: E402  (third-party restatements reused by the shims)
from neural_jacobian_field_amd import synthetic  # noqa: E402


# --------------------------------------------------------------------------------------
# shims for absent third-party packages
# --------------------------------------------------------------------------------------
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            _module(parent)
        setattr(sys.modules[parent], leaf, m)
    return m


class _Subscriptable:
    def __class_getitem__(cls, item):
        return cls


class _NeRFEncoding(nn.Module):
    def __init__(self, in_dim, num_frequencies, min_freq_exp, max_freq_exp, include_input=False,
                 implementation="torch"):
        super().__init__()
        assert min_freq_exp == 0 and max_freq_exp == num_frequencies - 1 and include_input
        self.in_dim, self.num_frequencies = in_dim, num_frequencies

    def get_out_dim(self):
        return self.in_dim * self.num_frequencies * 2 + self.in_dim

    def forward(self, x):
        return orc.nerf_positional_encoding(x, self.num_frequencies)


class _SHEncoding(nn.Module):
    def __init__(self, levels=4, implementation="tcnn"):
        super().__init__()
        assert levels == 4

    def get_out_dim(self):
        return 16

    def forward(self, x):
        return orc.sh4_encoding(x)


class _BasicBlock(nn.Module):
    def __init__(self, cin, planes, stride, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = None
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), norm_layer(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _ResNet34(nn.Module):
    def __init__(self, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (planes, blocks) in enumerate(zip([64, 128, 256, 512], [3, 4, 6, 3]), start=1):
            layers = []
            for bi in range(blocks):
                layers.append(_BasicBlock(cin, planes, 2 if (bi == 0 and li > 1) else 1, norm_layer))
                cin = planes
            setattr(self, f"layer{li}", nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def install_shims():
    _module("jaxtyping", **{k: _Subscriptable for k in ["Float", "Int", "Int64", "Bool", "UInt8", "Shaped"]})
    _module("omegaconf", DictConfig=dict)
    _module("nerfstudio")
    _module("nerfstudio.cameras.camera_utils", normalize_with_norm=None)
    _module("nerfstudio.utils.colormaps", apply_depth_colormap=lambda x, **k: x)
    _module("nerfstudio.field_components.encodings", NeRFEncoding=_NeRFEncoding, SHEncoding=_SHEncoding)
    _module("torchvision")
    _module("torchvision.models", resnet34=lambda pretrained=False, norm_layer=None: _ResNet34(norm_layer))
    _module("torchvision.utils", flow_to_image=lambda x: x)
    sys.path.insert(0, REF)


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(out)} arrays)")


def rigid(seed, batch):
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(batch, 3, 3, generator=g))
    q = q * torch.sign(torch.linalg.det(q))[:, None, None]
    m = torch.eye(4)[None].repeat(batch, 1, 1)
    m[:, :3, :3] = torch.matrix_exp(0.15 * (q - q.transpose(1, 2)))
    m[:, :3, 3] = 0.2 * torch.randn(batch, 3, generator=g)
    return m.contiguous()


def randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def rand(seed, *shape):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed))


def k_norm(batch):
    k = torch.tensor([[0.8, 0.0, 0.5], [0.0, 0.9, 0.48], [0.0, 0.0, 1.0]])
    return k[None].repeat(batch, 1, 1).contiguous()


def load_seeded(module, prefix, seed=0):
    """Overwrite every parameter/buffer of ``module`` from (seed, prefix + name)."""
    sd = module.state_dict()
    new = {k: synthetic.seeded_tensor(prefix + k, tuple(v.shape), seed) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return {k: tuple(v.shape) for k, v in sd.items()}


def main():
    install_shims()
    torch.set_num_threads(1)  # bit-stable reductions
    from neural_jacobian_field.rendering import geometry, ray_samplers
    from neural_jacobian_field.model_components import activations, pixel_aligned_features
    from neural_jacobian_field.model_components.resnet_fc import MlpCfg, ResnetFC
    from neural_jacobian_field.models import model as ref_model
    from neural_jacobian_field.models.decoder import (ActionDecoderFlowMlpCfg, ActionDecoderJacobianMlpCfg,
                                                       ActionDecoderJacobianTransformerCfg, DensityDecoderMlpCfg)
    from neural_jacobian_field.models.decoder.action_decoder import PixelEncoding
    from neural_jacobian_field.models.decoder.action_decoder_jacobian import TransformerCfg
    from neural_jacobian_field.models.encoder import EncoderResnetCfg
    from neural_jacobian_field.utils import convention, loss_utils

    print("writing fixtures to", HERE)

    # ---------------- a1/a2/a18(projection)/a20: geometry --------------------------------------
    coords, selector = geometry.get_pixel_coordinates(5, 7)
    B = 2
    c2w = rigid(11, B)
    K = k_norm(B)
    xy = coords.reshape(1, -1, 2).repeat(B, 1, 1)
    o, d, z = geometry.get_world_rays_with_z(xy, K, c2w)
    k_pix = convention.denormalize_intrinsics(K, width=7, height=5)
    pts = randn(12, B, 9, 3) + torch.tensor([0.0, 0.0, 3.0])
    uv = geometry.project_world_coords_to_camera(pts, c2w, k_pix)
    save("geometry", coords=coords, selector=selector, c2w=c2w, k_norm=K, xy=xy, origins=o, directions=d, z=z,
         k_pix=k_pix, pts=pts, uv=uv)

    # ---------------- a4/a10/a11: samplers -----------------------------------------------------
    R = 6
    ob = ray_samplers.RayBundle(origins=o[:, :R], directions=d[:, :R], nears=torch.full((B, R, 1), 0.5),
                                fars=torch.full((B, R, 1), 10.0))
    uni = ray_samplers.UniformSampler(single_jitter=False)
    uni.eval()
    s_eval = uni(ob, num_samples=12)
    uni.train()
    torch.manual_seed(100)
    s_train = uni(ob, num_samples=12)
    uni1 = ray_samplers.UniformSampler(single_jitter=True)
    uni1.train()
    torch.manual_seed(101)
    s_train1 = uni1(ob, num_samples=12)
    dens = torch.exp(1.5 * randn(13, B, R, 12, 1))
    w = s_eval.get_weights(dens)
    # ragged deltas: zero and negative widths must contribute nothing (ray_samplers.py:84-88)
    rag = ray_samplers.RaySamples(origins=s_eval.origins, directions=s_eval.directions, starts=s_eval.starts,
                                  ends=s_eval.ends, deltas=s_eval.deltas.clone())
    rag.deltas[:, :, 3] = 0.0
    rag.deltas[:, :, 7] = -0.25
    w_rag = rag.get_weights(dens)
    pdf = ray_samplers.PDFSampler(include_original=False, single_jitter=False)
    pdf.eval()
    p_eval = pdf(ob, s_eval, w, num_samples=10)
    w_zero = torch.zeros_like(w)
    w_zero[0, 0, 5] = 1.0  # delta-like pdf + all-zero rays (padding path, ray_samplers.py:378-382)
    p_zero = pdf(ob, s_eval, w_zero, num_samples=10)
    pdf.train()
    torch.manual_seed(102)
    p_train = pdf(ob, s_train, s_train.get_weights(dens), num_samples=10)
    save("samplers", origins=ob.origins, directions=ob.directions, near=ob.nears, far=ob.fars,
         eval_starts=s_eval.starts, eval_ends=s_eval.ends, eval_sp0=s_eval.spacing_starts, eval_sp1=s_eval.spacing_ends,
         eval_pos=s_eval.get_positions(),
         train_starts=s_train.starts, train_ends=s_train.ends, train1_starts=s_train1.starts, train1_ends=s_train1.ends,
         dens=dens, weights=w, rag_deltas=rag.deltas, weights_rag=w_rag,
         pdf_eval_starts=p_eval.starts, pdf_eval_ends=p_eval.ends, pdf_eval_sp0=p_eval.spacing_starts,
         pdf_eval_sp1=p_eval.spacing_ends, w_zero=w_zero, pdf_zero_starts=p_zero.starts, pdf_zero_ends=p_zero.ends,
         pdf_train_starts=p_train.starts, pdf_train_ends=p_train.ends)

    # ---------------- a5: pixel-aligned features ------------------------------------------------
    feats = randn(14, B, 16, 6, 8)
    xyz = randn(15, B, 40, 3) * torch.tensor([1.0, 1.0, 1.5]) + torch.tensor([0.0, 0.0, 3.0])
    xyz[0, 0] = torch.tensor([50.0, -40.0, 2.0])  # far outside the image: exercises border padding
    xyz[1, 1] = torch.tensor([-30.0, 60.0, 1.0])
    pf, pc, puv = pixel_aligned_features.get_pixel_aligned_features(xyz, c2w, K, feats)
    save("pixel_aligned", feats=feats, xyz=xyz, c2w=c2w, k_norm=K, out_feats=pf, out_xyz_cam=pc, out_uv=puv)

    # ---------------- a7/a8: ResnetFC + activation ---------------------------------------------
    mlp_cfg = MlpCfg(n_blocks=5, d_hidden=128, combine_layer=3, combine_type="mean", beta=0.0)
    zin, xin = randn(16, 2, 10, 512), randn(17, 2, 10, 63)
    outs = {}
    for d_out in (1, 16, 24):
        net = ResnetFC(mlp_cfg, d_in=63, d_latent=512, d_out=d_out)
        load_seeded(net, f"fc{d_out}.", seed=3)
        outs[f"out{d_out}"] = net(zin, xin).output
    act = activations.init_density_activation("trunc_exp")
    pre = randn(18, 4, 9) * 3
    save("resnet_fc", z=zin, x=xin, pre=pre, dens=act(pre), **outs)

    # ---------------- models ---------------------------------------------------------------------
    enc_cfg = EncoderResnetCfg(name="resnet", upsample_interp="bilinear", num_layers=4, use_first_pool=True,
                               norm_type="batch")
    dens_cfg = DensityDecoderMlpCfg(name="density_mlp", mlp=mlp_cfg)
    mlp_dec = ActionDecoderJacobianMlpCfg(name="jacobian_mlp", mlp=mlp_cfg)
    tr_dec = ActionDecoderJacobianTransformerCfg(
        name="jacobian_transformer", mlp=mlp_cfg,
        transformer=TransformerCfg(attn_feat_dim=64, attn_head_dim=64, num_attn_heads=8, attn_depth=3, attn_mlp_dim=64))

    flow_dec = ActionDecoderFlowMlpCfg(name="flow_mlp", mlp=mlp_cfg)  # the reference's direct-flow ablation decoder

    def build(dec_cfg, action_dim, n_prop, n_nerf):
        rcfg = ref_model.RenderingCfg(num_proposal_samples=tuple(n_prop), num_nerf_samples=n_nerf, single_jitter=False,
                                      proposal_warmup=5000, proposal_update_every=5, use_proposal_weight_anneal=True,
                                      proposal_weights_anneal_max_num_iters=1000, proposal_weights_anneal_slope=10.0)
        cfg = ref_model.ModelCfg(action_dim=action_dim, rendering=rcfg, encoder=enc_cfg, density_decoder=dens_cfg,
                                 action_decoder=dec_cfg)
        m = ref_model.Model(cfg)
        shapes = load_seeded(m, "", seed=0)
        return m, shapes

    manifest = {}
    H = W = 16
    coords16, _ = geometry.get_pixel_coordinates(H, W)
    ctx_c2w = torch.eye(4)[None].repeat(B, 1, 1) + 0.0
    ctx_c2w[1] = rigid(21, 1)[0]
    trg_c2w = rigid(22, B)
    trg_c2w[:, :3, 3] *= 0.5
    Kn = k_norm(B)
    image = rand(23, B, 3, H, W)
    sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(24))[:20]
    xy16 = coords16.reshape(1, -1, 2)[:, sel].repeat(B, 1, 1)
    ro, rd, rz = geometry.get_world_rays_with_z(xy16, Kn, trg_c2w)
    kpix = convention.denormalize_intrinsics(Kn, width=W, height=H)
    z_near, z_far = torch.tensor([0.5, 0.4]), torch.tensor([10.0, 6.0])

    for tag, dec_cfg, A in (("mlp", mlp_dec, 8), ("transformer", tr_dec, 6), ("flow", flow_dec, 5)):
        model, shapes = build(dec_cfg, A, [16], 12)
        manifest[tag] = {k: list(v) for k, v in shapes.items()}
        action = 0.1 * randn(25, B, A)
        cam = ref_model.CameraInput(input_image=image, ctxt_extrinsics=ctx_c2w, ctxt_intrinsics=Kn,
                                    trgt_extrinsics=trg_c2w, trgt_intrinsics=kpix)
        rin = ref_model.RenderingInput(origins=ro, directions=rd, z_near=z_near, z_far=z_far)
        rob = ref_model.RobotInput(robot_action=action)
        model.eval()
        if tag == "flow":
            # flow_mlp: Model.forward + the decoder on the final samples.  No visualisation features (640 hidden
            # channels nothing reads), no encode_image / infer_optical_flow (the reference's flow_mlp.encode_image
            # returns a map object, action_decoder_flow.py:246-279).  The action is scaled up so that its path through
            # lin_z is well above the comparison tolerance.
            action = 5.0 * action
            rob = ref_model.RobotInput(robot_action=action)
            with torch.no_grad():
                feats_e = model.encoder.forward(image)
                out = model.forward(cam, rin, rob)
                out0 = model.forward(cam, rin, ref_model.RobotInput(robot_action=torch.zeros_like(action)))
                penc = PixelEncoding(features=feats_e, extrinsics=ctx_c2w, intrinsics=Kn, action=action)
                rb = model.compute_ray_bundle(rin)
                samples, pos, dirs, wl, sl = model.compute_proposal(rb, penc)
                dec = model.decoder.forward(world_space_xyz=pos, world_space_dir=dirs, pixel_encoding=penc)
            save("model_flow", image=image, features=feats_e, ctxt_c2w=ctx_c2w, ctxt_k_norm=Kn, trgt_c2w=trg_c2w,
                 trgt_k_pix=kpix, origins=ro, directions=rd, z_near=z_near, z_far=z_far, action=action,
                 rgb=out.standard_output.rgb, depth=out.standard_output.depth,
                 optical_flow=out.standard_output.optical_flow, optical_flow_zero_action=out0.standard_output.optical_flow,
                 final_starts=samples.starts, final_ends=samples.ends, final_positions=pos,
                 dec_density=dec.density, dec_color=dec.color, dec_flow=dec.flow)
            continue
        with torch.no_grad():
            feats_e = model.encoder.forward(image)
            out = model.forward(cam, rin, rob, compute_vis_features=True)
            penc = PixelEncoding(features=feats_e, extrinsics=ctx_c2w, intrinsics=Kn, action=action)
            # per-sample decoder outputs on the final samples (recomputed through the public pieces)
            rb = model.compute_ray_bundle(rin)
            samples, pos, dirs, wl, sl = model.compute_proposal(rb, penc)
            dec = model.decoder.forward(world_space_xyz=pos, world_space_dir=dirs, pixel_encoding=penc)
            prop_d = model.proposal_networks[0].get_density(sl[0].get_positions(), penc)
            enc_out = model.encode_image(cam, rin, rob)
            flow_inf = model.infer_optical_flow(enc_out, cam, ref_model.RobotInput(robot_action=action * 2 + 0.05))
            dho, extras = model.compute_density(pos, penc) if False else (None, None)
        arrays = dict(
            image=image, features=feats_e, ctxt_c2w=ctx_c2w, ctxt_k_norm=Kn, trgt_c2w=trg_c2w, trgt_k_pix=kpix,
            origins=ro, directions=rd, z_near=z_near, z_far=z_far, action=action,
            rgb=out.standard_output.rgb, depth=out.standard_output.depth, optical_flow=out.standard_output.optical_flow,
            vis_action_features=out.vis_output.action_features, vis_steps=out.vis_output.steps,
            vis_weights=out.vis_output.weights, vis_ray_positions=out.vis_output.ray_positions,
            vis_ray_positions_warped=out.vis_output.ray_positions_warped,
            prop_starts=sl[0].starts, prop_ends=sl[0].ends, prop_density=prop_d, prop_weights=wl[0],
            final_starts=samples.starts, final_ends=samples.ends, final_positions=pos,
            dec_density=dec.density, dec_color=dec.color, dec_flow=dec.flow, dec_action_features=dec.action_features,
            enc_density=enc_out.density, enc_action_features=enc_out.action_features, enc_weights=enc_out.weights,
            enc_positions=enc_out.ray_samples_positions, infer_flow=flow_inf,
        )
        # training mode: stratified jitter from the global RNG + annealed proposal weights
        model.train()
        model.encoder.eval()  # keep BatchNorm on running stats so the encoder is deterministic
        model.step_before_iter(300)
        torch.manual_seed(200)
        with torch.no_grad():
            out_t = model.forward(cam, rin, rob)
        arrays.update(train_anneal=np.float32(model.proposal_sampler._anneal), train_rgb=out_t.standard_output.rgb,
                      train_depth=out_t.standard_output.depth, train_flow=out_t.standard_output.optical_flow,
                      train_w0=out_t.training_output.weights_list[0], train_w1=out_t.training_output.weights_list[1],
                      train_starts0=out_t.training_output.ray_samples_list[0].starts,
                      train_starts1=out_t.training_output.ray_samples_list[1].starts)
        save(f"model_{tag}", **arrays)

    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("  state_dict_manifest.json")

    # ---------------- a18: compositing on free-standing tensors ---------------------------------
    S = 9
    wts = torch.softmax(randn(30, B, 5, S, 1), dim=-2) * 0.9
    rgbs, posn, sflow = rand(31, B, 5, S, 3), randn(32, B, 5, S, 3) + torch.tensor([0.0, 0.0, 3.0]), 0.05 * randn(33, B, 5, S, 3)
    st = torch.sort(rand(34, B, 5, S + 1, 1) * 9 + 0.5, dim=-2).values
    rs = ray_samplers.RaySamples(origins=None, directions=None, starts=st[..., :-1, :], ends=st[..., 1:, :])
    c_rgb = ref_model.Model.render_rgb(rgbs, wts, None)
    c_depth, c_steps = ref_model.Model.render_depth(wts, rs)
    c_flow, c_p, c_pw = ref_model.Model.render_optical_flow(wts, posn, sflow, trg_c2w, kpix)
    save("composite", weights=wts, rgb=rgbs, positions=posn, scene_flow=sflow, starts=rs.starts, ends=rs.ends,
         trgt_c2w=trg_c2w, trgt_k_pix=kpix, out_rgb=c_rgb, out_depth=c_depth, out_steps=c_steps, out_flow=c_flow,
         out_pos=c_p, out_pos_warped=c_pw)

    # ---------------- a20: training-step contract ------------------------------------------------
    depth_t = rand(35, B, 5, 1) * 4
    depth_t[0, 0] = 0.0
    dl = loss_utils.ds_nerf_depth_loss(wts, depth_t, c_steps, rs.ends - rs.starts, torch.tensor([0.001]))
    save("losses", weights=wts, depth_target=depth_t, steps=c_steps, lengths=rs.ends - rs.starts, depth_loss=dl)


def config1_plumbing():
    """BASELINE.json config 0/C1: the 2D tutorial model's CPU plumbing -- UnetJacobianField.forward =
    UNet -> rearrange -> einsum with the command (project/jacobian/models/jacobian_models/unet_jacobian.py:38-66)."""
    import importlib.util
    pkg = types.ModuleType("jacobian"); pkg.__path__ = [os.path.join(REF, "jacobian")]; sys.modules["jacobian"] = pkg
    mods = types.ModuleType("jacobian.models"); mods.__path__ = [os.path.join(REF, "jacobian", "models")]
    sys.modules["jacobian.models"] = mods  # bypass jacobian/models/__init__.py (imports wandb)
    from jacobian.models.jacobian_models.unet_jacobian import UnetJacobianField, UnetJacobianFieldCfg
    torch.manual_seed(7)
    net = UnetJacobianField(UnetJacobianFieldCfg(command_dim=2, spatial_dim=2)).eval()
    img, cmd = rand(40, 1, 3, 128, 128), randn(41, 1, 2)
    with torch.no_grad():
        out = net(img, cmd)
    save("config1_unet2d", jacobian=out.jacobian, cmd=cmd, flow=out.flow)


if __name__ == "__main__":
    main()
    config1_plumbing()
