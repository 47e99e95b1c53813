"""Every code path of the plain-fp16 mode (precision "f16") that the parity suite's jacobian_mlp frames do not reach: the
fp16 map producers (projection, pyramid), the point-query kernels, the transformer and flow_mlp decoders, the pyramid route
through the ResNet encoder, ragged shapes, bit-reproducibility and ray-shard exactness.  Run with -m gpu.

Tolerance of this file (the reduced mode's per-network figure, oracle/parity_harness.py::REDUCED_TOL):
quantities evaluated at GIVEN positions against the REFERENCE's fixture tensors -- 4e-3 norm-wise (2 x 2e-3: they pass through
two networks); end-to-end pixels are compared with the same model's exact-fp32 forward and only for sanity (placement noise):
rgb 3e-2, depth 5e-2."""
import pytest
import torch

from test_model_api_gpu import (_from_reference_features, _inputs, dev, flow_model_and_golden, model_and_golden,  # noqa: F401
                                transformer_model_and_golden)

pytestmark = pytest.mark.gpu
AT_POSITIONS = 4e-3


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("shape", [(1, 16, 16, 384), (2, 9, 13, 832), (1, 5, 7, 40)])
def test_fp16_map_projection_vs_float64(dev, shape):  # noqa: F811
    """njf_project_features with NJF_PRECISION_F16: the error-compensated projection rounded ONCE to fp16."""
    from neural_jacobian_field_amd import hip
    b, hf, wf, n = shape
    g = torch.Generator().manual_seed(hf * 100 + n)
    feats = (torch.randn(b, 512, hf, wf, generator=g) * 3).to(dev)
    wz = (torch.randn(512, n, generator=g) * 0.05).to(dev)
    bz = torch.randn(n, generator=g).to(dev)
    out = torch.empty(b, hf, wf, n, device=dev, dtype=torch.float16)
    hip.project_features(feats, wz, bz, out, precision="f16")
    ref = torch.einsum("bkhw,kn->bhwn", feats.double(), wz.double()) + bz.double()
    assert rel(out, ref) < 2.0 ** -11, rel(out, ref)
    # element-wise: within one fp16 rounding of the float64 value (+ the fp32 projection's own absolute rounding error)
    assert ((out.double().cpu() - ref.cpu()).abs() <= ref.cpu().abs() * 2.0 ** -11 + 4e-6 * ref.abs().max().item()).all()
    with pytest.raises(ValueError, match="float16"):
        hip.project_features(feats, wz, bz, torch.empty(b, hf, wf, n, device=dev), precision="f16")


@pytest.mark.parametrize("sizes", [((64, 24, 40), (64, 12, 20), (128, 6, 10), (256, 3, 5)),      # blocked up-sampled add
                                   ((64, 18, 30), (64, 9, 15), (128, 5, 8), (256, 3, 4)),         # per-texel form
                                   ((512, 6, 10),)])                                                # one level: straight into the fp16 map
def test_fp16_map_pyramid_vs_float64(dev, sizes):  # noqa: F811
    import torch.nn.functional as F
    from neural_jacobian_field_amd import hip
    g = torch.Generator().manual_seed(78)
    b, n = 2, 832
    levels = [torch.randn(b, c, h, w, generator=g).to(dev) for c, h, w in sizes]
    wz = (torch.randn(512, n, generator=g) * 0.05).to(dev)
    bz = torch.randn(n, generator=g).to(dev)
    feats = torch.cat([F.interpolate(lv, levels[0].shape[-2:], mode="bilinear", align_corners=False) for lv in levels], dim=1)
    ref = torch.einsum("bkhw,kn->bhwn", feats.double(), wz.double()) + bz.double()
    out = torch.empty(b, sizes[0][1], sizes[0][2], n, device=dev, dtype=torch.float16)
    hip.project_pyramid(levels, wz, bz, out, precision="f16")
    assert ((out.double().cpu() - ref.cpu()).abs() <= ref.cpu().abs() * 2.0 ** -11 + 4e-6 * ref.abs().max().item()).all()


def _decoder_rows(model, g, keys=("density", "color", "flow", "action_features")):
    from neural_jacobian_field_amd.decoder import PixelEncoding
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    pos = g["final_positions"]
    dirs = g["directions"][..., None, :].expand(pos.shape).contiguous()
    dec = model.decoder.forward(pos, dirs, enc)
    return {k: rel(getattr(dec, k), g["dec_" + k]) for k in keys if getattr(dec, k) is not None}, enc, pos


def test_f16_point_queries_at_reference_sample_locations(model_and_golden, margins):  # noqa: F811
    """decoder.forward / get_density / encode_image / compute_density (njf_points_forward, every MODE) in the plain-fp16 mode,
    on the reference's own sample positions, against the reference's fixture tensors."""
    model, g = model_and_golden
    model.set_precision("f16")
    try:
        rows, enc, pos = _decoder_rows(model, g)
        prop_pos = g["origins"][..., None, :] + g["directions"][..., None, :] * (g["prop_starts"] + g["prop_ends"]) / 2
        rows["proposal.get_density"] = rel(model.proposal_networks[0].get_density(prop_pos, enc), g["prop_density"])
        fo = model.decoder.encode_image(pos, enc)
        rows["encode_image.density"] = rel(fo.density, g["enc_density"])
        rows["encode_image.action_features"] = rel(fo.action_features, g["enc_action_features"])
        head, extras = model.compute_density(pos.reshape(pos.shape[0], -1, 3), enc)
        rows["compute_density.density"] = rel(head.density.reshape(g["dec_density"].shape), g["dec_density"])
        # end to end (jacobian_mlp render kernel + proposal kernel) against the same model in exact fp32: sanity only
        with _from_reference_features(model, g):
            out16 = model.forward(*_inputs(g)).standard_output
            model.set_precision("f32")
            out32 = model.forward(*_inputs(g)).standard_output
        e2e = {"rgb": rel(out16.rgb, out32.rgb), "depth": rel(out16.depth, out32.depth)}
    finally:
        model.set_precision("f16x2")
    margins.record("f16.paths[jacobian_mlp @ reference positions]",
                   [{"key": k, "err": float(f"{v:.3e}"), "limit": AT_POSITIONS, "ok": v <= AT_POSITIONS, "needs_floor": False,
                     "floor": 0.0, "floor_fp64": 0.0, "self_noise_floor_used": False} for k, v in rows.items()])
    assert all(v <= AT_POSITIONS for v in rows.values()), rows
    assert e2e["rgb"] < 3e-2 and e2e["depth"] < 5e-2, e2e


def test_f16_transformer_head(transformer_model_and_golden, margins):  # noqa: F811
    """The folded transformer head (64-wide layers: the generic plain-fp16 chunk form, 64-channel fp16 query block)."""
    model, g = transformer_model_and_golden
    model.set_precision("f16")
    try:
        rows, enc, pos = _decoder_rows(model, g)
        fo = model.decoder.encode_image(pos, enc)
        rows["encode_image.action_features"] = rel(fo.action_features, g["enc_action_features"])
        with _from_reference_features(model, g):
            out16 = model.forward(*_inputs(g), compute_vis_features=True)
            model.set_precision("f32")
            out32 = model.forward(*_inputs(g), compute_vis_features=True)
    finally:
        model.set_precision("f16x2")
    margins.record("f16.paths[jacobian_transformer @ reference positions]",
                   [{"key": k, "err": float(f"{v:.3e}"), "limit": AT_POSITIONS, "ok": v <= AT_POSITIONS, "needs_floor": False,
                     "floor": 0.0, "floor_fp64": 0.0, "self_noise_floor_used": False} for k, v in rows.items()])
    assert all(v <= AT_POSITIONS for v in rows.values()), rows
    assert rel(out16.standard_output.rgb, out32.standard_output.rgb) < 3e-2
    assert torch.isfinite(out16.vis_output.action_features).all() and torch.isfinite(out16.standard_output.optical_flow).all()


def test_f16_flow_mlp_decoder(flow_model_and_golden):  # noqa: F811
    """flow_mlp adds its per-image action bias to an fp16 map (decoder.ActionDecoderFlowMlp.hoisted_map)."""
    model, g = flow_model_and_golden
    model.set_precision("f16")
    try:
        rows, _, _ = _decoder_rows(model, g, keys=("density", "color", "flow"))
    finally:
        model.set_precision("f16x2")
    assert all(v <= AT_POSITIONS for v in rows.values()), rows


def test_f16_through_the_resnet_encoder_equals_the_concatenated_route(model_and_golden):  # noqa: F811
    """Model.forward on an image: the fp16 hoisted map comes from the encoder's latents (njf_project_pyramid + one rounding) --
    against the same forward on the concatenated 512-channel encoder output (njf_project_features): the two fp16 maps differ by
    single fp16 roundings, the frames by what that does to sample placement."""
    model, g = model_and_golden
    model.set_precision("f16")
    try:
        out_pyr = model.forward(*_inputs(g)).standard_output
        feats = model.encoder.forward(g["image"])
        original = model._encode_for_render
        model._encode_for_render = lambda image: feats
        try:
            out_cat = model.forward(*_inputs(g)).standard_output
        finally:
            model._encode_for_render = original
    finally:
        model.set_precision("f16x2")
    assert torch.isfinite(out_pyr.rgb).all() and torch.isfinite(out_pyr.optical_flow).all()
    assert rel(out_pyr.rgb, out_cat.rgb) < 3e-2 and rel(out_pyr.depth, out_cat.depth) < 5e-2


@pytest.mark.parametrize("shape", [dict(batch=3, height=16, width=20, rays=37, s_prop=40, s_final=48),
                                   dict(batch=1, height=16, width=16, rays=5, s_prop=33, s_final=31),
                                   dict(batch=5, height=16, width=16, rays=1, s_prop=64, s_final=65, action_dim=3)])
def test_f16_ragged_shapes(dev, shape, margins):  # noqa: F811
    import parity_harness as ph
    rep = ph.run_parity_case(device=dev, precision="f16", **shape)
    margins.record(f"parity[ragged:f16:{shape['rays']}r]", rep["rows"])
    assert rep["ok"], {k: v for k, v in rep.items() if k not in ("rows", "truth_rows")}


def test_f16_is_bit_reproducible_and_shard_exact(dev):  # noqa: F811
    """Same inputs twice -> identical bits; a ray shard renders the bits of the full batch (rays are independent units in this
    mode too: the one-rank-per-GPU split of section 6 applies unchanged)."""
    import parity_harness as ph
    from neural_jacobian_field_amd.renderer import RenderRequest
    case = ph.make_case(2, 32, 32, 300, 8, seed=3, identity_context=False)
    req = RenderRequest(vis=True, sample_weights=True)
    a, _, _ = ph.hip_forward(case, 64, 64, dev, request=req, precision="f16")
    b, _, _ = ph.hip_forward(case, 64, 64, dev, request=req, precision="f16")
    assert torch.equal(a.rgb, b.rgb) and torch.equal(a.depth, b.depth) and torch.equal(a.optical_flow, b.optical_flow)
    assert torch.equal(a.bins_list[1], b.bins_list[1])
    sub = dict(case, origins=case["origins"][:, 100:231].contiguous(), directions=case["directions"][:, 100:231].contiguous())
    c, _, _ = ph.hip_forward(sub, 64, 64, dev, request=RenderRequest(), precision="f16")
    assert torch.equal(c.rgb, a.rgb[:, 100:231]) and torch.equal(c.optical_flow, a.optical_flow[:, 100:231])


def test_f16_shading_with_compensated_placement(model_and_golden):  # noqa: F811
    """set_precision("f16", proposal_precision="f16x2"): the proposal pass keeps fp32-class sample PLACEMENT (its own fp32 hoisted
    map; the networks no longer share one projection), only the final pass runs in plain fp16 -- the end-to-end pixels are then
    within the per-network figure of the exact-fp32 frame instead of carrying placement noise."""
    model, g = model_and_golden
    try:
        with _from_reference_features(model, g):
            model.set_precision("f16", proposal_precision="f16x2")
            assert model.decoder.precision == "f16" and all(p.precision == "f16x2" for p in model.proposal_networks)
            mixed = model.forward(*_inputs(g)).standard_output
            model.set_precision("f32")
            exact = model.forward(*_inputs(g)).standard_output
    finally:
        model.set_precision("f16x2")
    errs = {"rgb": rel(mixed.rgb, exact.rgb), "depth": rel(mixed.depth, exact.depth), "flow": rel(mixed.optical_flow, exact.optical_flow)}
    assert errs["rgb"] < AT_POSITIONS and errs["depth"] < AT_POSITIONS and errs["flow"] < 2e-2, errs
