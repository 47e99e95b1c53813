"""GPU tests of the drop-in API (Model / decoders / samplers / geometry) against the REFERENCE's own
golden vectors (tests/golden/*.npz, produced by importing the reference).  Run with -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model_and_golden(dev, golden):
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import Model
    g = golden("model_mlp")
    cfg = model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                               "action_decoder": {"name": "jacobian_mlp"}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", 8), seed=0), strict=True)
    model.to(dev).eval().requires_grad_(False)  # inference: the in-kernel compositing path
    g64 = {k + "_f64": v for k, v in golden("model_mlp_f64").items()}  # the reference evaluated in float64 (floors)
    return model, {k: v.to(dev) for k, v in {**g, **g64}.items()}


def _inputs(g):
    from neural_jacobian_field_amd.model import CameraInput, RenderingInput, RobotInput
    cam = CameraInput(input_image=g["image"], ctxt_extrinsics=g["ctxt_c2w"], ctxt_intrinsics=g["ctxt_k_norm"],
                      trgt_extrinsics=g["trgt_c2w"], trgt_intrinsics=g["trgt_k_pix"])
    return cam, RenderingInput(g["origins"], g["directions"], g["z_near"], g["z_far"]), RobotInput(g["action"])


def test_encoder_matches_reference(model_and_golden, margins):
    model, g = model_and_golden
    margins("model_mlp.encoder", "features", model.encoder(g["image"]), g["features"], g["features_f64"], tol=1e-5)  # MIOpen vs CPU


def test_fused_trunk_epilogues_equal_the_library_ops(model_and_golden, margins):
    """The frozen trunk's convolution epilogues as ONE launch each (njf_bn_act: eval-mode batch norm [+ skip] + ReLU, in place on the
    convolution's output; encoder._Block.forward under no_grad) against the library's batch-norm / add / ReLU kernels on the same
    convolutions: every latent within 2e-6 (the two evaluate the same expression; the convolutions themselves move by ~5e-7 run to
    run), for the golden image, a batch of three, and a size whose planes are not a multiple of four floats (scalar path); the
    concatenated feature map through the fused trunk against the reference's golden features like test_encoder_matches_reference;
    grad-enabled and train-mode calls never take the fused path."""
    from neural_jacobian_field_amd import encoder as enc_mod
    model, g = model_and_golden
    enc = model.encoder
    assert enc_mod._FUSED_EPILOGUES, "NJF_ENCODER_FUSED=0 in the test environment"
    gen = torch.Generator(device="cpu").manual_seed(4)
    dev = g["image"].device
    images = [g["image"], torch.rand(3, 3, 64, 96, generator=gen).to(dev), torch.rand(1, 3, 60, 60, generator=gen).to(dev)]
    calls = []
    real = enc_mod._fused_epilogue
    try:
        enc_mod._fused_epilogue = lambda x, bn: calls.append(real(x, bn)) or calls[-1]
        for im in images:
            with torch.no_grad():
                calls.clear()
                fused = enc._latents(im)
                assert calls and all(calls), "the fused path did not run"
                enc_mod._FUSED_EPILOGUES = False
                try:
                    plain = enc._latents(im)
                finally:
                    enc_mod._FUSED_EPILOGUES = True
            for a, b in zip(fused, plain):
                assert a.shape == b.shape and rel(a, b) <= 2e-6, (tuple(im.shape), rel(a, b))
        calls.clear()
        enc._latents(g["image"])                                     # grad mode on: library ops
        assert calls and not any(calls)
        with torch.no_grad():
            margins("model_mlp.encoder[fused epilogues]", "features", enc(g["image"]), g["features"], g["features_f64"], tol=1e-5)
    finally:
        enc_mod._fused_epilogue = real


def test_frozen_encoder_trunk_as_hip_graph_equals_the_eager_trunk(model_and_golden, margins):
    """EncoderResnet.forward_pyramid of a frozen encoder in eval mode: the first call with a shape runs eagerly, the second captures
    the trunk as one HIP graph, later calls replay it on a copy of the image.  The latents must be what the eager trunk gives for
    EVERY image (same kernels in the same order: held to 1e-6, MIOpen's own run-to-run spread), the caller owns them (a later call
    does not change an earlier result), a weight change re-captures, and grad-enabled / train-mode calls stay eager."""
    from neural_jacobian_field_amd.encoder import EncoderResnet
    model, g = model_and_golden
    enc = model.encoder
    if EncoderResnet._graph_disabled:
        pytest.skip("NJF_ENCODER_GRAPH=0")
    EncoderResnet._graph_states.pop(enc, None)
    gen = torch.Generator(device="cpu").manual_seed(0)
    images = [g["image"]] + [torch.rand(g["image"].shape, generator=gen).to(g["image"].device) for _ in range(3)]
    with torch.no_grad():
        eager = [[lv.clone() for lv in enc._latents(im)] for im in images]
        got, kept = [], None
        for i, im in enumerate(images + images):
            out = enc.forward_pyramid(im).levels
            state = EncoderResnet._graph_states.get(enc)
            assert state is not None and (state[0] == "seen") == (i == 0), (i, state[0])     # captured on the second call
            if i == 1:
                kept = (out, [lv.clone() for lv in out])
            got.append(out)
        for i, out in enumerate(got):
            for lv, ref in zip(out, eager[i % len(images)]):
                assert lv.shape == ref.shape and rel(lv, ref) <= 1e-6, (i, rel(lv, ref))
        assert all(torch.equal(a, b) for a, b in zip(*kept))                                  # replays did not touch a returned result
        graph = EncoderResnet._graph_states[enc][0]
        enc.model.conv1.weight.mul_(1.0)                                                      # version bump: the capture is stale
        enc.forward_pyramid(images[0])
        assert EncoderResnet._graph_states[enc][0] == "seen"
        out = enc.forward_pyramid(images[1]).levels
        assert EncoderResnet._graph_states[enc][0] is not graph and rel(out[-1], eager[1][-1]) <= 1e-6
    before = EncoderResnet._graph_states[enc]
    enc.forward_pyramid(images[0])                       # grad mode: eager, state untouched
    assert EncoderResnet._graph_states[enc] is before
    # and the whole forward pass through the captured trunk is the golden one
    # (two runs of the MIOpen trunk differ by ~5e-7, which the renderer's flow amplifies to ~1e-4: held to the golden's own floors)
    cam, rin, rob = _inputs(g)
    with torch.no_grad():
        out = model.forward(cam, rin, rob, compute_vis_features=True)
    assert EncoderResnet._graph_states[enc] is before and before[0] != "seen"               # (it was a replay)
    _check_forward(margins, "model_mlp.forward[through the graphed encoder]", out, g)


def _noise(g, key, encoder):
    """The reference's own fp32 movement of output `key` under a one-ulp ray perturbation (+ a 1e-5 feature perturbation when
    the comparison runs through the MIOpen encoder); scalars stored next to the float64 golden."""
    kinds = ("floor_ulp", "floor_enc") if encoder else ("floor_ulp",)
    return [g[f"{k}.{key}_f64"].item() if f"{k}.{key}_f64" in g else g[f"{k}.{key}"].item() for k in kinds
            if f"{k}.{key}_f64" in g or f"{k}.{key}" in g]


def _check_forward(margins, case, out, g, vis=True, encoder=True, truth_assert=False):
    """Model.forward's outputs against the reference's fp32 golden.  Bound per output: max(1e-4, 2 x floor), floor = the
    largest of |ref32 - ref64| and the reference's own movement under unavoidable input perturbations (`_noise`).
    ``truth_assert``: additionally hold every output to the truth-referenced element-wise criterion against the reference's
    float64 run (oracle/parity_harness.py::truth_columns)."""
    so = out.standard_output
    m = lambda key, name, got: margins(case, key, got, g[name], g[name + "_f64"], self_noise=_noise(g, name, encoder),
                                       truth_assert=truth_assert)
    m("rgb", "rgb", so.rgb)
    m("depth", "depth", so.depth)
    m("optical_flow", "optical_flow", so.optical_flow)
    if vis:
        vo = out.vis_output
        m("vis.steps", "vis_steps", vo.steps)
        m("vis.ray_positions", "vis_ray_positions", vo.ray_positions)
        m("vis.ray_positions_warped", "vis_ray_positions_warped", vo.ray_positions_warped)
        m("vis.action_features", "vis_action_features", vo.action_features)
        m("vis.weights", "vis_weights", vo.weights)


def _from_reference_features(model, g):
    """Context manager: Model.forward on the REFERENCE's encoder output (isolates the rendering path from MIOpen)."""
    import contextlib

    @contextlib.contextmanager
    def scope():
        original = model._encode_for_render
        model._encode_for_render = lambda image: g["features"]
        try:
            yield
        finally:
            model._encode_for_render = original
    return scope()


def test_model_forward_vs_reference_golden(model_and_golden, margins):
    """End to end through the MIOpen encoder; batch element 1 has a general context pose."""
    model, g = model_and_golden
    out = model.forward(*_inputs(g), compute_vis_features=True)
    _check_forward(margins, "model_mlp.forward[through encoder]", out, g)
    assert out.training_output is None


@pytest.mark.parametrize("precision", ["f32", "f16x2", "f16f6"])
def test_model_forward_from_reference_features(model_and_golden, margins, precision):
    """The rendering path alone (the reference's encoder output is fed in), in every MFMA precision, same bounds."""
    model, g = model_and_golden
    model.set_precision(precision)
    try:
        with _from_reference_features(model, g):
            out = model.forward(*_inputs(g), compute_vis_features=True)
    finally:
        model.set_precision("f16x2")
    _check_forward(margins, f"model_mlp.forward[{precision}]", out, g, encoder=False, truth_assert=precision == "f32")


def test_decoder_forward_at_reference_sample_locations(model_and_golden, margins):
    """ActionDecoder.forward / DensityDecoderMlp.get_density / encode_image on the reference's own sample positions.
    Floors: the reference's float64 decoder evaluated at the SAME (fp32) positions -- what is left is one-ulp
    camera-space differences times the positional encoding's gain (general context pose on batch element 1)."""
    from neural_jacobian_field_amd.decoder import PixelEncoding
    model, g = model_and_golden
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    pos = g["final_positions"]
    dirs = g["directions"][..., None, :].expand(pos.shape).contiguous()
    dec = model.decoder.forward(pos, dirs, enc)
    c = "model_mlp.decoder@ref-positions"
    margins(c, "density", dec.density, g["dec_density"], g["dec_density_f64"])
    margins(c, "color", dec.color, g["dec_color"], g["dec_color_f64"])
    margins(c, "flow", dec.flow, g["dec_flow"], g["dec_flow_f64"])
    margins(c, "action_features", dec.action_features, g["dec_action_features"], g["dec_action_features_f64"])
    prop_pos = g["origins"][..., None, :] + g["directions"][..., None, :] * (g["prop_starts"] + g["prop_ends"]) / 2
    margins(c, "proposal.get_density", model.proposal_networks[0].get_density(prop_pos, enc), g["prop_density"], g["prop_density_f64"])
    fo = model.decoder.encode_image(pos, enc)
    margins(c, "encode_image.density", fo.density, g["enc_density"], g["encpos_density_f64"])
    margins(c, "encode_image.action_features", fo.action_features, g["enc_action_features"], g["encpos_action_features_f64"])
    head, extras = model.compute_density(pos.reshape(pos.shape[0], -1, 3), enc)
    margins(c, "compute_density.density", head.density.reshape(g["dec_density"].shape), g["dec_density"], g["dec_density_f64"])
    margins(c, "compute_density.jacobian", extras["jacobian_head_output"].reshape(g["dec_action_features"].shape),
            g["dec_action_features"], g["dec_action_features_f64"])


def test_identity_context_element_is_tight(model_and_golden):
    """Batch element 0 of the fixture has the identity context pose the reference dataset guarantees
    (data/dataset/dataset.py:363-365): there the per-sample outputs must agree to 1e-4 or better."""
    from neural_jacobian_field_amd.decoder import PixelEncoding
    model, g = model_and_golden
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    pos = g["final_positions"]
    dirs = g["directions"][..., None, :].expand(pos.shape).contiguous()
    dec = model.decoder.forward(pos, dirs, enc)
    assert rel(dec.density[0], g["dec_density"][0]) < 1e-4
    assert rel(dec.color[0], g["dec_color"][0]) < 1e-4
    assert rel(dec.action_features[0], g["dec_action_features"][0]) < 1e-4


def test_encode_image_and_infer_optical_flow(model_and_golden, margins):
    from neural_jacobian_field_amd.model import ModelInferenceEncoding, RobotInput
    model, g = model_and_golden
    cam, rin, rob = _inputs(g)
    enc = model.encode_image(cam, rin, rob)
    assert enc.density.shape == g["enc_density"].shape and enc.action_features.shape == g["enc_action_features"].shape
    # per-sample weights at INDEPENDENTLY placed samples: the floor is the reference's own fp32-vs-fp64 run
    margins("model_mlp.encode_image", "weights", enc.weights, g["enc_weights"], g["enc_weights_f64"],
            self_noise=[g["floor_ulp.enc_weights_f64"].item(), g["floor_enc.vis_weights_f64"].item()])
    # infer_optical_flow on the REFERENCE's cached encoding: pure compositing + projection
    ref_enc = ModelInferenceEncoding(g["enc_density"], g["enc_action_features"], g["enc_weights"], g["enc_positions"])
    action = (g["action"] * 2 + 0.05).requires_grad_(True)
    flow = model.infer_optical_flow(ref_enc, cam, RobotInput(action))
    margins("model_mlp.infer_optical_flow", "flow", flow, g["infer_flow"])
    flow.square().sum().backward()  # the inverse-dynamics loop differentiates w.r.t. the action
    assert action.grad is not None and torch.isfinite(action.grad).all()


def test_training_mode_outputs(model_and_golden):
    model, g = model_and_golden
    model.train()
    model.encoder.eval()
    model.step_before_iter(300)
    try:
        torch.manual_seed(0)
        out = model.forward(*_inputs(g))
    finally:
        model.eval()
        model.proposal_sampler.set_anneal(1.0)
    to = out.training_output
    assert len(to.weights_list) == 2 and len(to.ray_samples_list) == 2
    assert to.weights_list[0].shape == g["train_w0"].shape and to.weights_list[1].shape == g["train_w1"].shape
    s0 = to.ray_samples_list[0]
    assert (s0.deltas > 0).all() and (s0.spacing_starts >= 0).all() and (s0.spacing_ends <= 1).all()
    assert ((to.weights_list[1].sum(-2) <= 1 + 1e-5).all())
    # stratified bins stay inside their strata (ray_samplers.py:226-233)
    edges = torch.linspace(0, 1, 17, device=s0.starts.device)
    centers = (edges[1:] + edges[:-1]) / 2
    lower = torch.cat([edges[:1], centers])[:-1]
    upper = torch.cat([centers, edges[-1:]])[:-1]
    b = s0.spacing_starts[..., 0]
    assert (b >= lower - 1e-6).all() and (b <= upper + 1e-6).all()


def test_patch_render_equals_forward(model_and_golden):
    from neural_jacobian_field_amd import geometry
    from neural_jacobian_field_amd.model import RenderingInput
    model, g = model_and_golden
    cam, _, rob = _inputs(g)
    h = w = 8
    k_norm = g["ctxt_k_norm"]
    o, d, _ = geometry.full_frame_rays(h, w, k_norm, g["trgt_c2w"])
    rin = RenderingInput(o, d, g["z_near"], g["z_far"])
    # (both calls render from the reference's encoder output: MIOpen's convolutions are not bit-reproducible call to call --
    # measured 1.04e-5 on rgb between two encoder runs -- and with identical features the two calls ARE the same launches)
    with _from_reference_features(model, g):
        ro = model.patch_render(cam, rin, rob, render_height=h, render_width=w)
        out = model.forward(cam, rin, rob, compute_vis_features=True)
    assert ro.rgb.shape == (2, h, w, 3) and ro.weights.shape == (2, h, w, 12)
    assert torch.equal(ro.rgb.reshape(2, -1, 3), out.standard_output.rgb)
    assert torch.equal(ro.action_features.reshape(2, h * w, -1), out.vis_output.action_features)
    through_encoder = model.patch_render(cam, rin, rob, render_height=h, render_width=w)   # ... and through MIOpen: close
    assert rel(through_encoder.rgb, ro.rgb) < 1e-3
    # colour-mapped outputs (model.py:598-626) are produced on the device
    assert ro.depth_rgb.shape == (2, h, w, 3) and ro.depth_rgb.is_cuda and 0 <= ro.depth_rgb.min() and ro.depth_rgb.max() <= 1
    assert ro.flow_rgb.shape == (2, h, w, 3) and ro.flow_rgb.dtype == torch.uint8 and ro.flow_rgb.is_cuda
    # and the rendered Jacobian field feeds the reference's sensitivity colouring without leaving the device
    from neural_jacobian_field_amd.inference import jacobian_color_map as cm
    sens = cm.compute_joint_sensitivity(ro.action_features, cam.trgt_extrinsics[:, None, None, None])
    assert sens.shape == (2, 8, h, w) and sens.is_cuda and sens.min() >= 0 and sens.max() <= 1
    image = cm.visualize_joint_sensitivity(sens, torch.tensor(cm.JACOBIAN_COLORMAP["model_allegro"]).t())
    assert image.shape == (2, h, w, 3) and image.dtype.name == "uint8"


def test_geometry_and_samplers_vs_reference(dev, golden):
    from neural_jacobian_field_amd import geometry
    from neural_jacobian_field_amd.ray_samplers import PDFSampler, RayBundle, RaySamples, UniformSampler
    g = {k: v.to(dev) for k, v in golden("geometry").items()}
    coords, sel = geometry.get_pixel_coordinates(5, 7, dev)
    assert rel(coords, g["coords"]) < 1e-6 and torch.equal(sel, g["selector"])
    o, d, z = geometry.get_world_rays_with_z(g["xy"], g["k_norm"], g["c2w"])
    assert rel(o, g["origins"]) < 1e-6 and rel(d, g["directions"]) < 1e-6 and rel(z, g["z"]) < 1e-6
    o2, d2, _ = geometry.full_frame_rays(5, 7, g["k_norm"], g["c2w"])
    assert rel(d2, g["directions"]) < 1e-6
    assert torch.equal(geometry.denormalize_intrinsics(g["k_norm"], 7, 5), g["k_pix"])

    s = {k: v.to(dev) for k, v in golden("samplers").items()}
    rb = RayBundle(s["origins"], s["directions"], s["near"], s["far"])
    uni = UniformSampler().eval()
    smp = uni(rb, num_samples=12)
    assert rel(smp.starts, s["eval_starts"]) < 1e-6 and rel(smp.ends, s["eval_ends"]) < 1e-6
    assert rel(smp.get_positions(), s["eval_pos"]) < 1e-6
    assert rel(smp.get_weights(s["dens"]), s["weights"]) < 1e-5
    rag = RaySamples(smp.origins, smp.directions, smp.starts, smp.ends, deltas=s["rag_deltas"])
    assert rel(rag.get_weights(s["dens"]), s["weights_rag"]) < 1e-5           # zero / negative widths
    pdf = PDFSampler(include_original=False).eval()
    p = pdf(rb, smp, s["weights"], num_samples=10)
    assert rel(p.starts, s["pdf_eval_starts"]) < 1e-5 and rel(p.ends, s["pdf_eval_ends"]) < 1e-5
    pz = pdf(rb, smp, s["w_zero"], num_samples=10)                               # all-zero rays + delta pdf
    assert rel(pz.starts, s["pdf_zero_starts"]) < 1e-5 and rel(pz.ends, s["pdf_zero_ends"]) < 1e-5


def test_generic_sampler_route_equals_fused_route(model_and_golden, margins):
    """ProposalNetworkSampler with arbitrary density callbacks (reference API) vs the fused kernel."""
    from neural_jacobian_field_amd.decoder import PixelEncoding
    model, g = model_and_golden
    cam, rin, rob = _inputs(g)
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    rb = model.compute_ray_bundle(rin)
    smp, pos, dirs, wl, sl = model.compute_proposal(rb, enc)
    bins, wl2, _ = model.proposal_sampler.generate_ray_samples_fused(rb, list(model.proposal_networks), enc, g["z_near"],
                                                                    g["z_far"], True)
    assert rel(wl[0], wl2[0]) < 1e-5
    assert rel(smp.spacing_bins(), bins) < 1e-5
    margins("model_mlp.compute_proposal", "prop_weights", wl[0], g["prop_weights"], g["prop_weights_f64"],
            self_noise=_noise(g, "prop_weights", False))


# ---- jacobian_transformer decoder (default Allegro head; fixture uses A=6 -> exercises key masking) ----
@pytest.fixture(scope="module")
def transformer_model_and_golden(dev, golden):
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import Model
    g = golden("model_transformer")
    cfg = model_cfg_from_dict({"action_dim": 6, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                               "action_decoder": {"name": "jacobian_transformer"}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_transformer", 6), seed=0), strict=True)
    model.to(dev).eval().requires_grad_(False)  # inference: the in-kernel compositing path
    g64 = {k + "_f64": v for k, v in golden("model_transformer_f64").items()}
    return model, {k: v.to(dev) for k, v in {**g, **g64}.items()}


def test_transformer_decoder_at_reference_sample_locations(transformer_model_and_golden, margins):
    from neural_jacobian_field_amd.decoder import PixelEncoding
    model, g = transformer_model_and_golden
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    pos = g["final_positions"]
    dirs = g["directions"][..., None, :].expand(pos.shape).contiguous()
    dec = model.decoder.forward(pos, dirs, enc)
    # identity-context batch element: tight, no floor needed
    assert rel(dec.action_features[0], g["dec_action_features"][0]) < 1e-4
    assert rel(dec.flow[0], g["dec_flow"][0]) < 1e-4
    assert rel(dec.density[0], g["dec_density"][0]) < 1e-4
    # general pose element: the reference's own float64 evaluation at the same positions is the yardstick
    c = "model_transformer.decoder@ref-positions"
    margins(c, "action_features", dec.action_features, g["dec_action_features"], g["dec_action_features_f64"])
    margins(c, "flow", dec.flow, g["dec_flow"], g["dec_flow_f64"])
    margins(c, "density", dec.density, g["dec_density"], g["dec_density_f64"])
    margins(c, "color", dec.color, g["dec_color"], g["dec_color_f64"])
    fo = model.decoder.encode_image(pos, enc)
    margins(c, "encode_image.action_features", fo.action_features, g["enc_action_features"], g["encpos_action_features_f64"])


def test_transformer_model_forward_vs_reference_golden(transformer_model_and_golden, margins):
    model, g = transformer_model_and_golden
    out = model.forward(*_inputs(g), compute_vis_features=True)
    _check_forward(margins, "model_transformer.forward[through encoder]", out, g)
    with _from_reference_features(model, g):
        out = model.forward(*_inputs(g), compute_vis_features=True)
    _check_forward(margins, "model_transformer.forward", out, g, encoder=False)


@pytest.mark.parametrize("kind,tag,A", [("jacobian_mlp", "mlp", 8), ("jacobian_transformer", "transformer", 6)])
def test_arm_head_vs_reference_golden(dev, golden, margins, kind, tag, A):
    """use_arm_model + switch_mode("arm") (action_decoder_jacobian.py:89-90, 306-313, 330-331, 400-407, 438-446) for both
    decoders against the reference run in arm mode (tests/golden/model_arm.npz, arm_action_dim == action_dim): the decoder at
    the reference's sample positions and Model.forward end to end; regular mode on the same weights reproduces the
    reference's regular-mode flow, so the switch -- and the re-pack of the Jacobian block it triggers -- really selects
    another head; and the arm head trains in action mode (gradient lands on jacobian_head_arm.*, none on jacobian_head.*)."""
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.decoder import PixelEncoding
    from neural_jacobian_field_amd.model import Model
    raw = golden("model_arm")
    g = {k[len(tag) + 1:]: v.to(dev) for k, v in raw.items() if k.startswith(tag + ".") and v.dtype.is_floating_point}
    g.update({k: v.to(dev) for k, v in raw.items() if "." not in k})
    cfg = model_cfg_from_dict({"action_dim": A, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                               "action_decoder": {"name": kind, "use_arm_model": True, "arm_action_dim": A}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes(kind, A, arm_action_dim=A), seed=0), strict=True)
    model.to(dev).eval().requires_grad_(False)
    with _from_reference_features(model, g):
        regular = model.forward(*_inputs(g)).standard_output.optical_flow
    margins(f"model_arm[{tag}].forward[regular mode]", "optical_flow", regular, g["regular_optical_flow"], tol=5e-3)
    model.decoder.switch_mode("arm")
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    pos = g["final_positions"]
    dirs = g["directions"][..., None, :].expand(pos.shape).contiguous()
    dec = model.decoder.forward(pos, dirs, enc)
    c = f"model_arm[{tag}].decoder@ref-positions"
    margins(c, "action_features", dec.action_features, g["dec_action_features"], g["dec_action_features_f64"])
    margins(c, "flow", dec.flow, g["dec_flow"], g["dec_flow_f64"])
    margins(c, "density", dec.density, g["dec_density"], g["dec_density_f64"])
    with _from_reference_features(model, g):
        out = model.forward(*_inputs(g), compute_vis_features=True)
    c = f"model_arm[{tag}].forward"
    for key, got in (("rgb", out.standard_output.rgb), ("depth", out.standard_output.depth),
                     ("optical_flow", out.standard_output.optical_flow), ("vis_action_features", out.vis_output.action_features)):
        margins(c, key, got, g[key], g[key + "_f64"], self_noise=_noise(g, key, False))
    assert rel(out.standard_output.optical_flow, regular) > 1e-2          # the two heads are different functions
    # action-mode training of the arm head (the ResnetFC backward chain serves whichever ResnetFC head is active)
    model.decoder.freeze_non_action_parameters()
    for n, p in model.named_parameters():
        p.requires_grad = n.startswith("decoder.jacobian")
    try:
        with _from_reference_features(model, g):
            flow = model.forward(*_inputs(g)).standard_output.optical_flow
        (flow ** 2).mean().backward()
        named = dict(model.decoder.named_parameters())
        arm = [n for n in named if n.startswith("jacobian_head_arm.")]
        assert arm and all(named[n].grad is not None and torch.isfinite(named[n].grad).all() for n in arm)
        assert sum(float(named[n].grad.abs().sum()) for n in arm) > 0
        assert all(p.grad is None for n, p in named.items() if n.startswith("jacobian") and not n.startswith("jacobian_head_arm."))
    finally:
        model.requires_grad_(False)
        model.zero_grad(set_to_none=True)


def test_transformer_head_with_eight_keys(dev, golden, margins):
    """A = 8: every key slot of the folded attention is live (the A = 6 fixture above exercises the masking)."""
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.decoder import PixelEncoding
    from neural_jacobian_field_amd.model import Model
    g = {k: v.to(dev) for k, v in golden("model_transformer8").items()}
    cfg = model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                               "action_decoder": {"name": "jacobian_transformer"}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_transformer", 8), seed=0), strict=True)
    model.to(dev).eval().requires_grad_(False)
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    pos = g["final_positions"]
    dirs = g["directions"][..., None, :].expand(pos.shape).contiguous()
    dec = model.decoder.forward(pos, dirs, enc)
    c = "model_transformer8.decoder@ref-positions"
    margins(c, "action_features", dec.action_features, g["dec_action_features"], g["dec_action_features_f64"])
    margins(c, "flow", dec.flow, g["dec_flow"], g["dec_flow_f64"])
    margins(c, "density", dec.density, g["dec_density"], g["dec_density_f64"])
    with _from_reference_features(model, g):
        out = model.forward(*_inputs(g), compute_vis_features=True)
    c = "model_transformer8.forward"
    for key, got in (("rgb", out.standard_output.rgb), ("depth", out.standard_output.depth),
                     ("optical_flow", out.standard_output.optical_flow), ("vis_action_features", out.vis_output.action_features)):
        margins(c, key, got, g[key], g[key + "_f64"], self_noise=_noise(g, key, False))


def test_two_proposal_levels_vs_reference_golden(dev, golden, margins):
    """num_proposal_samples = [16, 12]: the level loop of ProposalNetworkSampler.generate_ray_samples
    (rendering/ray_samplers.py:497-552) on the fused route AND on the generic route (arbitrary density callbacks)."""
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.decoder import PixelEncoding
    from neural_jacobian_field_amd.model import Model
    g = {k: v.to(dev) for k, v in golden("model_mlp2").items()}
    cfg = model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": [16, 12], "num_nerf_samples": 10},
                               "action_decoder": {"name": "jacobian_mlp"}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", 8, num_proposal_networks=2), seed=0),
                          strict=True)
    model.to(dev).eval().requires_grad_(False)
    cam, rin, rob = _inputs(g)
    with _from_reference_features(model, g):
        out = model.forward(cam, rin, rob, compute_vis_features=True)
    c = "model_mlp2.forward"
    for key, got in (("rgb", out.standard_output.rgb), ("depth", out.standard_output.depth),
                     ("optical_flow", out.standard_output.optical_flow), ("vis_action_features", out.vis_output.action_features),
                     ("vis_weights", out.vis_output.weights)):
        margins(c, key, got, g[key], g[key + "_f64"], self_noise=_noise(g, key, False))
    # per level: weights and sample placement, fused route
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    rb = model.compute_ray_bundle(rin)
    bins, wl, bl = model.proposal_sampler.generate_ray_samples_fused(rb, list(model.proposal_networks), enc, g["z_near"],
                                                                   g["z_far"], True)
    assert len(wl) == 2 and wl[0].shape == g["prop_weights0"].shape and wl[1].shape == g["prop_weights1"].shape
    smp = [rb.samples_from_bins(b) for b in bl] + [rb.samples_from_bins(bins)]
    c = "model_mlp2.levels[fused]"
    n = lambda k: _noise(g, k, False)
    margins(c, "weights0", wl[0], g["prop_weights0"], g["prop_weights0_f64"], self_noise=n("prop_weights0"))
    margins(c, "weights1", wl[1], g["prop_weights1"], g["prop_weights1_f64"], self_noise=n("prop_weights1"))
    margins(c, "starts1", smp[1].starts, g["prop_starts1"], g["prop_starts1_f64"], self_noise=n("prop_starts1"))
    margins(c, "final_starts", smp[2].starts, g["final_starts"], g["final_starts_f64"], self_noise=n("final_starts"))
    margins(c, "final_ends", smp[2].ends, g["final_ends"], g["final_ends_f64"], self_noise=n("final_ends"))
    # generic route (reference API: density_fns callbacks)
    s_fin, pos, dirs, wl2, sl2 = model.compute_proposal(rb, enc)
    c = "model_mlp2.levels[generic]"
    margins(c, "weights0", wl2[0], g["prop_weights0"], g["prop_weights0_f64"], self_noise=n("prop_weights0"))
    margins(c, "weights1", wl2[1], g["prop_weights1"], g["prop_weights1_f64"], self_noise=n("prop_weights1"))
    margins(c, "final_starts", s_fin.starts, g["final_starts"], g["final_starts_f64"], self_noise=n("final_starts"))


# ---- flow_mlp decoder (the reference's direct-flow ablation, models/decoder/action_decoder_flow.py) ----
@pytest.fixture(scope="module")
def flow_model_and_golden(dev, golden):
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import Model
    g = golden("model_flow")
    cfg = model_cfg_from_dict({"action_dim": 5, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                               "action_decoder": {"name": "flow_mlp"}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("flow_mlp", 5), seed=0), strict=True)
    model.to(dev).eval().requires_grad_(False)
    g64 = {k + "_f64": v for k, v in golden("model_flow_f64").items()}
    return model, {k: v.to(dev) for k, v in {**g, **g64}.items()}


def test_flow_mlp_decoder_at_reference_sample_locations(flow_model_and_golden, margins):
    """The action enters the flow head as a latent input; on the fused path it is a per-image bias of the hoisted map."""
    from neural_jacobian_field_amd.decoder import PixelEncoding
    model, g = flow_model_and_golden
    enc = PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], g["action"])
    pos = g["final_positions"]
    dirs = g["directions"][..., None, :].expand(pos.shape).contiguous()
    dec = model.decoder.forward(pos, dirs, enc)
    assert rel(dec.flow[0], g["dec_flow"][0]) < 1e-4        # identity-context batch element: tight
    assert rel(dec.density[0], g["dec_density"][0]) < 1e-4
    c = "model_flow.decoder@ref-positions"                  # general pose element: the reference's float64 run is the yardstick
    margins(c, "flow", dec.flow, g["dec_flow"], g["dec_flow_f64"])
    margins(c, "color", dec.color, g["dec_color"], g["dec_color_f64"])
    margins(c, "density", dec.density, g["dec_density"], g["dec_density_f64"])
    # DecoderOutput.action_features: the flow head's 5 x 128 hidden features (action_decoder_flow.py:168-176); the values are pinned
    # against the reference in tests/test_training_gpu.py::test_flow_mlp_arm_head_and_action_mode_training_vs_reference_golden
    assert dec.action_features.shape == (*pos.shape[:3], 640) and torch.isfinite(dec.action_features).all()
    # the reference's flow_mlp.encode_image as it is (action_decoder_flow.py:246-279): a map object yielding the density alone
    only = list(model.decoder.encode_image(pos, enc))
    assert len(only) == 1 and only[0].shape == dec.density.shape and rel(only[0], dec.density) < 1e-6


def test_flow_mlp_model_forward_vs_reference_golden(flow_model_and_golden, margins):
    from neural_jacobian_field_amd.model import RobotInput
    model, g = flow_model_and_golden
    cam, rin, rob = _inputs(g)
    out = model.forward(cam, rin, rob)
    _check_forward(margins, "model_flow.forward[through encoder]", out, g, vis=False)
    # a different action on the same image: the hoisted map's feature part is cached, its action bias is not
    out0 = model.forward(cam, rin, RobotInput(torch.zeros_like(g["action"])))
    margins("model_flow.forward[zero action]", "optical_flow", out0.standard_output.optical_flow, g["optical_flow_zero_action"],
            g["optical_flow_zero_action_f64"], self_noise=_noise(g, "optical_flow", True))
    assert rel(out0.standard_output.optical_flow, g["optical_flow"]) > 1e-2   # and the two really differ
    with pytest.raises(NotImplementedError):
        model.encode_image(cam, rin, rob)


def test_hoisted_map_cache_is_not_fooled_by_recycled_memory(model_and_golden):
    """Consecutive forwards on different images (the freed feature tensor's address is typically reused by the
    allocator) must re-project the feature map."""
    from neural_jacobian_field_amd.model import CameraInput
    model, g = model_and_golden
    cam, rin, rob = _inputs(g)
    outs = []
    for k in range(3):
        img = torch.rand(g["image"].shape, generator=torch.Generator().manual_seed(50 + k)).to(g["image"].device)
        c = CameraInput(img, cam.ctxt_extrinsics, cam.ctxt_intrinsics, cam.trgt_extrinsics, cam.trgt_intrinsics)
        with torch.no_grad():
            outs.append(model.forward(c, rin, rob).standard_output.rgb.clone())
    assert rel(outs[0], outs[1]) > 1e-4 and rel(outs[1], outs[2]) > 1e-4
    with torch.no_grad():   # and the same image again reproduces its result
        img = torch.rand(g["image"].shape, generator=torch.Generator().manual_seed(50)).to(g["image"].device)
        c = CameraInput(img, cam.ctxt_extrinsics, cam.ctxt_intrinsics, cam.trgt_extrinsics, cam.trgt_intrinsics)
        assert rel(model.forward(c, rin, rob).standard_output.rgb, outs[0]) < 1e-5


def test_inverse_dynamics_least_squares(model_and_golden):
    """one fused render of the tracked rays, then Gauss-Newton.  The linearisation must reproduce
    Model.forward's optical flow for any command, and the solve must reproduce the command behind a target flow."""
    from neural_jacobian_field_amd.inverse_dynamics import linearize_flow, solve_action
    from neural_jacobian_field_amd.model import RobotInput
    model, g = model_and_golden
    cam, rin, rob = _inputs(g)
    lin = linearize_flow(model, cam, rin)
    fwd = model.forward(cam, rin, rob).standard_output.optical_flow
    assert rel(lin.optical_flow(rob.robot_action), fwd) < 1e-3  # two encoder runs: MIOpen ulps, amplified by the encoding
    # a control-step sized command (few-pixel flow on this 16-pixel image), compared in flow space
    truth = torch.randn_like(rob.robot_action) * 0.002
    target = model.forward(cam, rin, RobotInput(truth)).standard_output.optical_flow
    got = solve_action(lin, target, iterations=20)
    assert rel(lin.optical_flow(got), target) < 1e-2


def test_graphed_control_step_matches_eager(model_and_golden):
    """The control step captured as one HIP graph (encoder + fused render + LM iterations) replays to the eager
    result, also after the inputs change (new image, new target)."""
    from neural_jacobian_field_amd.inverse_dynamics import GraphedInverseDynamics, linearize_flow, solve_action
    from neural_jacobian_field_amd.model import CameraInput
    model, g = model_and_golden
    cam, rin, rob = _inputs(g)
    ctrl = GraphedInverseDynamics(model, cam, rin, iterations=6)
    gen = torch.Generator(device="cpu").manual_seed(4)
    for trial in range(2):
        image = torch.rand(cam.input_image.shape, generator=gen).to(cam.input_image.device)
        cam_t = CameraInput(image, cam.ctxt_extrinsics, cam.ctxt_intrinsics, cam.trgt_extrinsics, cam.trgt_intrinsics)
        lin = linearize_flow(model, cam_t, rin)
        target = lin.optical_flow(torch.randn_like(rob.robot_action) * 0.002)
        eager = solve_action(lin, target, iterations=6)
        got = ctrl(image, target).clone()
        assert rel(lin.optical_flow(got), target) < 1e-2
        assert rel(lin.optical_flow(got), lin.optical_flow(eager)) < 1e-2   # two encoder runs: MIOpen ulps


def test_solve_action_kernel_matches_the_tensor_restatement(dev):
    """njf_solve_action (all LM iterations in one launch) against oracle/lm_reference.py on synthetic linearisations:
    ragged ray counts beyond one 256-ray chunk, A up to 16, masks, non-zero starts; and bit-reproducible."""
    import lm_reference
    from neural_jacobian_field_amd.inverse_dynamics import FlowLinearization, solve_action
    gen = torch.Generator().manual_seed(12)
    for b, r, a in ((2, 40, 6), (1, 700, 8), (3, 256, 16), (1, 5, 2)):
        pos = (torch.rand(b, r, 3, generator=gen) * torch.tensor([1.0, 1.0, 0.5]) + torch.tensor([-0.5, -0.5, 1.5])).to(dev)
        jac = (torch.randn(b, r, 3, a, generator=gen) * 0.05).to(dev)
        ext = torch.eye(4).repeat(b, 1, 1)
        ext[:, :3, 3] = torch.randn(b, 3, generator=gen) * 0.05
        k = torch.tensor([[200.0, 0, 128], [0, 210.0, 120], [0, 0, 1]]).repeat(b, 1, 1)
        lin = FlowLinearization(pos, jac, ext.to(dev), k.to(dev))
        truth = (torch.randn(b, a, generator=gen) * 0.5).to(dev)
        target = lin.optical_flow(truth)
        mask = (torch.rand(b, r, generator=gen) > 0.2).float().to(dev)
        init = (torch.randn(b, a, generator=gen) * 0.1).to(dev)
        for kwargs in (dict(), dict(visible_mask=mask), dict(init_action=init, iterations=5)):
            got = solve_action(lin, target, **kwargs)
            ref = lm_reference.lm_solve_action(lin, target, **kwargs)
            assert torch.allclose(got, ref, atol=2e-4, rtol=1e-3), (b, r, a, kwargs.keys(), (got - ref).abs().max().item())
            assert torch.equal(got, solve_action(lin, target, **kwargs))
        if 2 * r >= a:
            assert rel(lin.optical_flow(solve_action(lin, target)), target) < 1e-3


def test_fp16_range_guard_falls_back_to_f32(model_and_golden):
    """The split-precision modes hold the leading bits of every operand in fp16: a checkpoint whose hidden activations
    exceed 65,504 must be detected and moved to the exact-fp32 MFMA path by calibrate_precision (measured on the fp32
    path's activation dumps -- the kernels' integer ReLU can hide the NaNs an overflow produces); a normal checkpoint
    keeps its precision."""
    import copy
    import warnings
    model, g = model_and_golden
    cam, rin, rob = _inputs(g)
    ranges = model.activation_range(cam, rin, rob)
    assert {"density_head", "jacobian_head", "color_head", "proposal_networks.0", "positional_encoding", "weights"} <= set(ranges)
    assert all(0 < v < 1000 for v in ranges.values()), ranges
    for prec in ("f16x2", "f16f6"):
        model.set_precision(prec)
        assert model.calibrate_precision(cam, rin, rob) == prec and model.decoder.precision == prec
    for head in ("density_head", "jacobian_head"):
        big = copy.deepcopy(model)
        with torch.no_grad():   # blow up the first hidden layer of one head: activations ~1e6
            getattr(big.decoder, head).lin_in.weight.mul_(3.0e5)
            getattr(big.decoder, head).lin_in.bias.mul_(3.0e5)
        for prec in ("f16x2", "f16f6"):
            big.set_precision(prec)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                assert big.calibrate_precision(cam, rin, rob) == "f32"
            assert any("fp16's range" in str(w.message) and head in str(w.message) for w in caught), [str(w.message) for w in caught]
            out = big.forward(cam, rin, rob).standard_output
            assert torch.isfinite(out.rgb).all() and big.decoder.precision == "f32"
    model.set_precision("f16x2")


def test_plain_checkpoint_load_never_renders_finite_but_wrong_pixels(model_and_golden):
    """INTEGRATION.md section A as written -- Model(cfg), load_state_dict, forward / patch_render, nothing else -- on a
    checkpoint whose hidden activations leave fp16's range (first layer of the density head scaled by 3e5, activations
    ~1e6).  In the fp16-carried default precision the overflow does NOT surface by itself (hi becomes inf, the integer
    ReLU can flush the NaN): the automatic range check of the first forward pass after load_state_dict must warn and move
    the model to exact fp32 products, whose output equals a model that was put on "f32" by hand, bit for bit."""
    import copy
    import warnings
    from neural_jacobian_field_amd.model import Model
    src, g = model_and_golden
    cam, rin, rob = _inputs(g)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    sd["decoder.density_head.lin_in.weight"] *= 3.0e5
    sd["decoder.density_head.lin_in.bias"] *= 3.0e5
    dev = g["image"].device
    plain = Model(copy.deepcopy(src.cfg)).to(dev).eval().requires_grad_(False)
    plain.load_state_dict(sd)                                   # the snippet of INTEGRATION.md section A ends here
    assert plain.decoder.precision == "f16f6"                   # the package default, fp16-carried
    # (every model below renders from the reference's encoder output: MIOpen's convolutions are not bit-reproducible from
    # run to run, and the comparison is bit for bit)
    with warnings.catch_warnings(record=True) as caught, _from_reference_features(plain, g):
        warnings.simplefilter("always")
        out = plain.forward(cam, rin, rob).standard_output
    assert any("fp16's range" in str(w.message) for w in caught), [str(w.message) for w in caught]
    assert plain.decoder.precision == "f32" and all(m.precision == "f32" for m in plain.proposal_networks)
    by_hand = Model(copy.deepcopy(src.cfg)).to(dev).eval().requires_grad_(False)
    by_hand.auto_range_check = False
    by_hand.load_state_dict(sd)
    by_hand.set_precision("f32")
    with _from_reference_features(by_hand, g):
        ref = by_hand.forward(cam, rin, rob).standard_output
    for a, b in ((out.rgb, ref.rgb), (out.depth, ref.depth), (out.optical_flow, ref.optical_flow)):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    # the guard is what made the difference: with it switched off the same checkpoint renders something else in f16f6
    unguarded = Model(copy.deepcopy(src.cfg)).to(dev).eval().requires_grad_(False)
    unguarded.auto_range_check = False
    unguarded.load_state_dict(sd)
    with _from_reference_features(unguarded, g):
        bad = unguarded.forward(cam, rin, rob).standard_output
    assert unguarded.decoder.precision == "f16f6"
    assert not torch.isfinite(bad.rgb).all() or rel(bad.rgb, ref.rgb) > 1e-3   # non-finite, or finite and WRONG
    # ... and a sane checkpoint is checked once, silently, and keeps the default precision
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        sane = Model(copy.deepcopy(src.cfg)).to(dev).eval().requires_grad_(False)
        sane.load_state_dict(src.state_dict())
        sane.forward(cam, rin, rob)
        checked = sane._range_checked
        sane.forward(cam, rin, rob)
    assert not caught and sane.decoder.precision == "f16f6" and checked is not None and sane._range_checked == checked


def test_empty_and_single_ray_batches(model_and_golden):
    """Edge sizes of the ray batch.  The reference cannot render an empty batch either (render_depth takes min()/max() of an
    empty tensor, model.py:277: RuntimeError): here it is refused before any launch, with the C ABI's shape error, and the
    model keeps working afterwards; ONE ray (a single wave, one half-empty tile) renders the same pixel as inside a batch."""
    from neural_jacobian_field_amd.model import RenderingInput
    model, g = model_and_golden
    cam, rin, rob = _inputs(g)
    with _from_reference_features(model, g):
        full = model.forward(cam, rin, rob).standard_output
        empty = RenderingInput(rin.origins[:, :0].contiguous(), rin.directions[:, :0].contiguous(), rin.z_near, rin.z_far)
        with pytest.raises((ValueError, RuntimeError)):
            model.forward(cam, empty, rob)
        one = RenderingInput(rin.origins[:, 5:6].contiguous(), rin.directions[:, 5:6].contiguous(), rin.z_near, rin.z_far)
        out = model.forward(cam, one, rob).standard_output
    assert out.rgb.shape == (full.rgb.shape[0], 1, 3)
    assert torch.equal(out.rgb, full.rgb[:, 5:6]) and torch.equal(out.optical_flow, full.optical_flow[:, 5:6])
    # depth is clipped with the bounds of the rays of the CALL (model.py:277), so only the un-clipped case must agree
    lo, hi = full.depth.min(), full.depth.max()
    inside = (full.depth[:, 5:6] > lo) & (full.depth[:, 5:6] < hi)
    assert torch.equal(out.depth[inside], full.depth[:, 5:6][inside])
