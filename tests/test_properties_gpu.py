"""Size-independent properties at BASELINE.json's full size (C2: 256x256 rays, 64+64 samples), randomised checks of
the stand-alone sampler ops against the oracle, and error behaviour of the binding.  Run with -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def full_frame(dev):
    """One full C2 frame through the fused path (default precision), with per-sample outputs."""
    import parity_harness as ph
    from neural_jacobian_field_amd.renderer import RenderRequest
    case = ph.make_case(1, 256, 256, None, 8, seed=0)
    res, fr, gmap = ph.hip_forward(case, 64, 64, dev, request=RenderRequest(vis=True, sample_weights=True, per_sample=True))
    torch.cuda.synchronize()
    return case, res, fr, gmap


def test_full_size_outputs_are_sane(full_frame):
    case, res, _, _ = full_frame
    for t in [res.rgb, res.depth, res.optical_flow, *res.extras.values()]:
        assert torch.isfinite(t).all()
    w = res.extras["weights"]
    assert (w >= 0).all() and (w <= 1 + 1e-6).all()
    assert (w.sum(-1) <= 1 + 1e-5).all()                       # alpha compositing never exceeds full opacity
    assert (res.rgb >= -1e-6).all() and (res.rgb <= 1 + 1e-5).all()
    near, far = case["cams"]["z_near"].item(), case["cams"]["z_far"].item()
    assert (res.depth >= near - 1e-4).all() and (res.depth <= far + 1e-4).all()
    for bins in res.bins_list:                                   # bin edges sorted inside [0, 1]
        assert (bins[..., 1:] >= bins[..., :-1]).all() and (bins >= 0).all() and (bins <= 1).all()
    assert (res.weights_list[0] >= 0).all() and (res.weights_list[0].sum(-2) <= 1 + 1e-5).all()


def test_full_size_ray_sharding_is_exact(full_frame, dev):
    """Rays are independent units: rendering the frame in uneven shards reproduces the full-frame result bit for bit
    (this is what makes the data-parallel ray sharding of parallel.py exact)."""
    import parity_harness as ph
    case, res, _, _ = full_frame
    bounds = [0, 7, 20000, 20031, 65536]
    rgb, flow, depth = [], [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        sub = dict(case)
        sub["origins"], sub["directions"] = case["origins"][:, lo:hi].contiguous(), case["directions"][:, lo:hi].contiguous()
        # depth: compare before the tensor-global clip (per-shard bounds differ; parallel.global_depth_clip all-reduces them)
        r, fr, gmap = ph.hip_forward(sub, 64, 64, dev)
        rgb.append(r.rgb); flow.append(r.optical_flow); depth.append(r.depth)
    assert torch.equal(torch.cat(rgb, 1), res.rgb)
    assert torch.equal(torch.cat(flow, 1), res.optical_flow)
    assert torch.equal(torch.cat(depth, 1), res.depth)          # clip is inactive here: depth lies inside every shard's bounds


def test_full_size_flow_is_linear_in_the_action(full_frame, dev):
    """flow_s = J_s . a (action_decoder_jacobian.py:128-145): per-sample 3-D flow scales linearly with the command, and
    density / colour / weights do not depend on it."""
    import parity_harness as ph
    from neural_jacobian_field_amd.renderer import RenderRequest
    case, res, _, _ = full_frame
    sub = dict(case)
    sub["origins"], sub["directions"] = case["origins"][:, :4096].contiguous(), case["directions"][:, :4096].contiguous()
    req = RenderRequest(per_sample=True)
    a = ph.hip_forward(sub, 64, 64, dev, request=req)[0]
    sub2 = dict(sub); sub2["action"] = case["action"] * -2.5
    b = ph.hip_forward(sub2, 64, 64, dev, request=req)[0]
    assert torch.equal(a.extras["density"], b.extras["density"]) and torch.equal(a.rgb, b.rgb)
    assert torch.equal(a.extras["jacobian"], b.extras["jacobian"])
    scale = a.extras["sample_flow"].abs().max()
    assert ((b.extras["sample_flow"] + 2.5 * a.extras["sample_flow"]).abs().max() / scale) < 1e-5


def test_full_size_checksums_are_reproducible(full_frame, dev):
    """Eight more full frames, in both precisions, must reproduce the first one bit for bit.  This is the race
    detector of the weight stream: a wave reading an LDS chunk before another wave's DMA share has landed shows up
    as a handful of differing pixels in some frames (it did, once, before stream_step waited vmcnt(0) in front of
    the barrier)."""
    import parity_harness as ph
    case, res, _, _ = full_frame
    for _ in range(8):
        again = ph.hip_forward(case, 64, 64, dev)[0]
        assert torch.equal(again.rgb, res.rgb) and torch.equal(again.depth, res.depth)
        assert torch.equal(again.optical_flow, res.optical_flow)
    first = ph.hip_forward(case, 64, 64, dev, precision="f32")[0]
    for _ in range(3):
        again = ph.hip_forward(case, 64, 64, dev, precision="f32")[0]
        assert torch.equal(again.rgb, first.rgb) and torch.equal(again.optical_flow, first.optical_flow)


def test_sampler_ops_randomised_vs_oracle(dev):
    """alpha_weights / pdf_resample on random, ragged and degenerate inputs (zero widths, zero weights, delta pdfs,
    sample counts that are not multiples of the 32-lane tile)."""
    import njf_oracle as orc
    from neural_jacobian_field_amd import hip
    g = torch.Generator().manual_seed(7)
    for trial in range(12):
        rays = int(torch.randint(1, 70, (1,), generator=g))
        s_in = int(torch.randint(1, 257, (1,), generator=g))
        s_out = int(torch.randint(1, 300, (1,), generator=g))
        bins = torch.sort(torch.rand(rays, s_in + 1, generator=g), -1).values
        bins[:, 0], bins[:, -1] = 0.0, 1.0
        deltas = (bins[:, 1:] - bins[:, :-1]) * 9.5
        dens = torch.exp(2.0 * torch.randn(rays, s_in, generator=g))
        if trial % 3 == 0:
            deltas[:, ::5] = 0.0
            deltas[:, 1::7] = -0.1
        ref_w = orc.alpha_weights(deltas[..., None], dens[..., None])[..., 0]
        out_w = torch.empty(rays, s_in, device=dev)
        hip.alpha_weights(deltas.to(dev).contiguous(), dens.to(dev).contiguous(), out_w)
        assert ((out_w.cpu() - ref_w).abs().max() / (ref_w.abs().max() + 1e-30)) < 1e-5
        w = ref_w.clone()
        if trial % 4 == 1:
            w[: rays // 2] = 0.0                                   # all-zero rays: padding path
        if trial % 4 == 2:
            w.zero_(); w[:, s_in // 2] = 1.0                       # delta pdf
        o = torch.zeros(rays, 3); d = torch.zeros(rays, 3); d[:, 2] = 1
        near, far = torch.full((rays, 1), 0.5), torch.full((rays, 1), 10.0)
        prev = orc._samples_from_bins(o, d, near, far, bins)
        ref = orc.pdf_resample(prev, w[..., None], s_out)
        ref_bins = torch.cat([ref.spacing_starts[..., 0], ref.spacing_ends[..., -1:, 0]], -1)
        nb = s_out + 1
        u = (torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb) + 1.0 / (2 * nb)).to(dev)
        out_bins = torch.empty(rays, nb, device=dev)
        hip.pdf_resample(w.to(dev).contiguous(), bins.to(dev).contiguous(), u, s_out, 1.0, out_bins)
        # an inverse-CDF sample can flip bins when u lands within an ulp of a cdf value; compare robustly
        err = (out_bins.cpu() - ref_bins).abs()
        assert err.max() < 1e-5 or (err > 1e-5).float().mean() < 2e-3, (trial, err.max())
        assert (out_bins[:, 1:] >= out_bins[:, :-1]).all()


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("shape", [(1, 16, 16, 384), (2, 9, 13, 832), (1, 5, 7, 40)])   # ragged texel counts / channel counts
def test_feature_projection_vs_float64(dev, precision, shape):
    """The lin_z hoist G = F . Wz + bz (resnet_fc.py:138-141 moved to texels) against a float64 contraction; both MFMA
    paths must sit at fp32 rounding level (the split path splits BOTH operands on the fly)."""
    from neural_jacobian_field_amd import hip
    b, hf, wf, n = shape
    g = torch.Generator().manual_seed(hf * 100 + n)
    feats = (torch.randn(b, 512, hf, wf, generator=g) * 3).to(dev)
    wz = (torch.randn(512, n, generator=g) * 0.05).to(dev)
    bz = torch.randn(n, generator=g).to(dev)
    out = torch.empty(b, hf, wf, n, device=dev)
    hip.project_features(feats, wz, bz, out, precision=precision)
    ref = torch.einsum("bkhw,kn->bhwn", feats.double(), wz.double()) + bz.double()
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    f32 = torch.einsum("bkhw,kn->bhwn", feats, wz) + bz
    floor = ((f32.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < max(2e-6, 4 * floor), (err, floor)


@pytest.mark.parametrize("sizes", [((64, 24, 40), (64, 12, 20), (128, 6, 10), (256, 3, 5)),      # ResNet-34 ratios: 4 x 4-blocked up-sampled add
                                   ((64, 8, 16), (64, 4, 8), (128, 2, 4), (256, 1, 2)),          # ... with a level one texel high
                                   ((64, 24, 40), (64, 12, 20), (128, 6, 10), (256, 5, 7)),      # a non-2^-s level: per-texel form
                                   ((64, 18, 30), (64, 9, 15), (128, 5, 8), (256, 3, 4))])       # sides not divisible by 4: per-texel form
@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_pyramid_producer_equals_projecting_the_concatenated_map(dev, precision, sizes):
    """encoder_resnet.py:78-86 (bilinear up-sampling to the conv1 resolution + concatenation) followed by the lin_z
    hoist, against the fused producer that projects every level at its own resolution and adds the up-sampled
    projections: the same linear map, evaluated in a different order.  Both forms of the up-sampled add are covered (the
    blocked one reproduces the per-texel one bit for bit: profiles/r03_ab_variants.txt section 8)."""
    import torch.nn.functional as F
    from neural_jacobian_field_amd import hip
    g = torch.Generator().manual_seed(77)
    b, n = 2, 832
    levels = [torch.randn(b, c, h, w, generator=g).to(dev) for c, h, w in sizes]
    wz = (torch.randn(512, n, generator=g) * 0.05).to(dev)
    bz = torch.randn(n, generator=g).to(dev)
    feats = torch.cat([F.interpolate(lv, levels[0].shape[-2:], mode="bilinear", align_corners=False) for lv in levels], dim=1)
    ref = torch.einsum("bkhw,kn->bhwn", feats.double(), wz.double()) + bz.double()
    out = torch.empty(b, sizes[0][1], sizes[0][2], n, device=dev)
    hip.project_pyramid(levels, wz, bz, out, precision=precision)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 3e-6, err


def test_upsample_concat_equals_the_encoder_tail(dev):
    """njf_upsample_concat = F.interpolate(bilinear, align_corners=False) of every latent to the level-0 resolution +
    torch.cat (encoder_resnet.py:78-86), written channels-last in one pass (odd sizes, non-integer scale factors)."""
    import torch.nn.functional as F
    from neural_jacobian_field_amd import hip
    g = torch.Generator().manual_seed(78)
    levels = [torch.randn(2, c, h, w, generator=g).to(dev) for c, h, w in ((64, 24, 40), (64, 12, 20), (128, 6, 10), (256, 5, 7))]
    ref = torch.cat([F.interpolate(lv, levels[0].shape[-2:], mode="bilinear", align_corners=False) for lv in levels], dim=1)
    out = hip.upsample_concat(levels)
    assert out.shape == (2 * 24 * 40, 512)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, 512)
    assert torch.equal(out[:, :64], ref[:, :64])                  # level 0 is a pure layout change
    assert ((out - ref).abs().max() / ref.abs().max()).item() < 2e-6


@pytest.mark.parametrize("sizes", [((64, 24, 40), (64, 12, 20), (128, 6, 10), (256, 3, 5)),      # the ResNet-34 ratios 1, 2, 4, 8
                                   ((64, 24, 40), (64, 12, 20), (128, 6, 10), (256, 5, 7)),      # non-integer ratios
                                   ((64, 17, 9), (64, 9, 5), (128, 5, 3), (256, 1, 1))])         # odd sizes, a 1 x 1 level
def test_upsample_concat_backward_equals_autograd(dev, sizes):
    """njf_upsample_concat_backward = the adjoint of the encoder tail (encoder_resnet.py:78-86): against autograd through
    F.interpolate(bilinear, align_corners=False) + torch.cat (float64), for integer and non-integer size ratios; gather
    form, so two launches agree bit for bit."""
    import torch.nn.functional as F
    from neural_jacobian_field_amd import hip
    g = torch.Generator().manual_seed(sum(h * w for _, h, w in sizes))
    b = 2
    levels = [torch.randn(b, c, h, w, generator=g).to(dev).double().requires_grad_(True) for c, h, w in sizes]
    h0, w0 = sizes[0][1:]
    up = torch.cat([F.interpolate(lv, (h0, w0), mode="bilinear", align_corners=False) for lv in levels], dim=1)
    grad = torch.randn(b * h0 * w0, 512, generator=g).to(dev)                       # channels-last, as the backward pass holds it
    up.backward(grad.double().reshape(b, h0, w0, 512).permute(0, 3, 1, 2))
    got = hip.upsample_concat_backward(grad, [tuple(lv.shape) for lv in levels])
    again = hip.upsample_concat_backward(grad, [tuple(lv.shape) for lv in levels])
    for lv, o, o2 in zip(levels, got, again):
        assert o.shape == lv.shape and torch.equal(o, o2)
        err = ((o.double() - lv.grad).abs().max() / lv.grad.abs().max()).item()
        assert err < 2e-6, (tuple(lv.shape), err)
    assert torch.equal(got[0], grad.reshape(b, h0, w0, 512)[..., :64].permute(0, 3, 1, 2))   # level 0 is a pure layout change


@pytest.mark.parametrize("points,channels,texels", [(5000, 128, 331), (777, 64, 50), (3, 5, 2)])
def test_footprint_scatter_equals_index_add(dev, points, channels, texels):
    """njf_scatter_footprint = the input gradient of the bilinear sampling: four weighted index_add_ calls in one launch
    (float64 reference; heavy collisions on purpose -- many points share few texels)."""
    from neural_jacobian_field_amd import hip
    g = torch.Generator().manual_seed(points)
    grad = torch.randn(points, channels, generator=g).to(dev)
    idx = torch.randint(0, texels, (points, 4), generator=g, dtype=torch.int32).to(dev)
    w = torch.rand(points, 4, generator=g).to(dev)
    idx[1::2] = idx[0::2][: idx[1::2].shape[0]]                     # runs of equal footprints exercise the in-register merge
    out = torch.full((texels, channels), 0.5, device=dev)          # accumulates into what is there
    hip.scatter_footprint(grad, idx, w, out, run_length=7)         # ragged last run (points % 7 != 0)
    out1 = torch.full((texels, channels), 0.5, device=dev)
    hip.scatter_footprint(grad, idx, w, out1)                      # no merging: same sums
    assert ((out - out1).abs().max() / out1.abs().max()).item() < 2e-6
    ref = torch.full((texels, channels), 0.5, dtype=torch.float64, device=dev)
    for c in range(4):
        ref.index_add_(0, idx[:, c].long(), grad.double() * w[:, c:c + 1].double())
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-6, err
    with pytest.raises(ValueError):
        hip.scatter_footprint(grad, idx[:, :3].contiguous(), w, out)
    if channels % 64 == 0:   # several gradients of the same points in one launch (a strided view, like deltas[0:6:2])
        stack = torch.randn(5, points, channels, generator=g).to(dev)
        view = stack[0:5:2]
        out3 = torch.zeros(texels, 3 * channels, device=dev)
        hip.scatter_footprint(view, idx, w, out3, run_length=7)
        for s in range(3):
            ref = torch.zeros(texels, channels, dtype=torch.float64, device=dev)
            for c in range(4):
                ref.index_add_(0, idx[:, c].long(), view[s].double() * w[:, c:c + 1].double())
            got = out3[:, s * channels:(s + 1) * channels].double()
            assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-6, s


@pytest.mark.parametrize("points,channels", [(5000, 128), (513, 64), (7, 128)])
def test_relu_backward_step_equals_torch(dev, points, channels):
    """njf_relu_backward = residual + upstream * [act > 0] and its column sums (one ResnetFC backward layer step)."""
    from neural_jacobian_field_amd import hip
    g = torch.Generator().manual_seed(channels + points)
    up = torch.randn(points, channels, generator=g).to(dev)
    act = torch.relu(torch.randn(points, channels, generator=g)).to(dev)        # ReLU'd forward activation: zeros and positives
    res = torch.randn(points, channels, generator=g).to(dev)
    out, colsum = hip.relu_backward(up, act, res)
    ref = res + up * (act > 0)
    assert torch.equal(out, ref)
    assert ((colsum.double() - ref.double().sum(0)).abs().max() / ref.double().sum(0).abs().max()).item() < 1e-5
    out2, none = hip.relu_backward(up, act, None, want_colsum=False)
    assert torch.equal(out2, up * (act > 0)) and none is None
    again, colsum2 = hip.relu_backward(up, act, res)
    assert torch.equal(colsum, colsum2)                                          # deterministic reduction order


def test_binding_error_behaviour(dev):
    from neural_jacobian_field_amd import hip
    z = torch.zeros(2, 300, device=dev)
    with pytest.raises(ValueError, match="samples"):
        hip.pdf_resample(z, torch.zeros(301, device=dev), torch.zeros(9, device=dev), 8, 1.0, torch.zeros(2, 9, device=dev))
    with pytest.raises(ValueError, match="contiguous"):
        hip.alpha_weights(z.t(), z.t(), z.t())
    with pytest.raises(ValueError, match="float32"):
        hip.alpha_weights(z.double(), z.double(), z.double())
    with pytest.raises(ValueError, match="precision"):
        hip.precision_code("fp8")
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import Model
    with pytest.raises(ValueError, match="action_dim"):
        Model(model_cfg_from_dict({"action_dim": 11}))


@pytest.fixture(scope="module")
def transformer_c2(dev):
    """The C2 frame (256 x 256 image, 128 x 128 x 512 map, 64 + 64 samples, A = 8, 2,048 rays) through the ORACLE with the
    jacobian_transformer decoder: fp32, float64, and the float64 decoder + compositing at the fp32 run's sample locations --
    computed once for the three precisions of the test below."""
    import njf_oracle as orc
    import parity_harness as ph
    from neural_jacobian_field_amd import synthetic
    B, H, W, S, A, R = 1, 256, 256, 64, 8, 2048
    case = ph.make_case(B, H, W, R, A, seed=0)
    params = synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_transformer", A, with_encoder=False), seed=0)
    c = case["cams"]

    def oracle(cv):
        p = {k: cv(v) for k, v in params.items()}
        return orc.model_forward(p, features=cv(case["feats"]), ctxt_c2w=cv(c["ctxt_c2w"]), ctxt_k_norm=cv(c["ctxt_k_norm"]),
                                 trgt_c2w=cv(c["trgt_c2w"]), trgt_k_pix=cv(case["k_pix"]), origins=cv(case["origins"]),
                                 directions=cv(case["directions"]), z_near=cv(c["z_near"]), z_far=cv(c["z_far"]),
                                 action=cv(case["action"]), num_proposal_samples=[S], num_nerf_samples=S,
                                 decoder_kind="jacobian_transformer")

    ref = oracle(lambda t: t)
    ref_bins = torch.cat([ref.samples_list[1].spacing_starts[..., 0], ref.samples_list[1].spacing_ends[..., -1:, 0]], -1)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        r64 = oracle(lambda t: t.double() if t.is_floating_point() else t)
        p64 = {k: v.double() for k, v in params.items()}
        enc64 = orc.PixelEncoding(case["feats"].double(), c["ctxt_c2w"].double(), c["ctxt_k_norm"].double(), case["action"].double())
        smp64 = orc.samples_from_bins(case["origins"].double(), case["directions"].double(), c["z_near"].double(), c["z_far"].double(),
                                      ref_bins.double())
        f64 = orc.final_stage(p64, smp64, case["directions"].double(), enc64, c["trgt_c2w"].double(), case["k_pix"].double(),
                              "jacobian_transformer")
    finally:
        torch.set_default_dtype(prev)
    return dict(case=case, params=params, ref=ref, r64=r64, f64=f64, ref_bins=ref_bins, S=S, A=A)


@pytest.mark.parametrize("precision", ["f32", "f16x2", "f16f6"])
def test_transformer_head_at_full_c2_size_vs_oracle(dev, precision, margins, transformer_c2):
    """The reference's shipped Allegro decoder (jacobian_transformer, action_decoder_jacobian.py:340-446) on the C2 frame at
    full size on a 2,048-ray subset, against the CPU oracle (floors: the oracle's own fp32-vs-float64 run; the
    reference-generated goldens for this head are 16 x 16): end to end, and the per-sample Jacobian / composited action
    features / flow / density at the oracle's own final bins."""
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput
    t = transformer_c2
    case, params, ref, r64, f64, S, A = t["case"], t["params"], t["ref"], t["r64"], t["f64"], t["S"], t["A"]
    cfg = model_cfg_from_dict({"action_dim": A, "encoder": {"name": "precomputed"},
                               "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                               "action_decoder": {"name": "jacobian_transformer"}})
    model = Model(cfg).to(dev).eval().requires_grad_(False)
    model.load_state_dict({k: v.to(dev) for k, v in params.items()}, strict=True)
    model.set_precision(precision)
    model.encoder.set_features(case["feats"].to(dev))
    c = case["cams"]
    d = lambda x: x.to(dev)
    cam = CameraInput(None, d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(c["trgt_c2w"]), d(case["k_pix"]))
    rin = RenderingInput(d(case["origins"]), d(case["directions"]), d(c["z_near"]), d(c["z_far"]))
    rob = RobotInput(d(case["action"]))
    out = model.forward(cam, rin, rob, compute_vis_features=True)
    tag = f"C2@full[transformer head:{precision}]"
    ta = precision == "f32"   # the truth-referenced element-wise criterion is asserted for the exact-fp32-product mode
    margins(tag, "rgb", out.standard_output.rgb, ref.rgb, r64.rgb, truth_assert=ta)
    margins(tag, "depth", out.standard_output.depth, ref.depth, r64.depth, truth_assert=ta)
    margins(tag, "optical_flow", out.standard_output.optical_flow, ref.optical_flow, r64.optical_flow, truth_assert=ta)
    with torch.no_grad():   # the head itself at identical sample locations (the oracle's final bins injected)
        outs, *_ = model._fused_render(cam, rin, rob, model._encode_for_render(None), want_lists=False, want_vis=True,
                                       want_samples=True, final_bins=t["ref_bins"].to(dev))
    margins(tag, "s_jacobian", outs["jacobian"], ref.jacobian, f64.jacobian, truth_assert=ta)
    margins(tag, "s_action_features", outs["action_features"], ref.action_features, f64.action_features, truth_assert=ta)
    margins(tag, "s_optical_flow", outs["flow"], ref.optical_flow, f64.optical_flow, truth_assert=ta)
    margins(tag, "s_density", outs["density"], ref.density, f64.density, truth_assert=ta)


def test_joint_hoist_equals_the_per_network_maps(dev):
    """Model._joint_hoist (ONE projection for the proposal nets and the decoder of a frame) writes, channel range by channel
    range, exactly the maps the networks produce on their own (same products in the same order: bit for bit) -- from a feature
    tensor and from the encoder's latents (pyramid producer), for the MLP and the transformer decoder; and a frame rendered
    through it equals the frame rendered from the per-network maps."""
    import parity_harness as ph
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.encoder import FeaturePyramid
    from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(2, 512, 12, 20, generator=g).to(dev)
    levels = [torch.randn(2, c, h, w, generator=g).to(dev) for c, h, w in ((64, 12, 20), (64, 6, 10), (128, 3, 5), (256, 2, 3))]
    case = ph.make_case(2, 24, 40, 64, 8, seed=1, identity_context=False)
    for name in ("jacobian_mlp", "jacobian_transformer"):
        cfg = model_cfg_from_dict({"action_dim": 8, "encoder": {"name": "precomputed"},
                                   "rendering": {"num_proposal_samples": [16, 12], "num_nerf_samples": 10},
                                   "action_decoder": {"name": name}})
        model = Model(cfg).to(dev).eval().requires_grad_(False)
        model.load_state_dict({k: v.to(dev) for k, v in synthetic.seeded_state_dict(synthetic.model_shapes(name, 8, num_proposal_networks=2, with_encoder=False), seed=0).items()})
        for f in (feats, FeaturePyramid(levels)):
            gmap, prop_bases, dec_base = model._joint_hoist(f)
            nets = [*model.proposal_networks, model.decoder]
            for net, base in zip(nets, [*prop_bases, dec_base]):
                own = net.hoisted_map(f)
                assert torch.equal(gmap[..., base:base + own.shape[-1]], own), (name, type(f).__name__, base)
            assert gmap.shape[-1] == sum(n.hoisted_map(f).shape[-1] for n in nets)
        # a frame through the joint map == the frame from the per-network maps
        c = case["cams"]
        d = lambda t: t.to(dev)
        cam = CameraInput(None, d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(c["trgt_c2w"]), d(case["k_pix"]))
        rin = RenderingInput(d(case["origins"]), d(case["directions"]), d(c["z_near"]), d(c["z_far"]))
        rob = RobotInput(d(case["action"]))
        model.encoder.set_features(feats)
        a = model.forward(cam, rin, rob).standard_output
        model._joint_hoist = lambda f: None
        model.reset_image_cache()
        b = model.forward(cam, rin, rob).standard_output
        del model._joint_hoist
        assert torch.equal(a.rgb, b.rgb) and torch.equal(a.depth, b.depth) and torch.equal(a.optical_flow, b.optical_flow)
        # point queries on the image of a frame read THEIR channel range of the frame's joint map -- no second projection,
        # no duplicate map -- and return bit for bit what they return from a map of their own
        from neural_jacobian_field_amd.decoder import PixelEncoding
        enc = PixelEncoding(feats, d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(case["action"]))
        xyz = (torch.rand(2, 5, 7, 3, generator=g) * torch.tensor([1.0, 1.0, 2.0]) + torch.tensor([-0.5, -0.5, 1.0])).to(dev)
        own = (model.compute_density(xyz.reshape(2, 35, 3), enc)[0].density, model.proposal_networks[1].get_density(xyz, enc))
        model.reset_image_cache()
        for net in (model.decoder, *model.proposal_networks):
            net._hoist.gmap = net._hoist.features = None
        model.forward(cam, rin, rob)                                   # the frame: ONE joint projection
        assert model._joint_lookup(feats) is not None
        shared = (model.compute_density(xyz.reshape(2, 35, 3), enc)[0].density, model.proposal_networks[1].get_density(xyz, enc))
        assert all(net._hoist.gmap is None for net in (model.decoder, *model.proposal_networks))   # nothing projected again
        assert torch.equal(own[0], shared[0]) and torch.equal(own[1], shared[1])
        # ... and a DEEP COPY of the model with other weights never reads this model's map (the record a network holds is plain
        # data keyed on the feature tensor's identity and on the network's own packed-weights version, not a back-reference)
        import copy
        other = copy.deepcopy(model)
        with torch.no_grad():
            for prm in other.decoder.density_head.parameters():
                prm.mul_(1.5)
        theirs = other.compute_density(xyz.reshape(2, 35, 3), enc)[0].density
        assert not torch.equal(theirs, shared[0])
        assert torch.equal(model.compute_density(xyz.reshape(2, 35, 3), enc)[0].density, shared[0])


def test_sharded_frame_step_equals_the_unsharded_forward(dev):
    """parallel.ShardedFrameStep (frame-level reductions in the render kernel's epilogue, njf_reduce_frame_partials,
    njf_assemble_frame) on a ragged 3-way split rendered rank by rank on one GPU: the assembled frame equals the plain
    Model.forward of the whole frame -- rgb and flow bit for bit, depth including the tensor-global clip (model.py:277) --
    the two losses equal torch's mse on the full outputs, and both kernels equal their tensor-op restatements
    (oracle/frame_reference.py) on the same packets."""
    import frame_reference as fr
    import parity_harness as ph
    from neural_jacobian_field_amd import hip, parallel
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput
    B, H, W, S = 2, 24, 20, 48                                   # 480 rays per element: 160 + 160 + 160 would be even, so use 7 ranks
    case = ph.make_case(B, H, W, None, 8, seed=2, identity_context=False)
    cfg = model_cfg_from_dict({"action_dim": 8, "encoder": {"name": "precomputed"},
                               "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                               "action_decoder": {"name": "jacobian_mlp"}})
    model = Model(cfg).to(dev).eval().requires_grad_(False)
    model.load_state_dict({k: v.to(dev) for k, v in case["params"].items()}, strict=True)
    model.encoder.set_features(case["feats"].to(dev))
    c = case["cams"]
    d = lambda t: t.to(dev)
    cam = CameraInput(None, d(c["ctxt_c2w"]), d(c["ctxt_k_norm"]), d(c["trgt_c2w"]), d(case["k_pix"]))
    rob = RobotInput(d(case["action"]))
    o, dr = d(case["origins"]), d(case["directions"])
    R = o.shape[1]
    full = model.forward(cam, RenderingInput(o, dr, d(c["z_near"]), d(c["z_far"])), rob).standard_output
    g = torch.Generator().manual_seed(5)
    trgt_rgb, trgt_flow = torch.rand(B, R, 3, generator=g).to(dev), torch.randn(B, R, 2, generator=g).to(dev)
    world = 7
    packets, steps = [], []
    for k in range(world):
        st = parallel.ShardedFrameStep(model, B, R, dev, world_size=world, rank=k, collective=False)
        st.set_targets(trgt_rgb[:, st.lo:st.hi], trgt_flow[:, st.lo:st.hi])
        out = st.local(cam, RenderingInput(o[:, st.lo:st.hi].contiguous(), dr[:, st.lo:st.hi].contiguous(), d(c["z_near"]),
                                           d(c["z_far"])), rob)
        assert out.standard_output.rgb.data_ptr() == st.rgb.data_ptr()       # written in place: no copy, no cat
        rec = torch.empty(4, device=dev)
        fr.reduce_frame_partials(st.partials, rec)                            # kernel vs restatement on the same partials
        assert torch.equal(rec[:2], st.record[:2]) and ph.rel_err(rec[2:], st.record[2:]) < 1e-5
        packets.append(st.packets[k].clone())
        steps.append(st)
    assert model.frame_io is None
    pk = torch.stack(packets)
    frame, scal = torch.empty(B, R, 6, device=dev), torch.empty(6, device=dev)
    hip.assemble_frame(pk, B, R, frame, scal, steps[0].rgb_scale, steps[0].flow_scale)
    frame_ref, scal_ref = torch.empty_like(frame), torch.empty_like(scal)
    fr.assemble_frame(pk, B, R, frame_ref, scal_ref, steps[0].rgb_scale, steps[0].flow_scale)
    assert torch.equal(frame, frame_ref) and torch.equal(scal[:2], scal_ref[:2]) and ph.rel_err(scal[2:], scal_ref[2:]) < 1e-5
    assert torch.equal(frame[..., 0:3], full.rgb) and torch.equal(frame[..., 4:6], full.optical_flow)
    assert torch.equal(frame[..., 3:4], full.depth)                           # same global clip bounds, same values
    assert abs(scal[4].item() / torch.nn.functional.mse_loss(full.rgb, trgt_rgb).item() - 1) < 1e-5
    assert abs(scal[5].item() / (0.01 * torch.nn.functional.mse_loss(full.optical_flow, trgt_flow).item()) - 1) < 1e-5
    # the whole step on one rank (world 1), eager and as a replayed HIP graph, reproduces the frame bit for bit
    one = parallel.ShardedFrameStep(model, B, R, dev, world_size=1, rank=0)
    one.set_targets(trgt_rgb, trgt_flow)
    rin = RenderingInput(o, dr, d(c["z_near"]), d(c["z_far"]))
    f1, s1, _ = one(cam, rin, rob)
    assert torch.equal(f1, frame) and ph.rel_err(s1[2:], scal[2:]) < 1e-5
    f1 = f1.clone()
    one.capture(cam, rin, rob)
    one.frame.zero_()
    f2, s2, _ = one()
    torch.cuda.synchronize()
    assert torch.equal(f2, f1) and torch.equal(s2, s1)
    # capture()'s contract: the static inputs are refilled IN PLACE between replays.  Moving both cameras must change the
    # replayed frame exactly as it changes an eager step -- the camera inverses are launched inside the graph, not served
    # from Model's per-tensor cache filled during the warm-up (a stale w2c rendered a silently wrong frame)
    f2, s2 = f2.clone(), s2.clone()
    cam.trgt_extrinsics.copy_(cam.trgt_extrinsics @ d(ph.general_pose(41, B, scale=0.05)))
    cam.ctxt_extrinsics.copy_(cam.ctxt_extrinsics @ d(ph.general_pose(42, B, scale=0.02)))
    f3, s3, _ = one()
    f3, s3 = f3.clone(), s3.clone()
    eager = parallel.ShardedFrameStep(model, B, R, dev, world_size=1, rank=0)
    eager.set_targets(trgt_rgb, trgt_flow)
    f4, s4, _ = eager(cam, rin, rob)
    torch.cuda.synchronize()
    assert not torch.equal(f3[..., 0:3], f2[..., 0:3]) and not torch.equal(f3[..., 4:6], f2[..., 4:6])
    assert torch.equal(f3, f4) and torch.equal(s3, s4)
    with pytest.raises(ValueError, match="captured"):      # new input OBJECTS would be silently ignored by a replay: refused
        one(cam, RenderingInput(o.clone(), dr, d(c["z_near"]), d(c["z_far"])), rob)


# ---- BASELINE config 5 at its full size: 512 x 512 rays, F = [1,512,256,256], A = 6 --------------------------------------
@pytest.fixture(scope="module")
def full_frame_c5(dev):
    import parity_harness as ph
    from neural_jacobian_field_amd.renderer import RenderRequest
    case = ph.make_case(1, 512, 512, None, 6, seed=0)
    res, _, _ = ph.hip_forward(case, 64, 64, dev, request=RenderRequest(vis=True, sample_weights=True))
    torch.cuda.synchronize()
    return case, res


def test_c5_full_size_outputs_are_sane(full_frame_c5):
    case, res = full_frame_c5
    assert res.rgb.shape == (1, 512 * 512, 3) and res.extras["action_features"].shape == (1, 512 * 512, 18)
    for t in [res.rgb, res.depth, res.optical_flow, *res.extras.values()]:
        assert torch.isfinite(t).all()
    w = res.extras["weights"]
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()
    assert (res.rgb >= -1e-6).all() and (res.rgb <= 1 + 1e-5).all()
    near, far = case["cams"]["z_near"].item(), case["cams"]["z_far"].item()
    assert (res.depth >= near - 1e-4).all() and (res.depth <= far + 1e-4).all()
    for bins in res.bins_list:
        assert (bins[..., 1:] >= bins[..., :-1]).all() and (bins >= 0).all() and (bins <= 1).all()


def test_c5_full_size_ray_sharding_is_exact_and_reproducible(full_frame_c5, dev):
    """The 8-GPU partition of config 5 (32,768 rays per rank) plus a ragged split, bit for bit; and a second full frame."""
    import parity_harness as ph
    case, res = full_frame_c5
    n = 512 * 512
    for bounds in ([n * i // 8 for i in range(9)], [0, 13, 100000, 100031, n]):
        rgb, flow, depth = [], [], []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            sub = dict(case)
            sub["origins"], sub["directions"] = case["origins"][:, lo:hi].contiguous(), case["directions"][:, lo:hi].contiguous()
            r = ph.hip_forward(sub, 64, 64, dev)[0]
            rgb.append(r.rgb); flow.append(r.optical_flow); depth.append(r.depth)
        assert torch.equal(torch.cat(rgb, 1), res.rgb) and torch.equal(torch.cat(flow, 1), res.optical_flow)
        assert torch.equal(torch.cat(depth, 1), res.depth)
    again = ph.hip_forward(case, 64, 64, dev)[0]
    assert torch.equal(again.rgb, res.rgb) and torch.equal(again.depth, res.depth) and torch.equal(again.optical_flow, res.optical_flow)


def test_c5_fp6_corrected_final_pass_stays_within_the_bound(full_frame_c5, dev):
    """Config 5 is the low-precision-MFMA configuration of BASELINE.json: the default path (final pass on "f16f6", proposal
    pass on "f16x2") against "f16x2" everywhere on the full 512 x 512 frame.  Identical proposal pass => identical sample
    locations, so the difference is the fp6 correction error alone."""
    import parity_harness as ph
    case, res = full_frame_c5
    rx = ph.hip_forward(case, 64, 64, dev, precision="f16x2")[0]
    assert all(torch.equal(a, b) for a, b in zip(rx.bins_list, res.bins_list))     # same samples
    assert not torch.equal(rx.rgb, res.rgb)                                         # ... but not the same arithmetic
    assert ph.rel_err(res.rgb, rx.rgb) < 1e-4 and ph.rel_err(res.depth, rx.depth) < 1e-4
    assert ph.rel_err(res.optical_flow, rx.optical_flow) < 1e-4


# ---- config 3 at its size: four views, 128 + 128 samples per ray --------------------------------------------------------------
@pytest.fixture(scope="module")
def full_frame_c3(dev):
    import parity_harness as ph
    from neural_jacobian_field_amd.renderer import RenderRequest
    case = ph.make_case(4, 256, 256, None, 8, seed=3)
    res, _, _ = ph.hip_forward(case, 128, 128, dev, request=RenderRequest(vis=True, sample_weights=True))
    torch.cuda.synchronize()
    return case, res


def test_c3_full_size_outputs_are_sane(full_frame_c3):
    """BASELINE config 3 (B = 4 context views, 256 x 256 rays each, 128 proposal + 128 final samples: 33.5 M points per
    level, four 32-point tiles per ray and four batch elements behind one launch)."""
    case, res = full_frame_c3
    assert res.rgb.shape == (4, 256 * 256, 3) and res.extras["weights"].shape == (4, 256 * 256, 128)
    for t in [res.rgb, res.depth, res.optical_flow, *res.extras.values()]:
        assert torch.isfinite(t).all()
    w = res.extras["weights"]
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()
    assert (res.rgb >= -1e-6).all() and (res.rgb <= 1 + 1e-5).all()
    for b in range(4):
        near, far = case["cams"]["z_near"][b].item(), case["cams"]["z_far"][b].item()
        assert (res.depth[b] >= near - 1e-4).all() and (res.depth[b] <= far + 1e-4).all()
    for bins in res.bins_list:
        assert bins.shape[-1] == 129 and (bins[..., 1:] >= bins[..., :-1]).all() and (bins >= 0).all() and (bins <= 1).all()


def test_c3_full_size_batch_elements_and_ray_shards_are_independent(full_frame_c3, dev):
    """Each of the four views rendered alone, and one view in ragged ray shards, reproduce the batched launch bit for bit
    (except depth, whose clip bounds are tensor-global by the reference's definition, model.py:277: compared un-clipped
    through the weights and positions instead)."""
    import parity_harness as ph
    case, res = full_frame_c3
    for b in (0, 3):
        sub = dict(case)
        sub["cams"] = {k: (v[b:b + 1].contiguous() if torch.is_tensor(v) and v.shape[:1] == (4,) else v) for k, v in case["cams"].items()}
        for k in ("origins", "directions", "feats", "action", "k_pix"):
            sub[k] = case[k][b:b + 1].contiguous()
        r = ph.hip_forward(sub, 128, 128, dev)[0]
        assert torch.equal(r.rgb, res.rgb[b:b + 1]) and torch.equal(r.optical_flow, res.optical_flow[b:b + 1])
    n = 256 * 256
    bounds = [0, 5, 30000, 30033, n]
    rgb, flow = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        sub = dict(case)
        sub["origins"], sub["directions"] = case["origins"][:, lo:hi].contiguous(), case["directions"][:, lo:hi].contiguous()
        r = ph.hip_forward(sub, 128, 128, dev)[0]
        rgb.append(r.rgb); flow.append(r.optical_flow)
    assert torch.equal(torch.cat(rgb, 1), res.rgb) and torch.equal(torch.cat(flow, 1), res.optical_flow)


# ---- pixel-aligned sampling edge cases through the kernels' own footprint ------------------------------
def test_footprint_and_camera_transform_vs_reference_fixture(dev, golden):
    """get_pixel_aligned_features (model_components/pixel_aligned_features.py:11-35) on the reference's fixture, which
    holds points far outside the image (border padding) next to ordinary ones: every point becomes a one-sample ray, the
    training-forward dump of njf_proposal_forward returns the kernel's OWN bilinear footprint (4 texel indices + weights)
    and camera-space coordinates, and sum_c w_c F[texel_c] must equal the reference's grid_sample output."""
    from neural_jacobian_field_amd import hip
    from neural_jacobian_field_amd.renderer import pdf_u_eval
    g = golden("pixel_aligned")
    feats, xyz, c2w, k = g["feats"], g["xyz"], g["c2w"], g["k_norm"]           # [B,C,Hf,Wf], [B,N,3]
    b, n = xyz.shape[:2]
    hf, wf = feats.shape[-2:]
    near, far = torch.tensor([1.0, 0.5]), torch.tensor([3.0, 2.5])
    d = torch.tensor([0.0, 0.0, 1.0]).expand(b, n, 3).contiguous()
    o = (xyz - d * ((near + far) / 2)[:, None, None]).contiguous()           # one sample whose mid-point is the point
    f32 = dict(dtype=torch.float32, device=dev)
    cams = hip.make_cameras(torch.inverse(c2w).to(dev).contiguous(), k.to(dev).contiguous(), near.to(dev), far.to(dev))
    gmap = torch.zeros(b, hf, wf, hip.ZDIM, **f32)
    pts = b * n
    dump = {"act": torch.empty(11, pts, 128, **f32), "pe": torch.empty(pts, 64, **f32),
            "foot_idx": torch.empty(pts, 4, dtype=torch.int32, device=dev), "foot_w": torch.empty(pts, 4, **f32)}
    bins_out = torch.empty(b, n, 2, **f32)
    hip.proposal_forward(o.to(dev), d.to(dev), cams, hip.make_feature_map(gmap), 0, torch.zeros(hip.RESNET_W_FLOATS, **f32),
                         torch.zeros(hip.RESNET_B_FLOATS, **f32), torch.tensor([0.0, 1.0], device=dev), 1, pdf_u_eval(1, dev), 1,
                         1.0, bins_out, precision="f32", dump=dump)
    torch.cuda.synchronize()
    # camera-space coordinates: encoding slots 30, 31 of the first half and 30 of the second (x, y | z)
    pe = dump["pe"].cpu().reshape(b, n, 64)
    cam_xyz = torch.stack([pe[..., 30], pe[..., 31], pe[..., 62]], -1)
    ref_cam = g["out_xyz_cam"]
    assert ((cam_xyz - ref_cam).abs().max() / ref_cam.abs().max()) < 1e-6
    # bilinear footprint against grid_sample(bilinear, border, align_corners=True)
    idx, w = dump["foot_idx"].cpu().long().reshape(b, n, 4), dump["foot_w"].cpu().reshape(b, n, 4)
    flat = feats.permute(0, 2, 3, 1).reshape(b * hf * wf, -1)                  # texel-major, batch offsets included in idx
    got = (flat[idx] * w[..., None]).sum(-2)
    ref = g["out_feats"]
    assert got.shape == ref.shape
    assert ((got - ref).abs().max() / ref.abs().max()) < 1e-5
    assert (idx >= 0).all() and (idx < b * hf * wf).all() and ((w.sum(-1) - 1).abs() < 1e-6).all()
    # the fixture's out-of-image points clamp to the border: a single texel carries (almost) all of the weight
    assert w[0, 0].max() > 0.999 and w[1, 1].max() > 0.999


def test_points_tile_straddling_batch_elements(dev, golden):
    """A 32-point tile of points_kernel that spans two batch elements with DIFFERENT cameras and actions (40 points per
    element: tile 1 holds points 32..39 of element 0 and 0..23 of element 1) equals the per-element evaluation."""
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.decoder import PixelEncoding
    from neural_jacobian_field_amd.model import Model
    g = {k: v.to(dev) for k, v in golden("model_mlp").items()}
    cfg = model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                               "action_decoder": {"name": "jacobian_mlp"}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", 8), seed=0), strict=True)
    model.to(dev).eval().requires_grad_(False)
    pos = g["final_positions"][:, :4, :10].contiguous()                       # [2, 4, 10, 3]: 40 points per element
    dirs = g["directions"][:, :4, None, :].expand(pos.shape).contiguous()
    action = torch.stack([g["action"][0], -3.0 * g["action"][1] + 0.2])
    both = model.decoder.forward(pos, dirs, PixelEncoding(g["features"], g["ctxt_c2w"], g["ctxt_k_norm"], action))
    for e in range(2):
        sl = slice(e, e + 1)
        one = model.decoder.forward(pos[sl], dirs[sl], PixelEncoding(g["features"][sl].contiguous(), g["ctxt_c2w"][sl].contiguous(),
                                                                    g["ctxt_k_norm"][sl].contiguous(), action[sl].contiguous()))
        for key in ("density", "color", "flow", "action_features"):
            assert torch.equal(getattr(both, key)[sl], getattr(one, key)), (e, key)
    assert not torch.equal(both.flow[0], both.flow[1])


def test_invert_4x4_vs_lapack(dev):
    """njf_invert_4x4 (one launch) against torch.linalg.inv (six rocSOLVER launches): rigid poses, general well- and
    ill-conditioned matrices, matrices that need row exchanges; the identity is reproduced exactly."""
    from neural_jacobian_field_amd import hip
    g = torch.Generator().manual_seed(3)
    a = torch.randn(64, 4, 4, generator=g)
    rigid = torch.eye(4).repeat(64, 1, 1)
    rigid[:, :3, :3] = torch.matrix_exp(0.7 * (a[:, :3, :3] - a[:, :3, :3].transpose(1, 2)))
    rigid[:, :3, 3] = a[:, :3, 3]
    perm = torch.eye(4)[[2, 0, 3, 1]].repeat(8, 1, 1) * torch.rand(8, 1, 1, generator=g).add(0.5)   # zero pivots without exchanges
    for name, m in (("rigid", rigid), ("general", a + 3 * torch.eye(4)), ("permuted", perm)):
        got = hip.inverse(m.to(dev))
        ref = torch.linalg.inv(m.double())
        err = ((got.cpu().double() - ref).abs().amax((1, 2)) / ref.abs().amax((1, 2))).max().item()
        cond = torch.linalg.cond(m.double()).max().item()
        assert err < 2e-7 * max(cond, 1.0), (name, err, cond)
    eye = torch.eye(4, device=dev).repeat(3, 2, 1, 1)
    assert torch.equal(hip.inverse(eye), eye) and hip.inverse(eye).shape == eye.shape


@pytest.mark.parametrize("precision", ["f32", None])
@pytest.mark.parametrize("world", [8, 7])
def test_c2_frame_from_its_ray_shards_digest(dev, world, precision):
    """SURVEY 8(e) at BASELINE's size (VERDICT r05 "next" #7a): the C2 frame (256 x 256 rays, 64 + 64 samples) rendered as the
    `world` ray shards of `bench.py --gpus world` -- one after the other on this GPU, each through parallel.ShardedFrameStep exactly
    as a rank would run it (`--simulate-world`: no exchange), eager AND as a replayed HIP graph; 7 ranks = the ragged split
    (65,536 = 2 x 9,363 + 5 x 9,362) -- then assembled from the per-rank packets.  The SHA-256 of the assembled frame [B,R,6] (rgb | clipped depth
    | flow) equals the one-rank step's, in the exact-fp32 headline precision and in the package default: sharding changes nothing but
    where a ray is computed (the loss scalars agree to the rounding of their sums' order)."""
    import hashlib
    from neural_jacobian_field_amd import hip, parallel, synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput
    B, H, W, S, A = 1, 256, 256, 64, 8
    case = synthetic.synthetic_case(B, H, W, None, A, seed=0, device=dev)
    c = case["cams"]
    cfg = model_cfg_from_dict({"action_dim": A, "encoder": {"name": "precomputed"},
                               "rendering": {"num_proposal_samples": [S], "num_nerf_samples": S},
                               "action_decoder": {"name": "jacobian_mlp"}})
    model = Model(cfg).to(dev).eval().requires_grad_(False)
    model.load_state_dict({k: v.to(dev) for k, v in case["params"].items()}, strict=True)
    model.set_precision(precision or hip.DEFAULT_PRECISION)
    model.encoder.set_features(case["feats"])
    cam = CameraInput(None, c["ctxt_c2w"], c["ctxt_k_norm"], c["trgt_c2w"], case["k_pix"])
    rob = RobotInput(case["action"])
    o, dr = case["origins"], case["directions"]
    R = o.shape[1]
    g = torch.Generator().manual_seed(100)
    trgt_rgb, trgt_flow = torch.rand(B, R, 3, generator=g).to(dev), torch.randn(B, R, 2, generator=g).to(dev)

    def digest(frame):
        return hashlib.sha256(frame.detach().float().cpu().numpy().tobytes()).hexdigest()

    one = parallel.ShardedFrameStep(model, B, R, dev, world_size=1, rank=0)
    one.set_targets(trgt_rgb, trgt_flow)
    one.new_image_each_step = True
    f1, s1, _ = one(cam, RenderingInput(o, dr, c["z_near"], c["z_far"]), rob)
    f1, s1 = f1.clone(), s1.clone()
    want = digest(f1)

    for graphed in (False, True):
        packets, scales, sizes = [], None, []
        for k in range(world):
            st = parallel.ShardedFrameStep(model, B, R, dev, world_size=world, rank=k, collective=False)
            st.new_image_each_step = True
            st.set_targets(trgt_rgb[:, st.lo:st.hi], trgt_flow[:, st.lo:st.hi])
            rin = RenderingInput(o[:, st.lo:st.hi].contiguous(), dr[:, st.lo:st.hi].contiguous(), c["z_near"], c["z_far"])
            if graphed:
                st.capture(cam, rin, rob)
                st.packet.zero_()
                st()                      # replay: local step of this rank (no exchange: collective=False)
            else:
                st.local(cam, rin, rob)
            torch.cuda.synchronize()
            packets.append(st.packets[k].clone())
            sizes.append(st.hi - st.lo)
            scales = (st.rgb_scale, st.flow_scale)
        assert sum(sizes) == R and (len(set(sizes)) == 2) == (R % world != 0)      # the 7-way split is ragged, the 8-way even
        frame, scal = torch.empty(B, R, 6, device=dev), torch.empty(6, device=dev)
        hip.assemble_frame(torch.stack(packets), B, R, frame, scal, *scales)
        torch.cuda.synchronize()
        assert digest(frame) == want, ("graph" if graphed else "eager", world, precision)
        # the depth-clip bounds are exact; the two loss SUMS are folded per rank and then over ranks, i.e. in another order than the
        # one-rank step folds its workgroups: equal to fp32 rounding of a 196,608-term sum, not bit for bit
        assert torch.equal(scal[:2], s1[:2]) and torch.allclose(scal[2:], s1[2:], rtol=1e-5, atol=0)
