"""GPU tests of the training-step contract and of the frame outputs (8f #4) on the PRODUCT side,
against vectors produced by the reference's own code (tests/golden/make_golden_r02.py imports the reference's
``ModelWrapper`` / ``Model.patch_render`` / ``inference.jacobian_color_map`` for real).  Run with -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = (("perception", "perception"), ("action_dense", "action"), ("action_tracked", "action"))
RAYS = 24


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


def _batch(g, case, dev):
    batch = {"context": {}, "target": {}, "scene": {}}
    pre = f"{case}.batch."
    for k, v in g.items():
        if k.startswith(pre):
            part, name = k[len(pre):].split(".", 1)
            batch[part][name] = v.to(dev)
    return batch


@pytest.mark.parametrize("case,mode", CASES)
def test_prepare_training_input_output_vs_reference(dev, golden, margins, case, mode):
    """model_wrapper.py:446-551 on both branches -- the random pixel set shared by the batch (:458-477, the RNG is
    consumed exactly like the reference does, so one seed gives one pixel set on both sides) and the tracked-pixel gather
    (:479-507) -- including depth / z (:513-516) and the de-normalised target intrinsics (utils/convention.py:110-125)."""
    from neural_jacobian_field_amd.model_wrapper import prepare_training_input_output
    g = golden("wrapper")
    batch = _batch(g, case, dev)
    torch.manual_seed(77)
    mi, mt = prepare_training_input_output(batch, mode, RAYS)
    c = f"wrapper.prepare[{case}]"
    ref = lambda k: g[f"{case}.{k}"].to(dev)
    margins(c, "origins", mi.rendering_input.origins, ref("origins"), tol=1e-6)
    margins(c, "directions", mi.rendering_input.directions, ref("directions"), tol=1e-6)
    margins(c, "target_depth/z", mt.depth, ref("target_depth"), tol=1e-6)
    assert torch.equal(mi.rendering_input.z_near, ref("z_near")) and torch.equal(mi.rendering_input.z_far, ref("z_far"))
    assert torch.equal(mi.camera_input.trgt_intrinsics, ref("trgt_intrinsics"))
    assert torch.equal(mt.rgb, ref("target_rgb"))
    assert torch.equal(mi.camera_input.input_image, batch["context"]["rgb"])
    assert torch.equal(mi.robot_input.robot_action, batch["context"]["robot_action"])
    if mode == "perception":
        assert mt.optical_flow is None and mt.visible_mask is None
    else:
        assert torch.equal(mt.optical_flow, ref("target_flow"))
        if case == "action_tracked":
            assert torch.equal(mt.visible_mask, ref("target_mask"))
        else:
            assert mt.visible_mask is None
    assert torch.equal(batch["target"]["depth"], mt.depth)  # the reference writes depth / z back into the batch (:516)


@pytest.mark.parametrize("case,mode", CASES)
def test_loss_terms_vs_reference_training_step(dev, golden, margins, case, mode):
    """The individual loss terms the reference's training_step logs (model_wrapper.py:117-163) and their sum, evaluated by
    the PRODUCT's loss functions on the reference's own model output and targets (so the stratified jitter of the
    train-mode forward needs no replay).  rgb / ds-nerf depth / flow are reference arithmetic; interlevel and distortion
    are the un-pinned nerfstudio terms, for which the fixture holds the oracle's interval-by-interval restatement."""
    from neural_jacobian_field_amd import model_wrapper as mw
    from neural_jacobian_field_amd.model import ModelOutput, ModelStandardOutput, ModelTarget, ModelTrainingOutput
    from neural_jacobian_field_amd.ray_samplers import RaySamples
    g = {k[len(case) + 1:]: v.to(dev) for k, v in golden("wrapper").items() if k.startswith(case + ".")}
    levels = sorted(int(k[1:]) for k in g if k[0] == "w" and k[1:].isdigit())
    weights_list = [g[f"w{i}"] for i in levels]
    samples = [RaySamples(origins=None, directions=None, starts=g[f"starts{i}"], ends=g[f"ends{i}"],
                          deltas=g[f"ends{i}"] - g[f"starts{i}"], spacing_starts=g[f"sp0_{i}"], spacing_ends=g[f"sp1_{i}"])
               for i in levels]
    out = ModelOutput(ModelStandardOutput(rgb=g["out_rgb"], depth=g["out_depth"], optical_flow=g["out_flow"]),
                      ModelTrainingOutput(weights_list=weights_list, ray_samples_list=samples), None)
    target = ModelTarget(rgb=g["target_rgb"], depth=g["target_depth"], optical_flow=g.get("target_flow"),
                         visible_mask=g.get("target_mask"))
    c = f"wrapper.losses[{case}]"
    if mode == "perception":
        terms = {"loss/rgb": mw.rgb_loss(out, target), "loss/depth": mw.depth_loss(out, target, 0.001),
                 "loss/interlevel": 1.0 * mw.interlevel_loss(weights_list, samples),
                 "loss/distortion": 0.01 * mw.distortion_loss(weights_list, samples)}
    else:
        terms = {"loss/flow_loss": mw.flow_loss(out, target)}
    for name, value in terms.items():
        margins(c, name, value.reshape(1), g[name].reshape(1), tol=2e-6)
    margins(c, "loss_total", sum(terms.values()).reshape(1), g["loss_total"].reshape(1), tol=2e-6)


def test_ds_nerf_depth_loss_vs_reference_vector(dev, golden, margins):
    """utils/loss_utils.py:9-35 through the product function, on the fixture that pins the oracle's copy."""
    from neural_jacobian_field_amd.model_wrapper import ds_nerf_depth_loss
    g = {k: v.to(dev) for k, v in golden("losses").items()}
    got = ds_nerf_depth_loss(g["weights"], g["depth_target"], g["steps"], g["lengths"], torch.tensor([0.001], device=dev))
    margins("wrapper.ds_nerf_depth_loss", "loss", got.reshape(1), g["depth_loss"].reshape(1), tol=2e-6)


def test_patch_render_and_sensitivity_vs_reference(dev, golden, margins):
    """Model.patch_render (models/model.py:527-628) against the reference's own patch_render (20 rays rendered there in
    3 patches, here in one pass), and the reference's joint-sensitivity colouring (inference/jacobian_color_map.py:53-109)
    evaluated ON THE DEVICE on the rendered Jacobian field."""
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.inference import jacobian_color_map as cm
    from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput
    g = {k: v.to(dev) for k, v in golden("model_mlp").items()}
    p = {k: v.to(dev) for k, v in golden("patch_render").items()}
    noise = golden("model_mlp_f64")   # the reference's own movement under 1-ulp ray / 1e-5 feature perturbations (same scene)
    name = {"depth_raw": "depth", "flow_raw": "optical_flow", "ray_positions": "vis_ray_positions", "steps": "vis_steps",
            "ray_positions_warped": "vis_ray_positions_warped", "action_features": "vis_action_features", "weights": "vis_weights"}
    floors = lambda key: [noise[f"{kind}.{name.get(key, key)}"].item() for kind in ("floor_ulp", "floor_enc")]
    cfg = model_cfg_from_dict({"action_dim": 8, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                               "action_decoder": {"name": "jacobian_mlp"}})
    model = Model(cfg)
    model.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp", 8), seed=0), strict=True)
    model.to(dev).eval().requires_grad_(False)
    cam = CameraInput(input_image=g["image"], ctxt_extrinsics=g["ctxt_c2w"], ctxt_intrinsics=g["ctxt_k_norm"],
                      trgt_extrinsics=g["trgt_c2w"], trgt_intrinsics=g["trgt_k_pix"])
    rin = RenderingInput(g["origins"], g["directions"], g["z_near"], g["z_far"])
    ro = model.patch_render(cam, rin, RobotInput(g["action"]), render_height=4, render_width=5)
    c = "patch_render"
    for key in ("rgb", "depth_raw", "flow_raw", "ray_positions", "ray_positions_warped", "action_features", "steps", "weights"):
        got = getattr(ro, key)
        assert got.is_cuda and got.shape == p[key].shape, key
        margins(c, key, got, p[key], p[key + "_f64"], self_noise=floors(key))
    assert ro.depth_rgb.shape == (2, 4, 5, 3) and ro.depth_rgb.is_cuda
    assert ro.flow_rgb.shape == (2, 4, 5, 3) and ro.flow_rgb.dtype == torch.uint8 and ro.flow_rgb.is_cuda
    # the reference's colouring of the rendered field, on the device
    s0 = cm.compute_joint_sensitivity(ro.action_features, None, mode=0)
    s1 = cm.compute_joint_sensitivity(ro.action_features, cam.trgt_extrinsics[:, None, None, None], mode=1)
    assert s0.is_cuda and s1.is_cuda
    margins(c, "sensitivity_mode0", s0, p["sensitivity_mode0"], p["sensitivity_mode0_f64"], self_noise=floors("action_features"))
    margins(c, "sensitivity_mode1", s1, p["sensitivity_mode1"], p["sensitivity_mode1_f64"], self_noise=floors("action_features"))
    # ... and on the reference's own field: pure colouring arithmetic, tight
    r0 = cm.compute_joint_sensitivity(p["action_features"], None, mode=0)
    margins(c, "sensitivity_mode0[ref field]", r0, p["sensitivity_mode0"], tol=1e-6)
    img = cm.visualize_joint_sensitivity(r0, p["color_map"])
    ref_img = p["sensitivity_image"].cpu().numpy()
    assert img.dtype == ref_img.dtype and img.shape == ref_img.shape and abs(img.astype(int) - ref_img.astype(int)).max() <= 1


def test_joint_sensitivity_colouring_on_device_vs_reference_golden(dev, golden):
    """The CPU test of tests/test_host_cpu.py on the GPU: inference/jacobian_color_map.py against visualization.npz."""
    from neural_jacobian_field_amd.inference import jacobian_color_map as cm
    g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in golden("visualization").items()}
    s0 = cm.compute_joint_sensitivity(g["jacobians"], None, mode=0)
    s1 = cm.compute_joint_sensitivity(g["jacobians"], g["extrinsics"], mode=1)
    assert s0.is_cuda and torch.allclose(s0, g["sensitivity_mode0"], atol=1e-6) and torch.allclose(s1, g["sensitivity_mode1_ext"], atol=1e-6)
    img = cm.visualize_joint_sensitivity(s0, g["color_map"])
    ref = g["image_mode0"].cpu().numpy()
    assert img.dtype == ref.dtype and img.shape == ref.shape and abs(img.astype(int) - ref.astype(int)).max() <= 1
    p0 = cm.compute_joint_sensitivity_point_cloud(g["points"])
    assert torch.allclose(p0, g["point_sensitivity"], atol=1e-6)
    assert torch.allclose(cm.visualize_joint_sensitivity_point_cloud(p0, g["color_map"], 0), g["point_colors_mode0"], atol=1e-6)
    assert torch.allclose(cm.visualize_joint_sensitivity_point_cloud(p0, g["color_map"], 1), g["point_colors_mode1"], atol=1e-6)
