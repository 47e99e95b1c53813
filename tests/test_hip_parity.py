"""GPU parity tests: fused HIP path (through the C ABI) vs the CPU oracle.  Run with -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json north_star: "within 1e-4 rel fp32" (norm-wise, see oracle/parity_harness.py)


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need an MI355X")
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


import parity_harness as _ph

PRECISIONS = ["f32", "f16x2", "f16f6"]  # "f16f6": the final pass with fp6-corrected products, proposal nets on f16x2
# the truth-referenced element-wise criterion (oracle/parity_harness.py::truth_columns: the HIP output is as close to the
# float64 result as the reference's own fp32 output is) is ASSERTED for these modes and recorded for the others
# (asserted on BASELINE's frames at full size -- 2,048 rays and up to 6 M elements per tensor; on the 17 ... 96-ray cases the
#  maxima of both sides are extreme values of a few hundred elements and the ratio is recorded, not asserted)
TRUTH_ASSERTED = ("f32",)


@pytest.mark.parametrize("case_id", range(len(_ph.PARITY_CASES)))
@pytest.mark.parametrize("precision", PRECISIONS)
def test_fused_forward_matches_oracle(device, case_id, precision, margins):
    """Every MFMA precision must meet the SAME bound, max(1e-4, 2 x the reference's own fp32-vs-fp64 difference of that
    quantity) -- "f16x2" and "f16f6" are error-compensated evaluations of fp32 operands, not reduced-precision modes.
    Checked per key against the CPU oracle (per-sample quantities at identical sample locations) AND, end to end,
    against the fp32 outputs of the reference itself (tests/golden/harness_reference.npz)."""
    import parity_harness as ph
    cfg = ph.PARITY_CASES[case_id]
    rep = ph.run_parity_case(device=device, tol=TOL, precision=precision, case_id=case_id, **cfg)
    # rows 9..11 are BASELINE.json's C2 / C3 / C5 frames at full size (2,048-ray subsets): named so in the margins table
    name = ph.FULL_SIZE_CASES.get(case_id)
    tag = f"{name}[{precision}]" if name else f"parity[{case_id}:{precision}]"
    margins.record(tag, rep["rows"])
    asserted = bool(name) and precision in TRUTH_ASSERTED
    margins.record_truth(tag, rep["truth_rows"], asserted=asserted)
    assert rep["floor_source"].startswith("reference"), rep["floor_source"]
    assert rep["ok"], {k: v for k, v in rep.items() if k not in ("rows", "truth_rows")}
    if asserted:
        assert rep["truth_ok"], [r for r in rep["truth_rows"] if not r["truth_ok"]]
    # the accelerated modes (incl. the package default): their own asserted criterion on EVERY tensor of >= 1,024 elements of
    # EVERY case (oracle/parity_harness.py::truth_asserted: rms <= max(1.5 x fp32 noise, 5e-6), max <= max(2 x, 5e-5))
    if precision != "f32":
        assert rep["truth_asserted_ok"], [r for r in rep["truth_rows"] if r["asserted_ok"] is False]


@pytest.mark.parametrize("case_id", range(len(_ph.PARITY_CASES)))
def test_plain_f16_mode_within_stated_tolerance(device, case_id, margins):
    """The reduced-precision mode (precision "f16": PLAIN fp16 products, fp16 hoisted maps, every network of the frame --
    BASELINE config 5's "fp16 MFMA fused-MLP", SURVEY 8d "tolerance stated separately").  It is NOT held to north_star's 1e-4.
    Stated tolerance, per compared quantity, norm-wise like every other row:
        err <= max(2e-3, f x model),   f = 2 (tensors of >= 1,024 elements) or 4 (below: extreme values of a few rays),
                                       model = the CPU oracle with every matrix operand rounded to fp16
    (oracle/njf_oracle.py::operand_rounding, the same roundings the mode performs) against the fp32 oracle on the same case --
    i.e. the HIP path may be no further from the reference arithmetic than twice what a correct plain-fp16 evaluation is.
    Rows are compared against the CPU oracle AND the reference's own fp32 outputs (ref_* keys); truth columns (against
    float64) take the model's error as e_ref and are recorded."""
    import parity_harness as ph
    cfg = ph.PARITY_CASES[case_id]
    rep = ph.run_parity_case(device=device, tol=TOL, precision="f16", case_id=case_id, **cfg)
    name = ph.FULL_SIZE_CASES.get(case_id)
    tag = f"{name}[f16]" if name else f"parity[{case_id}:f16]"
    margins.record(tag, rep["rows"])
    margins.record_truth(tag, rep["truth_rows"], asserted=True)
    assert rep["tol"] == ph.REDUCED_TOL
    assert rep["ok"], {k: v for k, v in rep.items() if k not in ("rows", "truth_rows")}
    # round 6 (VERDICT r05 "next" #2a): the element-wise criterion against the operand-rounding model -- what a lost bit of
    # precision would break (oracle/parity_harness.py::truth_asserted, reduced modes): rms <= 1.5 x / max <= 2 x the model's error on
    # per-network and per-sample tensors; median <= 1.5 x, rms <= 2.5 x, max <= 4 x on the placement-dominated pixels
    assert rep["truth_asserted_ok"], [r for r in rep["truth_rows"] if r["asserted_ok"] is False]


@pytest.mark.parametrize("case_id", sorted(_ph.FULL_SIZE_CASES))
def test_f16_shading_with_compensated_placement_at_full_size(device, case_id, margins):
    """``set_precision("f16", proposal_precision="f16x2")`` on BASELINE's full-size frames (C2, C3, C5), against the oracle AND the
    reference's fixture outputs (VERDICT r05 "next" #2b): the proposal pass keeps fp32-class sample PLACEMENT, only the final pass
    runs in plain fp16.  The end-to-end pixels are then per-network quantities: held to the operand-rounding model of the FINAL STAGE
    at the fp32 run's sample locations with the tight factors (norm-wise max(2e-3, 2 x model); element-wise rms <= 1.5 x, max <= 2 x),
    the proposal-stage rows (prop_weights, final_bins) to the fp32 rule, and in absolute terms rgb / depth to 4e-3 and the flow to
    1e-2 of their scales (the all-fp16 mode: 7e-3 ... 3e-2 on depth / flow)."""
    import parity_harness as ph
    cfg = ph.PARITY_CASES[case_id]
    rep = ph.run_parity_case(device=device, tol=TOL, precision="f16", proposal_precision="f16x2", case_id=case_id, **cfg)
    tag = f"{ph.FULL_SIZE_CASES[case_id]}[f16+f16x2prop]"
    margins.record(tag, rep["rows"])
    margins.record_truth(tag, rep["truth_rows"], asserted=True)
    assert rep["precision"] == "f16+f16x2"
    assert rep["ok"], {k: v for k, v in rep.items() if k not in ("rows", "truth_rows")}
    assert rep["truth_asserted_ok"], [r for r in rep["truth_rows"] if r["asserted_ok"] is False]
    e = rep["errors"]
    assert e["rgb"] <= 4e-3 and e["depth"] <= 4e-3 and e["optical_flow"] <= 1e-2, e
    assert e["ref_rgb"] <= 4e-3 and e["ref_depth"] <= 4e-3 and e["ref_optical_flow"] <= 1e-2, e


def test_plain_f16_mode_is_refused_where_it_does_not_exist(device):
    """Training forwards (activation dumps) and mixed decoder codes do not exist for the plain-fp16 mode: refused loudly."""
    import parity_harness as ph
    from neural_jacobian_field_amd import hip
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import CameraInput, Model, RenderingInput, RobotInput
    with pytest.raises(ValueError, match="mixed"):
        hip.precision_code("f16", "f16x2")
    with pytest.raises(ValueError, match="mixed"):
        hip.precision_code("f16f6", "f16")
    case = ph.make_case(1, 16, 16, 8, 8, seed=0)
    cfg = model_cfg_from_dict({"action_dim": 8, "encoder": {"name": "precomputed"},
                               "rendering": {"num_proposal_samples": [32], "num_nerf_samples": 32},
                               "action_decoder": {"name": "jacobian_mlp"}})
    m = Model(cfg).to(device)
    m.load_state_dict({k: v.to(device) for k, v in case["params"].items()})
    m.set_precision("f16")
    assert m.decoder.precision == "f16" and all(p.precision == "f16" for p in m.proposal_networks)
    m.encoder.set_features(case["feats"].to(device))
    c = case["cams"]
    cam = CameraInput(None, c["ctxt_c2w"].to(device), c["ctxt_k_norm"].to(device), c["trgt_c2w"].to(device), case["k_pix"].to(device))
    rin = RenderingInput(case["origins"].to(device), case["directions"].to(device), c["z_near"].to(device), c["z_far"].to(device))
    m.train()
    with pytest.raises(RuntimeError, match="inference mode"):
        m.forward(cam, rin, RobotInput(case["action"].to(device)))
    m.eval()
    with torch.no_grad():
        out = m.forward(cam, rin, RobotInput(case["action"].to(device))).standard_output
    assert torch.isfinite(out.rgb).all() and torch.isfinite(out.optical_flow).all()


RAGGED = [
    dict(batch=3, height=16, width=20, rays=37, s_prop=40, s_final=48),                 # half-empty second tile, odd ray count
    dict(batch=1, height=16, width=16, rays=5, s_prop=33, s_final=31),                  # one sample into a tile / one short of it
    dict(batch=5, height=16, width=16, rays=1, s_prop=64, s_final=65, action_dim=3),    # one ray per image, 3 tiles, A = 3
]


@pytest.mark.parametrize("shape", range(len(RAGGED)))
@pytest.mark.parametrize("precision", PRECISIONS)
def test_ragged_shapes_match_oracle(device, shape, precision, margins):
    """Sample counts that do not fill the 32-point tiles, ray counts that do not fill the 4-wave workgroups, a batch of
    single rays: the clamped lanes of a partial tile still take part in the quad gather and the scans (the bound's floors
    come from the oracle's own float64 run here: these shapes are not in the reference-generated harness fixture)."""
    import parity_harness as ph
    rep = ph.run_parity_case(device=device, tol=TOL, precision=precision, **RAGGED[shape])
    margins.record(f"parity[ragged{shape}:{precision}]", rep["rows"])
    margins.record_truth(f"parity[ragged{shape}:{precision}]", rep["truth_rows"], asserted=False)
    assert rep["ok"], {k: v for k, v in rep.items() if k not in ("rows", "truth_rows")}


@pytest.mark.parametrize("precision", PRECISIONS)
def test_reference_initialisation_of_the_jacobian_head(device, precision, margins):
    """The reference initialises every Linear of the Jacobian head with N(0, 1e-4) weights AND biases
    (action_decoder_jacobian.py:78-83), so at the start of action-mode training the head's activations are ~1e-3 and
    its output ~1e-6: operands near the bottom of fp16's normal range.  The split-precision path must still meet the
    bound there (dominant terms keep hi and lo exact enough; measured 5e-6 on the per-sample Jacobian)."""
    import parity_harness as ph

    def reference_init(params):
        gen = torch.Generator().manual_seed(11)
        for k, v in params.items():
            if k.startswith("decoder.jacobian_head."):
                params[k] = torch.randn(v.shape, generator=gen) * 1e-4

    rep = ph.run_parity_case(device=device, tol=TOL, precision=precision, batch=2, height=16, width=16, rays=64,
                             s_prop=32, s_final=32, param_hook=reference_init)
    margins.record(f"parity[reference-init:{precision}]", rep["rows"])
    margins.record_truth(f"parity[reference-init:{precision}]", rep["truth_rows"], asserted=False)
    assert rep["ok"], {k: v for k, v in rep.items() if k not in ("rows", "truth_rows")}
    # the regime every action-mode run starts in: fp16's subnormals cut the lo halves of N(0, 1e-4) weights, the split modes are
    # ~9 x noisier than fp32 arithmetic there -- at 1.4e-6 (rms) of the Jacobian's scale, inside their asserted criterion; what
    # that does to TRAINING from this initialisation is measured by tools/ab_reference_init.py (profiles/r05_ab_reference_init.json)
    assert rep["truth_asserted_ok"], [r for r in rep["truth_rows"] if r["asserted_ok"] is False]
    assert rep["errors"]["s_jacobian"] < 2e-5, rep["errors"]


def test_fused_kernels_are_bit_reproducible(device):
    """Same inputs, two launches: every output must be bitwise identical (no atomics, no races)."""
    import parity_harness as ph
    from neural_jacobian_field_amd.renderer import RenderRequest
    case = ph.make_case(2, 16, 16, 70, 8, seed=3, identity_context=False)
    req = RenderRequest(vis=True, sample_weights=True, per_sample=True)
    a, _, _ = ph.hip_forward(case, 64, 64, device, request=req)
    b, _, _ = ph.hip_forward(case, 64, 64, device, request=req)
    c, _, _ = ph.hip_forward(case, 64, 64, device, request=req, precision="f32")
    torch.cuda.synchronize()
    assert torch.equal(a.rgb, b.rgb) and torch.equal(a.depth, b.depth) and torch.equal(a.optical_flow, b.optical_flow)
    assert torch.equal(a.bins_list[1], b.bins_list[1])
    for k in a.extras:
        assert torch.equal(a.extras[k], b.extras[k]), k
    # and the split-precision path agrees with the exact-fp32 path far inside the parity bound
    # (per-sample weights are compared loosely: the two paths resample at bins that differ by ~1e-6, which the
    #  positional encoding amplifies -- see DESIGN.md section 5)
    assert ph.rel_err(a.rgb, c.rgb) < 1e-4 and ph.rel_err(a.depth, c.depth) < 1e-4
    assert ph.rel_err(a.extras["weights"], c.extras["weights"]) < 1e-3
