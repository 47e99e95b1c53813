"""CPU-only checks of the host logic and of the C-ABI library (no compute calls: there is no GPU here)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return g.LIB


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "njf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(njf_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 12
    lib = ctypes.CDLL(built)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/njf_hip.h but not exported"
    from neural_jacobian_field_amd import hip
    assert set(hip.EXPORTED_SYMBOLS) == set(declared)
    assert lib.njf_abi_version() == 20


def test_hoisted_channel_order(built):
    """njf_hoisted_channel (a host function: runs without a GPU) is the one definition of the hoisted map's in-block
    channel order; it must be the permutation the kernels' gather assumes (csrc/njf_device.h) for both block widths in use
    and both gather forms: F32 / F16X2 networks read logical feature 16*MB*hh + 16*m + 4*q + e at 32*m + 8*q + 4*hh + e
    (add_hoisted_latent_half), F16F6 networks accumulator register 4*e + i of block m at 16*MB*hh + 16*m + 4*i + e
    (add_hoisted_latent_quad: piece i, dword e of the lane's contiguous 16*MB floats)."""
    lib = ctypes.CDLL(built)
    F32, F16X2, F16F6 = 0, 1, 2
    for width in (128, 64):
        mb = width // 32
        for prec in (F32, F16X2):
            pos = [lib.njf_hoisted_channel(f, width, prec) for f in range(width)]
            assert sorted(pos) == list(range(width))
            for hh in range(2):
                for m in range(mb):
                    for q in range(4):
                        for e in range(4):
                            assert pos[16 * mb * hh + 16 * m + 4 * q + e] == 32 * m + 8 * q + 4 * hh + e
        pos = [lib.njf_hoisted_channel(f, width, F16F6) for f in range(width)]
        assert sorted(pos) == list(range(width))
        for hh in range(2):
            for m in range(mb):
                for e in range(4):
                    for i in range(4):
                        assert pos[16 * mb * hh + 16 * m + 4 * e + i] == 16 * mb * hh + 16 * m + 4 * i + e
        # the plain-fp16 networks (round 5) read a map of HALVES: 16-byte pieces hold 8 channels, the two lanes of a point adjacent
        # (add_hoisted_latent_f16): logical feature 16*MB*hh + 16*m + 8*q + e at 32*m + 16*q + 8*hh + e
        pos = [lib.njf_hoisted_channel(f, width, 3) for f in range(width)]
        assert sorted(pos) == list(range(width))
        for hh in range(2):
            for m in range(mb):
                for q in range(2):
                    for e in range(8):
                        assert pos[16 * mb * hh + 16 * m + 8 * q + e] == 32 * m + 16 * q + 8 * hh + e
    assert lib.njf_hoisted_channel(128, 128, 0) < 0 and lib.njf_hoisted_channel(0, 100, 0) < 0 and lib.njf_hoisted_channel(-1, 64, 0) < 0
    assert lib.njf_hoisted_channel(0, 128, 4) < 0 and lib.njf_hoisted_channel(0, 128, 0x21) < 0   # base precisions only


def test_flow_mlp_action_fold_is_exact_algebra():
    """What the fused flow_mlp path relies on (decoder.py::ActionDecoderFlowMlp): the action is constant per batch element
    and enters the flow head only through lin_z, so evaluating the head on cat[features, action] equals evaluating it on
    the features alone with lin_z's bias replaced by b + W_a a.  Checked on the CPU oracle (action_decoder_flow.py:165-183)."""
    import njf_oracle as orc
    from neural_jacobian_field_amd import synthetic
    a_dim, pts = 5, 37
    p = {k[len("decoder.flow_head."):]: v for k, v in
         synthetic.seeded_state_dict(synthetic.decoder_shapes("flow_mlp", a_dim), seed=4).items() if k.startswith("decoder.flow_head.")}
    g = torch.Generator().manual_seed(9)
    feats, pe = torch.randn(1, pts, 512, generator=g), torch.randn(1, pts, 63, generator=g)
    action = torch.randn(1, a_dim, generator=g)
    full = orc.flow_mlp({"flow_head." + k: v for k, v in p.items()}, feats, pe, action[:, None, :].expand(1, pts, a_dim))
    folded = dict(p)
    for i in range(3):
        w = p[f"lin_z.{i}.weight"]
        folded[f"lin_z.{i}.weight"] = w[:, :512].contiguous()
        folded[f"lin_z.{i}.bias"] = p[f"lin_z.{i}.bias"] + w[:, 512:] @ action[0]
    alone = orc.resnet_fc(folded, feats, pe)
    assert (full - alone).abs().max().item() < 1e-5 * full.abs().max().item()


def test_argument_validation_happens_before_any_launch(built):
    from neural_jacobian_field_amd import hip
    lib = hip.load_library()
    assert lib.njf_alpha_weights(None, None, 4, 4, None, None) == -1            # NULL pointers
    assert b"NULL" in lib.njf_error_string(-1)
    assert lib.njf_pdf_resample(1, 1, 0, 300, 1, 0, 8, 1.0, 4, 1, None) == -4    # s_in > 256
    assert lib.njf_generate_rays(None, 3, 3, 1, 1, 1, 10, 1, 1, None, None) == -2  # H*W != rays


def test_no_cpu_fallback():
    from neural_jacobian_field_amd import hip
    with pytest.raises(ValueError, match="GPU"):
        hip.alpha_weights(torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(2, 4))
    from neural_jacobian_field_amd.renderer import FusedRenderer
    with pytest.raises(ValueError, match="no CPU fallback"):
        FusedRenderer(torch.device("cpu"))


def test_environment_cannot_make_a_reduced_mode_the_default():
    """NJF_PRECISION picks the package default among the modes held to the fp32 parity bound; the reduced-precision "f16"
    (and anything unknown) is refused at import instead of being inherited silently by every process."""
    code = "import sys; sys.path.insert(0, %r); from neural_jacobian_field_amd import hip; print(hip.DEFAULT_PRECISION)" % ROOT
    for value, ok in (("f16x2", True), ("f32", True), ("f16", False), ("bf16", False)):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NJF_PRECISION=value), capture_output=True, text=True)
        assert (r.returncode == 0) == ok, (value, r.stderr[-400:])
        if ok:
            assert r.stdout.strip() == value
        else:
            assert "NJF_PRECISION" in r.stderr


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "neural-jacobian-field_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "njf_oracle" not in src and "parity_harness" not in src and "lm_reference" not in src, f


@pytest.mark.parametrize("tag,kind,adim", [("mlp", "jacobian_mlp", 8), ("transformer", "jacobian_transformer", 6),
                                           ("flow", "flow_mlp", 5)])
def test_model_state_dict_matches_reference_manifest(tag, kind, adim):
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import Model
    cfg = model_cfg_from_dict({
        "action_dim": adim,
        "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
        "action_decoder": {"name": kind},
    })
    model = Model(cfg)
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")) as f:
        ref = json.load(f)[tag]
    assert mine == ref


def test_config_from_reference_yaml_content():
    import yaml
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    text = """
action_dim: 6
rendering:
  num_proposal_samples: [ 256 ]
  num_nerf_samples: 256
  single_jitter: false
  proposal_warmup: 5000
  proposal_update_every: 5
  use_proposal_weight_anneal: true
  proposal_weights_anneal_max_num_iters: 1000
  proposal_weights_anneal_slope: 10.0
density_decoder: {name: density_mlp, mlp: {n_blocks: 5, d_hidden: 128, combine_layer: 3, combine_type: mean, beta: 0.0}}
action_decoder:
  name: jacobian_mlp
  mlp: {n_blocks: 5, d_hidden: 128, combine_layer: 3, combine_type: mean, beta: 0.0}
  num_frequencies: 10
  geometry_feature_dim: 15
  use_arm_model: False
  arm_action_dim: null
encoder: {name: resnet, use_first_pool: true, num_layers: 4, norm_type: batch, upsample_interp: bilinear}
"""
    cfg = model_cfg_from_dict(yaml.safe_load(text))
    assert cfg.action_dim == 6 and cfg.rendering.num_proposal_samples == (256,)
    assert cfg.action_decoder.name == "jacobian_mlp" and cfg.encoder.num_layers == 4
    flow = model_cfg_from_dict({"action_decoder": {"name": "flow_mlp", "num_frequncies": 10}})   # the reference's spelling
    assert flow.action_decoder.name == "flow_mlp" and flow.action_decoder.num_frequncies == 10
    with pytest.raises(KeyError):
        model_cfg_from_dict({"action_decoder": {"name": "no_such_decoder"}})


def test_anneal_schedule_matches_oracle():
    import njf_oracle as orc
    from neural_jacobian_field_amd.config import ModelCfg, RenderingCfg
    from neural_jacobian_field_amd.model import Model
    m = Model(ModelCfg(rendering=RenderingCfg(num_proposal_samples=(8,), num_nerf_samples=8)))
    for step in (0, 1, 300, 999, 5000):
        m.step_before_iter(step)
        assert abs(m.proposal_sampler._anneal - orc.anneal_value(step, 1000, 10.0)) < 1e-12


def test_shard_bounds_cover_all_rays():
    from neural_jacobian_field_amd.parallel import shard_bounds
    for n in (1, 7, 64, 65536, 65537):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, k) for k in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def _gloo_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from neural_jacobian_field_amd import parallel as par
    g = torch.Generator().manual_seed(0)
    rgb, trg = torch.rand(2, 101, 3, generator=g), torch.rand(2, 101, 3, generator=g)
    flow, tflow = torch.randn(2, 101, 2, generator=g), torch.randn(2, 101, 2, generator=g)
    depth = torch.rand(2, 101, 1, generator=g) * 12
    mm = torch.stack([torch.rand(2, 101, generator=g) + 0.5, torch.rand(2, 101, generator=g) + 9], -1)
    lo, hi = par.shard_bounds(101, world, rank)
    losses = par.sharded_losses(rgb[:, lo:hi], trg[:, lo:hi], flow[:, lo:hi], tflow[:, lo:hi])
    clipped = par.global_depth_clip(depth[:, lo:hi], mm[:, lo:hi])
    frame = par.gather_frame(clipped, 101)
    ref_rgb = torch.nn.functional.mse_loss(rgb, trg)
    ref_flow = 0.01 * torch.nn.functional.mse_loss(flow, tflow)
    ref_depth = torch.clip(depth, mm[..., 0].min(), mm[..., 1].max())
    # gradient bucket all-reduce == mean of the per-rank gradients
    lin = torch.nn.Linear(5, 3)
    for q in lin.parameters():
        q.grad = torch.full_like(q, float(rank + 1))
    par.allreduce_gradients(lin.parameters())
    grads_ok = all(torch.allclose(q.grad, torch.full_like(q, (1 + world) / 2)) for q in lin.parameters())
    # ranks that disagree on which gradients exist (rank 1 has none for the bias; nobody has one for `extra`): same bucket
    # size everywhere, zeros fill the gaps, a parameter nobody touched keeps grad = None
    lin2, extra = torch.nn.Linear(5, 3), torch.nn.Parameter(torch.zeros(4))
    lin2.weight.grad = torch.full_like(lin2.weight, float(rank + 1))
    lin2.bias.grad = torch.full_like(lin2.bias, 4.0) if rank == 0 else None
    par.allreduce_gradients([*lin2.parameters(), extra])
    grads_ok = (grads_ok and torch.allclose(lin2.weight.grad, torch.full_like(lin2.weight, (1 + world) / 2))
                and lin2.bias.grad is not None and torch.allclose(lin2.bias.grad, torch.full_like(lin2.bias, 4.0 / world))
                and extra.grad is None)
    # data_parallel_step: two ranks on different halves of a batch == one process on the whole batch
    torch.manual_seed(7)
    net, full = torch.nn.Linear(6, 2), torch.nn.Linear(6, 2)
    full.load_state_dict(net.state_dict())
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)
    half = slice(rank * 4, rank * 4 + 4)
    opt, opt_full = torch.optim.SGD(net.parameters(), lr=0.1), torch.optim.SGD(full.parameters(), lr=0.1)
    mean_loss = par.data_parallel_step(lambda: torch.nn.functional.mse_loss(net(x[half]), y[half]), net.parameters(), opt)
    opt_full.zero_grad()
    loss_full = torch.nn.functional.mse_loss(full(x), y)
    loss_full.backward()
    opt_full.step()
    step_ok = (all(torch.allclose(a, b, atol=1e-6) for a, b in zip(net.parameters(), full.parameters()))
               and abs(mean_loss - loss_full) < 1e-6)
    ok = (abs(losses["loss/rgb"] - ref_rgb) < 1e-6 and abs(losses["loss/flow_loss"] - ref_flow) < 1e-6
          and torch.equal(frame, ref_depth) and grads_ok and step_ok)
    open(os.path.join(tmp, f"ok{rank}"), "w").write(str(bool(ok)))
    dist.destroy_process_group()


def test_ray_sharding_world_size_2_gloo(tmp_path):
    """Ray-sharded loss / depth-clip / frame gather / gradient bucket / optimiser step over 2 gloo ranks equal the
    unsharded computation."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(open(os.path.join(tmp_path, f"ok{r}")).read() == "True" for r in range(2))


def _frame_step_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frame_standins
    from neural_jacobian_field_amd import parallel as par
    step, check = frame_standins.make_frame_step(par, world, rank)     # B = 2, R = 101: ragged, 51 + 50 rays
    frame, scalars, out = step(None, None, None)
    ok = out == "out" and check(frame, scalars)
    # a second step reuses every buffer (nothing is reallocated) and reproduces the first
    ptr = step.frame.data_ptr()
    frame2, scalars2, _ = step(None, None, None)
    ok = ok and frame2.data_ptr() == ptr and torch.equal(frame2, frame) and torch.equal(scalars2, scalars)
    open(os.path.join(tmp, f"fs{rank}"), "w").write(str(bool(ok)))
    dist.destroy_process_group()


def test_sharded_frame_step_world_size_2_gloo(tmp_path):
    """parallel.ShardedFrameStep over 2 gloo ranks (ragged shards): ONE all_gather of [pixels | 4 scalars] + the assemble
    step reproduce the unsharded frame, the tensor-global depth clip and the two losses.  The two HIP kernels are replaced
    by their tensor-op restatements (oracle/frame_reference.py); tests/test_properties_gpu.py checks the kernels against the
    same restatements on the GPU."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_frame_step_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(open(os.path.join(tmp_path, f"fs{r}")).read() == "True" for r in range(2))


def _run_bench(argv, env_extra=None, script="bench.py"):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, script), *argv], env=env, capture_output=True, text=True, timeout=600)


def _json_lines(text):
    rows = []
    for line in text.splitlines():
        if line.startswith("{"):
            try:
                rows.append(json.loads(line))
            except ValueError:
                pass
    return rows


def test_frame_standins_equal_the_oracle_restatements():
    """tests/frame_standins.py (what bench.py --dry-launch and the gloo tests put in place of the two frame-level kernels) holds
    its own tensor-op stand-ins so that the bench never executes oracle code; they must be the same functions as
    oracle/frame_reference.py, the checker of the real kernels on the GPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frame_reference as fr
    import frame_standins as st
    g = torch.Generator().manual_seed(4)
    world, batch, rays = 3, 2, 11                      # ragged: 4 + 4 + 3 rays
    cap = -(-rays // world)
    packets = torch.rand(world, 6 * batch * cap + 4, generator=g)
    packets[:, -4] = torch.tensor([0.7, 0.5, 0.9])
    packets[:, -3] = torch.tensor([8.0, 9.5, 9.0])
    outs = []
    for mod in (fr, st):
        frame, scal = torch.zeros(batch, rays, 6), torch.zeros(6)
        mod.assemble_frame(packets, batch, rays, frame, scal, 0.25, 0.125)
        rec = torch.zeros(4)
        mod.reduce_frame_partials(packets[:, :8].reshape(-1, 4), rec)
        outs.append((frame, scal, rec))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_bench_self_launches_its_ranks_under_gloo():
    """`python bench.py --gpus 2` as a PLAIN process (no torchrun, no WORLD_SIZE) must spawn two ranks itself and print ONE
    line whose n_gpus is 2, with the evidence that two ranks really exchanged data: backend, world size, one device record
    per rank (distinct pids), per-rank step times (VERDICT r03 "next" #1).  Driven without a GPU through --dry-launch: gloo,
    parallel.ShardedFrameStep on CPU tensors with the stand-in kernels of tests/frame_standins.py -- the launcher, the world
    check, the barriers, the max over ranks, the evidence and the line gate are bench.py's own code."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch", os.path.join(ROOT, "tests", "frame_standins.py")])
    assert r.returncode == 0, r.stderr[-2000:]
    rows = _json_lines(r.stdout)
    assert len(rows) == 1, r.stdout
    line = rows[0]
    assert line["n_gpus"] == 2 and line["dry_launch"] is True and line["value"] is None and line["frame_ok"] is True
    ev = line["rccl"]
    assert ev["backend"] == "gloo" and ev["world_size"] == 2 and ev["self_launched"] is True
    assert [d["rank"] for d in ev["devices"]] == [0, 1] and len({d["pid"] for d in ev["devices"]}) == 2
    assert len(ev["rank_step_ms"]["per_rank"]) == 2 and ev["rank_step_ms"]["max"] >= ev["rank_step_ms"]["min"] > 0


def test_bench_self_launches_eight_ranks_under_gloo():
    """The world size of BASELINE's multi-GPU configurations: `python bench.py --gpus 8` spawns eight ranks, the 101-ray frame
    shards raggedly (13 x 5 + 12 x 3), ONE all_gather per step assembles it on every rank, the line reports eight device
    records and eight step times."""
    r = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-launch", os.path.join(ROOT, "tests", "frame_standins.py")])
    assert r.returncode == 0, r.stderr[-2000:]
    rows = _json_lines(r.stdout)
    assert len(rows) == 1, r.stdout
    line = rows[0]
    assert line["n_gpus"] == 8 and line["frame_ok"] is True and line["rccl"]["world_size"] == 8
    assert [d["rank"] for d in line["rccl"]["devices"]] == list(range(8)) and len({d["pid"] for d in line["rccl"]["devices"]}) == 8
    assert len(line["rccl"]["rank_step_ms"]["per_rank"]) == 8


def test_bench_refuses_a_world_that_is_not_the_one_asked_for():
    """The case that used to print a one-GPU line for --gpus 8: a launcher environment with WORLD_SIZE=1.  Now: no JSON
    line, non-zero exit.  Same for a GPU run asked for more devices than are visible (none here)."""
    standins = os.path.join(ROOT, "tests", "frame_standins.py")
    r = _run_bench(["--gpus", "8", "--dry-launch", standins], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not _json_lines(r.stdout) and "WORLD_SIZE is 1" in r.stderr
    r = _run_bench(["--gpus", "2", "--dry-launch", standins], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not _json_lines(r.stdout)
    if not torch.cuda.is_available():
        r = _run_bench(["--gpus", "8"])
        assert r.returncode != 0 and not _json_lines(r.stdout) and "GPU(s) are visible" in r.stderr


def test_bench_train_refuses_a_wrong_world_too():
    """tools/bench_train.py (BASELINE config 4) launches through the same two calls: a launcher world that is not --gpus, or
    fewer visible GPUs than ranks, ends without a JSON line and with a non-zero status."""
    r = _run_bench(["--gpus", "8", "--mode", "action"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, script="tools/bench_train.py")
    assert r.returncode != 0 and not _json_lines(r.stdout) and "WORLD_SIZE is 1" in r.stderr
    if not torch.cuda.is_available():
        r = _run_bench(["--gpus", "2"], script="tools/bench_train.py")
        assert r.returncode != 0 and not _json_lines(r.stdout) and "GPU(s) are visible" in r.stderr


def test_line_gate_and_rank_evidence():
    """launch.check_line / launch.rank_evidence: a line whose n_gpus or evidence disagrees with --gpus is refused; a world
    of one without a process group still yields a well-formed record."""
    from neural_jacobian_field_amd import launch
    ev = launch.rank_evidence(None, torch.device("cpu"), 1.25)
    assert ev["world_size"] == 1 and ev["devices"][0]["rank"] == 0 and ev["rank_step_ms"]["per_rank"] == [1.25]
    assert launch.check_line({"n_gpus": 1, "rccl": ev}, 1)["n_gpus"] == 1
    with pytest.raises(SystemExit):
        launch.check_line({"n_gpus": 1, "rccl": ev}, 8)
    with pytest.raises(SystemExit):
        launch.check_line({"n_gpus": 8, "rccl": ev}, 8)


def test_rank_evidence_refuses_two_ranks_on_one_device(monkeypatch):
    """Two ranks that report the SAME bus id and uuid are one physical device (a mis-set LOCAL_RANK): no line.  Ranks that
    share only one of the two identifiers (partitions of one GPU) are distinct devices."""
    from neural_jacobian_field_amd import launch

    class Group:
        def __init__(self, recs):
            self.recs = recs

        def get_world_size(self):
            return len(self.recs)

        def get_backend(self):
            return "nccl"

        def all_gather_object(self, out, mine):
            for i, r in enumerate(self.recs):
                out[i] = dict(r, step_ms=1.0 + i)

    rec = lambda rank, bus, uuid: {"rank": rank, "local_rank": rank, "host": "node", "pid": 100 + rank, "device": f"cuda:{rank}",
                                   "pci_bus_id": bus, "uuid": uuid}
    monkeypatch.setattr(launch, "_device_record", lambda device: rec(0, "0000:05:00", "a"))
    dev = torch.device("cuda", 0)
    good = launch.rank_evidence(Group([rec(0, "0000:05:00", "a"), rec(1, "0000:15:00", "b")]), dev, 1.0)
    assert good["world_size"] == 2 and good["rank_step_ms"] == {"min": 1.0, "max": 2.0, "per_rank": [1.0, 2.0]}
    assert launch.rank_evidence(Group([rec(0, "0000:05:00", "a"), rec(1, "0000:05:00", "b")]), dev, 1.0)["world_size"] == 2
    with pytest.raises(SystemExit, match="distinct device"):
        launch.rank_evidence(Group([rec(0, "0000:05:00", "a"), rec(1, "0000:05:00", "a")]), dev, 1.0)
    with pytest.raises(SystemExit, match="ranks present"):
        launch.rank_evidence(Group([rec(0, "0000:05:00", "a"), rec(0, "0000:15:00", "b")]), dev, 1.0)


def test_arm_head_registration_and_mode_switch():
    """use_arm_model (action_decoder_jacobian.py:306-313, 400-407): a second ResnetFC Jacobian head under the reference's
    parameter names, selected by switch_mode (:89-90) -- for both decoders; the names equal the ones the reference registers
    (tests/golden/model_arm.npz: arm_keys); the mode switch changes which head the fused kernel evaluates (kind, hoisted
    channels); switching without the head, to an unknown mode, or with arm_action_dim != action_dim (which the reference's
    compute_flow cannot contract, :134-140) is refused."""
    import numpy as np
    from neural_jacobian_field_amd import hip, synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import Model
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_arm.npz"))
    for kind, tag, a in (("jacobian_mlp", "mlp", 8), ("jacobian_transformer", "transformer", 6)):
        cfg = lambda **dec: model_cfg_from_dict({"action_dim": a, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                                                 "action_decoder": {"name": kind, **dec}})
        m = Model(cfg(use_arm_model=True, arm_action_dim=a))
        m.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes(kind, a, arm_action_dim=a), seed=0), strict=True)
        assert sorted(k for k in m.state_dict() if "jacobian_head_arm" in k) == list(g[tag + ".arm_keys"])
        dec = m.decoder
        regular = (dec.JACOBIAN_KIND, dec.J_HOIST, dec.active_head_prefix)
        assert regular == ((hip.JACOBIAN_MLP, hip.ZDIM, "jacobian_head.") if tag == "mlp" else (hip.JACOBIAN_TRANSFORMER, hip.QDIM, ""))
        dec.switch_mode("arm")
        assert (dec.mode, dec.JACOBIAN_KIND, dec.J_HOIST, dec.active_head_prefix) == ("arm", hip.JACOBIAN_MLP, hip.ZDIM, "jacobian_head_arm.")
        dec.switch_mode("regular")
        assert (dec.JACOBIAN_KIND, dec.J_HOIST, dec.active_head_prefix) == regular
        with pytest.raises(ValueError):
            dec.switch_mode("leg")
        # the reference's action-mode freeze keeps BOTH heads trainable ("jacobian_head" is a substring of the arm head's names)
        dec.freeze_non_action_parameters()
        assert any(p.requires_grad for n, p in dec.named_parameters() if n.startswith("jacobian_head_arm."))
        assert not any(p.requires_grad for n, p in dec.named_parameters() if n.startswith(("density_head.", "color_head.")))
        plain = Model(cfg())
        assert not any("jacobian_head_arm" in k for k in plain.state_dict())
        with pytest.raises(AttributeError):
            plain.decoder.switch_mode("arm")
        other = Model(cfg(use_arm_model=True, arm_action_dim=a - 2))
        assert other.state_dict()["decoder.jacobian_head_arm.lin_out.weight"].shape[0] == 3 * (a - 2)
        with pytest.raises(ValueError, match="arm_action_dim"):
            other.decoder.switch_mode("arm")
        with pytest.raises(ValueError):
            Model(cfg(use_arm_model=True))
    # flow_mlp (action_decoder_flow.py:109-116, :122-123, :163-166): flow_head_arm under the reference's names, d_latent = 512 + arm_action_dim
    gf = np.load(os.path.join(ROOT, "tests", "golden", "model_flow_train.npz"))
    fcfg = lambda **dec: model_cfg_from_dict({"action_dim": 5, "rendering": {"num_proposal_samples": [16], "num_nerf_samples": 12},
                                              "action_decoder": {"name": "flow_mlp", **dec}})
    m = Model(fcfg(use_arm_model=True, arm_action_dim=5))
    m.load_state_dict(synthetic.seeded_state_dict(synthetic.model_shapes("flow_mlp", 5, arm_action_dim=5), seed=0), strict=True)
    assert sorted(k for k in m.state_dict() if "flow_head_arm" in k) == list(gf["arm_keys"])
    assert m.state_dict()["decoder.flow_head_arm.lin_z.0.weight"].shape == (128, 517)
    # registration order of the reference: density_head, flow_head, flow_head_arm, color_head (optimizer param groups follow it)
    tops = [k.split(".")[1] for k in m.state_dict() if k.startswith("decoder.")]
    assert [t for i, t in enumerate(tops) if i == 0 or tops[i - 1] != t] == ["density_head", "flow_head", "flow_head_arm", "color_head"]
    dec = m.decoder
    assert (dec.mode, dec.active_head_prefix, dec.kernel_action_dim) == ("regular", "flow_head.", 1)
    dec.switch_mode("arm")
    assert (dec.JACOBIAN_KIND, dec.J_HOIST, dec.active_head_prefix) == (hip.JACOBIAN_MLP, hip.ZDIM, "flow_head_arm.")
    dec.freeze_non_action_parameters()      # action_decoder_flow.py:281-288: "flow_head" is a substring of both heads' names
    assert all(p.requires_grad == n.startswith("flow_head") for n, p in dec.named_parameters())
    from neural_jacobian_field_amd import training
    for n, p in m.named_parameters():
        if "decoder" not in n:
            p.requires_grad = False
    assert training.is_action_mode(m) and training.action_kind(m) == "flow_mlp"
    names, tensors = training.action_params(m)
    assert names == ["flow_head_arm." + k for k in training.JACOBIAN_PARAM_ORDER] and tensors[-4].shape == (128, 517)
    with pytest.raises(AttributeError):
        Model(fcfg()).decoder.switch_mode("arm")
    with pytest.raises(ValueError, match="arm_action_dim"):
        Model(fcfg(use_arm_model=True, arm_action_dim=3)).decoder.switch_mode("arm")


def test_training_precision_follows_the_reference_matmul_switch():
    """training.py's "auto" settings: the TF32-class backward forms are selected by torch.set_float32_matmul_precision("high") --
    what the reference's train.py sets (train.py:64-65) -- for networks whose forward runs in a split precision, and only then."""
    from neural_jacobian_field_amd import training
    assert torch.get_float32_matmul_precision() == "highest"
    try:
        for fwd in (None, "f16f6", "f16x2", "f32"):
            assert (training.backward_precision(fwd), training.storage_precision(fwd), training.activation_dump_dtype(fwd)) == \
                ("f32", "f32", torch.float32)
        torch.set_float32_matmul_precision("high")
        for fwd in (None, "f16f6", "f16x2"):
            assert (training.backward_precision(fwd), training.storage_precision(fwd), training.activation_dump_dtype(fwd)) == \
                ("f16x2", "f16", torch.float16)
        assert (training.backward_precision("f32"), training.storage_precision("f32")) == ("f32", "f32")
        training.set_backward_precision("f32"); training.set_storage_precision("f32")     # explicit settings win
        assert (training.backward_precision("f16f6"), training.storage_precision("f16f6")) == ("f32", "f32")
        with pytest.raises(ValueError):
            training.set_backward_precision("bf16")
        with pytest.raises(ValueError):
            training.set_storage_precision("f8")
    finally:
        torch.set_float32_matmul_precision("highest")
        training.set_backward_precision("auto"); training.set_storage_precision("auto")


def test_folded_transformer_head_is_exact_algebra_and_differentiable():
    """training.folded_transformer: the folded form the fused kernels evaluate (and whose gradients njf_transformer_backward produces)
    equals the head in the reference's parameterisation (training.transformer_head = action_decoder_jacobian.py:418-446 with
    transformer.py:38-135) to float64 rounding, for A = 6 (two unused key slots per head) and A = 8 -- and gradients pushed through
    the fold's autograd graph equal autograd through the original head."""
    from neural_jacobian_field_amd import synthetic, training
    for a_dim in (6, 8):
        shapes = synthetic.decoder_shapes("jacobian_transformer", a_dim)
        p = {k[len("decoder."):]: v.double() for k, v in synthetic.seeded_state_dict(shapes, seed=3).items()
             if k.startswith("decoder.jacobian")}
        for v in p.values():
            v.requires_grad_(True)
        g = torch.Generator().manual_seed(a_dim)
        pts = 40
        xyz = torch.randn(pts, 63, generator=g, dtype=torch.float64)
        feats = torch.randn(pts, 512, generator=g, dtype=torch.float64)
        upstream = torch.randn(pts, 3 * a_dim, generator=g, dtype=torch.float64)
        ref = training.transformer_head(p, xyz, feats)
        ref_grads = torch.autograd.grad(ref, list(p.values()), upstream)
        folded = training.folded_transformer(p)
        norm = lambda t: (t - t.mean(-1, keepdim=True)) / torch.sqrt(t.var(-1, unbiased=False, keepdim=True) + 1e-5)
        x = torch.nn.functional.linear(torch.cat([xyz, feats], -1), p["jacobian_query_mlp.weight"], p["jacobian_query_mlp.bias"])
        valid = torch.arange(8) < a_dim
        for l in range(3):
            m, b = folded["mats"][l], folded["biases"][l]
            dots = (norm(x) @ m[0].t() + b[0]).reshape(pts, 8, 8).masked_fill(~valid, float("-inf"))
            x = x + torch.softmax(dots, -1).reshape(pts, 64) @ m[1].t() + b[1]
            x = x + torch.nn.functional.gelu(norm(x) @ m[2].t() + b[2]) @ m[3].t() + b[3]
        out = torch.nn.functional.linear(x, p["jacobian_head.weight"], p["jacobian_head.bias"])
        assert (out - ref).abs().max() <= 1e-12 * ref.abs().max()
        got = torch.autograd.grad(out, list(p.values()), upstream)
        for name, g_ref, g_got in zip(p, ref_grads, got):
            assert (g_got - g_ref).abs().max() <= 1e-9 * (g_ref.abs().max() + 1e-30), name
        # rows / columns of the unused key slots are structurally zero
        assert a_dim == 8 or float(folded["mats"].detach()[:, 0].reshape(3, 8, 8, 64)[:, :, a_dim:].abs().max()) == 0.0


def test_transformer_fold_on_one_flat_leaf_equals_the_per_tensor_fold_and_is_shared_within_a_step():
    """training.transformer_fold: the fold on ONE flat float64 copy of the 41 parameter tensors (three launches instead of 41
    conversions) gives the same folded matrices and -- through the flat leaf -- the same gradients as the fold on per-tensor
    float64 leaves; the entry the forward pass's pack leaves behind is the one the backward pass of the same step consumes, and a
    parameter update (version bump) invalidates it."""
    from neural_jacobian_field_amd import synthetic, training
    shapes = synthetic.decoder_shapes("jacobian_transformer", 6)
    p = {k[len("decoder."):]: v.clone().requires_grad_(True) for k, v in synthetic.seeded_state_dict(shapes, seed=5).items()
         if k.startswith("decoder.jacobian")}
    names, params = list(p), list(p.values())
    leaves = [t.detach().double().requires_grad_(True) for t in params]
    ref = training.folded_transformer(dict(zip(names, leaves)))
    g = torch.Generator().manual_seed(1)
    up_m = torch.randn(ref["mats"].shape, generator=g, dtype=torch.float64)
    up_b = torch.randn(ref["biases"].shape, generator=g, dtype=torch.float64)
    ref_grads = torch.autograd.grad([ref["mats"], ref["biases"]], leaves, [up_m, up_b], allow_unused=True)
    training._fold_cache.clear()
    with torch.no_grad():                                       # the pack runs under no_grad; the graph is built all the same
        flat, folded = training.transformer_fold(names, params)
    assert flat.requires_grad and folded["mats"].requires_grad
    assert torch.equal(folded["mats"].detach(), ref["mats"].detach()) and torch.equal(folded["biases"].detach(), ref["biases"].detach())
    flat2, folded2 = training.transformer_fold(names, params, consume=True)       # the backward pass of the same step
    assert flat2 is flat and folded2 is folded and not training._fold_cache
    (g_flat,) = torch.autograd.grad([folded["mats"], folded["biases"]], [flat], [up_m, up_b])
    for name, t, g_ref, g_got in zip(names, params, ref_grads, g_flat.split([t.numel() for t in params])):
        if g_ref is None:
            assert float(g_got.abs().max()) == 0.0, name
        else:
            assert torch.equal(g_got.reshape(t.shape), g_ref), name
    with torch.no_grad():
        training.transformer_fold(names, params)
        params[0].add_(1.0)                                      # an optimiser step
    flat3, _ = training.transformer_fold(names, params, consume=True)
    assert flat3 is not flat and not training._fold_cache
    frozen = [t.detach() for t in params]
    flat4, folded4 = training.transformer_fold(names, frozen)
    assert not flat4.requires_grad and not folded4["mats"].requires_grad and not training._fold_cache


def test_resnetfc_backward_latent_constant_columns_are_exact_algebra():
    """What flow_mlp's training relies on (training.resnetfc_backward, ``latent_constants``): with z = cat[f, a] and a constant per
    batch element, d lin_z.weight[:, C:] = sum_b a[b] (x) sum_{p in b} delta[p] -- checked against autograd of the plain formula."""
    g = torch.Generator().manual_seed(0)
    nb, pts, c, a_dim = 3, 7, 6, 4
    feats = torch.randn(nb, pts, c, generator=g, dtype=torch.float64)
    action = torch.randn(nb, a_dim, generator=g, dtype=torch.float64)
    w = torch.randn(5, c + a_dim, generator=g, dtype=torch.float64, requires_grad=True)
    upstream = torch.randn(nb, pts, 5, generator=g, dtype=torch.float64)
    z = torch.cat([feats, action[:, None, :].expand(nb, pts, a_dim)], dim=-1)
    ((z @ w.t()) * upstream).sum().backward()
    per_image = upstream.sum(1)                                            # [B,5]: sum over the points of an image
    assert torch.allclose(w.grad[:, c:], torch.einsum("bf,ba->fa", per_image, action), rtol=1e-12, atol=1e-12)


def _synthetic_linearization(device="cpu"):
    from neural_jacobian_field_amd.inverse_dynamics import FlowLinearization
    gen = torch.Generator().manual_seed(3)
    b, r, a = 2, 40, 6
    pos = torch.rand(b, r, 3, generator=gen) * torch.tensor([1.0, 1.0, 0.5]) + torch.tensor([-0.5, -0.5, 1.5])
    jac = torch.randn(b, r, 3, a, generator=gen) * 0.05
    ext = torch.eye(4).repeat(b, 1, 1)
    ext[1, :3, 3] = torch.tensor([0.1, -0.05, 0.02])
    k = torch.tensor([[200.0, 0, 128], [0, 200.0, 128], [0, 0, 1]]).repeat(b, 1, 1)
    truth = torch.randn(b, a, generator=gen)
    return FlowLinearization(pos.to(device), jac.to(device), ext.to(device), k.to(device)), truth.to(device)


def test_inverse_dynamics_lm_restatement_recovers_action():
    """The tensor-op restatement of the Levenberg-Marquardt solve (oracle/lm_reference.py, the checker of
    njf_solve_action) on a synthetic linearisation: flow is linear in the command up to the perspective divide, so the
    solve from zero must land on the generating command; masked rays must not influence it.  The product's
    solve_action is HIP-only and refuses CPU tensors."""
    import lm_reference
    from neural_jacobian_field_amd.inverse_dynamics import solve_action
    lin, truth = _synthetic_linearization()
    target = lin.optical_flow(truth)
    got = lm_reference.lm_solve_action(lin, target, iterations=6)
    assert torch.allclose(got, truth, atol=2e-3), (got - truth).abs().max()
    mask = torch.ones(target.shape[:2])
    mask[:, ::3] = 0
    corrupted = target.clone()
    corrupted[:, ::3] += 50.0
    got = lm_reference.lm_solve_action(lin, corrupted, iterations=6, visible_mask=mask)
    assert torch.allclose(got, truth, atol=2e-3)
    with pytest.raises(ValueError, match="GPU"):
        solve_action(lin, target)


def test_proposal_losses_match_the_loop_oracle():
    """interlevel / distortion losses (nerfstudio is absent and un-pinned: parity unpinned) -- the vectorised host
    expressions against the oracle's interval-by-interval restatement, including gradients."""
    import njf_oracle as orc
    from neural_jacobian_field_amd.model_wrapper import distortion_loss, interlevel_loss
    from neural_jacobian_field_amd.ray_samplers import RaySamples
    gen = torch.Generator().manual_seed(5)
    n, s_p, s_f = 5, 9, 7

    def level(s):
        edges = torch.sort(torch.rand(n, s + 1, generator=gen), dim=-1).values
        edges[:, 0], edges[:, -1] = 0.0, 1.0
        w = torch.rand(n, s, generator=gen)
        w = (w / w.sum(-1, keepdim=True) * 0.9).requires_grad_(True)
        smp = RaySamples(None, None, edges[:, :-1, None], edges[:, 1:, None], None, edges[:, :-1, None], edges[:, 1:, None])
        return edges, w, smp

    (e_p, w_p, smp_p), (e_f, w_f, smp_f) = level(s_p), level(s_f)
    got_i = interlevel_loss([w_p[..., None], w_f[..., None]], [smp_p, smp_f])
    got_d = distortion_loss([w_p[..., None], w_f[..., None]], [smp_p, smp_f])
    g_i, = torch.autograd.grad(got_i, w_p)
    g_d, = torch.autograd.grad(got_d, w_f)
    w_p2, w_f2 = w_p.detach().clone().requires_grad_(True), w_f.detach().clone().requires_grad_(True)
    ref_i = orc.interlevel_loss([w_p2, w_f2], [e_p, e_f])
    ref_d = orc.distortion_loss(w_f2, e_f)
    r_i, = torch.autograd.grad(ref_i, w_p2)
    r_d, = torch.autograd.grad(ref_d, w_f2)
    assert got_i.item() > 0 and torch.allclose(got_i, ref_i, rtol=1e-5, atol=1e-8)
    assert torch.allclose(got_d, ref_d, rtol=1e-5, atol=1e-8)
    assert torch.allclose(g_i, r_i, rtol=1e-4, atol=1e-8) and torch.allclose(g_d, r_d, rtol=1e-4, atol=1e-8)


def test_distortion_loss_prefix_form_equals_the_outer_form():
    """model_wrapper.distortion_loss evaluates nerfstudio's pairwise term sum_ij w_i w_j |m_i - m_j| with two prefix sums (O(S) per
    ray, float64) instead of the [R,S,S] outer difference.  Against the literal outer form evaluated in FLOAT64 (the truth of both):
    value and gradient, uniform weights and weights concentrated on a surface (where m_i W_i - M_i cancels), equal mid-points
    (zero-width bins) included; the fp32 outer form must not be closer to that truth than the prefix form is."""
    from neural_jacobian_field_amd.model_wrapper import distortion_loss, distortion_loss_outer
    from neural_jacobian_field_amd.ray_samplers import RaySamples
    gen = torch.Generator().manual_seed(11)
    n, s = 6, 48
    edges = torch.sort(torch.rand(n, s + 1, generator=gen), dim=-1).values
    edges[:, 0], edges[:, -1] = 0.0, 1.0
    edges[0, 10:14] = edges[0, 10]                                          # zero-width bins: equal mid-points
    spread = torch.rand(n, s, generator=gen)
    peak = torch.exp(-0.5 * ((torch.arange(s)[None] - 30.0) / 0.6) ** 2) + 1e-6      # a surface: almost all weight in two bins
    perm = torch.stack([torch.randperm(s, generator=gen) for _ in range(n)])            # samples of a ray in arbitrary order
    for raw, shuffled in ((spread, False), (peak.expand(n, s) * (1 + 0.1 * spread), False), (spread, True)):
        w0 = (raw / raw.sum(-1, keepdim=True) * 0.95)

        def run(fn, dtype):
            w = w0.to(dtype).clone().requires_grad_(True)
            e = edges.to(dtype)
            lo, hi = e[:, :-1], e[:, 1:]
            if shuffled:
                lo, hi = torch.gather(lo, -1, perm), torch.gather(hi, -1, perm)
            smp = RaySamples(None, None, lo[..., None], hi[..., None], None, lo[..., None], hi[..., None])
            loss = fn([w[..., None]], [smp])
            g, = torch.autograd.grad(loss, w)
            return loss.detach().double(), g.detach().double()

        truth, g_truth = run(distortion_loss_outer, torch.float64)
        got, g_got = run(distortion_loss, torch.float32)
        outer32, g_outer32 = run(distortion_loss_outer, torch.float32)
        assert abs(got - truth) <= 1e-6 * abs(truth) + 1e-12, (got, truth)       # (fp32 mid-points, fp32 intra term and mean)
        scale = g_truth.abs().max()
        assert (g_got - g_truth).abs().max() <= 1e-6 * scale, ((g_got - g_truth).abs().max(), scale)
        assert (g_got - g_truth).abs().max() <= (g_outer32 - g_truth).abs().max() + 2e-7 * scale
        assert abs(got - truth) <= abs(outer32 - truth) + 5e-7 * abs(truth)


def test_ctypes_structs_match_the_header():
    """Field names and order of every struct in include/njf_hip.h against the ctypes mirrors (hip.py) and the
    reference-side stub printed in INTEGRATION.md -- a missing trailing field would make the library read past the
    caller's struct."""
    from neural_jacobian_field_amd import hip
    header = open(os.path.join(ROOT, "include", "njf_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                names += [n.strip().lstrip("*").strip() for n in re.sub(r"^(const\s+)?\w+\s*\*?", "", decl, count=1).split(",")]
        return names

    pairs = {"NjfRenderOutputs": hip.RenderOutputs, "NjfActivationDump": hip.ActivationDump, "NjfCameras": hip.Cameras,
             "NjfFeatureMap": hip.FeatureMap, "NjfPyramidLevel": hip.PyramidLevel}
    for name, mirror in pairs.items():
        assert fields(name) == [f[0] for f in mirror._fields_], name
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class RenderOutputs\(C\.Structure\):.*?_fields_ = \[\(n, vp\) for n in \((.*?)\)\]( \+ \[(.*?)\])?", doc, flags=re.S)
    names = re.findall(r'"(\w+)"', m.group(1)) + re.findall(r'"(\w+)"', m.group(3) or "")   # pointer fields + trailing int fields
    assert names == fields("NjfRenderOutputs")


def test_jacobian_colour_mapping_matches_reference_golden(golden):
    """inference/jacobian_color_map.py (reference-owned arithmetic): sensitivity maps, colour mixing, point-cloud
    variants and the colour tables against vectors produced by the reference itself."""
    from neural_jacobian_field_amd.inference import jacobian_color_map as cm
    g = golden("visualization")
    s0 = cm.compute_joint_sensitivity(g["jacobians"], None, mode=0)
    s1 = cm.compute_joint_sensitivity(g["jacobians"], g["extrinsics"], mode=1)
    assert torch.allclose(s0, g["sensitivity_mode0"], atol=1e-6) and torch.allclose(s1, g["sensitivity_mode1_ext"], atol=1e-6)
    img = cm.visualize_joint_sensitivity(s0, g["color_map"])
    ref = g["image_mode0"].numpy() if hasattr(g["image_mode0"], "numpy") else g["image_mode0"]
    assert img.dtype == ref.dtype and img.shape == ref.shape and abs(img.astype(int) - ref.astype(int)).max() <= 1
    p0 = cm.compute_joint_sensitivity_point_cloud(g["points"])
    assert torch.allclose(p0, g["point_sensitivity"], atol=1e-6)
    assert torch.allclose(cm.visualize_joint_sensitivity_point_cloud(p0, g["color_map"], 0), g["point_colors_mode0"], atol=1e-6)
    assert torch.allclose(cm.visualize_joint_sensitivity_point_cloud(p0, g["color_map"], 1), g["point_colors_mode1"], atol=1e-6)
    assert torch.allclose(torch.tensor(cm.JACOBIAN_COLORMAP["model_allegro"]).t(), g["color_map"])
    for name in ("model_toy_arm", "model_pneumatic_hand_only", "model_allegro_transformer"):
        assert torch.allclose(torch.tensor(cm.JACOBIAN_COLORMAP[name]), g["colormap_" + name].float())


def test_flow_and_depth_colour_maps_are_well_formed():
    """Restated third-party helpers (parity unpinned): structural properties of the published algorithms."""
    from neural_jacobian_field_amd.visualization import apply_depth_colormap, flow_to_image, turbo
    ang = torch.linspace(0, 2 * torch.pi, 9)[:-1]
    flow = torch.stack([torch.cos(ang), torch.sin(ang)], 0).reshape(2, 2, 4)
    flow[:, 0, 0] = 0.0                                        # zero motion -> white
    img = flow_to_image(flow)
    assert img.dtype == torch.uint8 and img.shape == (3, 2, 4)
    assert img[:, 0, 0].tolist() == [255, 255, 255]
    assert img[:, 0, 1].float().std() > 50                      # unit-magnitude motion -> saturated hue
    batch = flow_to_image(torch.stack([flow, 0.5 * flow]))
    assert torch.equal(batch[0], img)                           # normalised by the batch maximum
    assert (batch[1].float() >= batch[0].float() - 1).all()     # half the motion -> paler
    x = torch.linspace(0, 1, 64)
    rgb = turbo(x)
    assert rgb.shape == (64, 3) and rgb.min() >= 0 and rgb.max() <= 1
    assert rgb[6, 2] > rgb[6, 0] + 0.3 and rgb[-6, 0] > rgb[-6, 2] + 0.3 and rgb[32, 1] > 0.9   # blue -> green -> red
    d = torch.rand(2, 5, 6, 1)
    out = apply_depth_colormap(d)
    assert out.shape == (2, 5, 6, 3)
    assert torch.allclose(apply_depth_colormap(d, accumulation=torch.zeros_like(d)), torch.ones_like(out))


def test_header_is_plain_c(tmp_path, built):
    """include/njf_hip.h is the drop-in boundary: it must compile as C99 and as C++ with no other headers, and a C
    translation unit that takes the address of every declared entry point must link against the library."""
    import subprocess
    header = os.path.join(ROOT, "include", "njf_hip.h")
    for lang, std in (("c", "c99"), ("c++", "c++17")):
        subprocess.run(["gcc", "-x", lang, f"-std={std}", "-fsyntax-only", "-Wall", "-Werror", header], check=True)
    names = sorted(set(re.findall(r"\b(njf_[a-z0-9_]+)\s*\(", open(header).read())))
    src = tmp_path / "link_check.c"
    src.write_text('#include "njf_hip.h"\n#include <stdio.h>\nint main(void) {\n  const void* fns[] = {'
                   + ", ".join(f"(const void*)&{n}" for n in names)
                   + '};\n  printf("%d %d\\n", (int)(sizeof(fns) / sizeof(fns[0])), njf_abi_version());\n  return 0;\n}\n')
    lib_dir = os.path.join(ROOT, "neural-jacobian-field_amd")
    exe = tmp_path / "link_check"
    subprocess.run(["gcc", "-std=c99", f"-I{os.path.join(ROOT, 'include')}", str(src), "-o", str(exe), f"-L{lib_dir}",
                    "-l:libnjf_hip.so", f"-Wl,-rpath,{lib_dir}", "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == len(names) and int(out[1]) == 20


def test_static_isa_properties_of_the_fused_kernels():
    """Regression guards that need no GPU (tools/isa_report.py, ~1 min of hipcc): (1) the weight-stream DMA must stay the
    MUBUF form -- a FLAT-encoded global_load_lds makes hipcc force every s_waitcnt to zero while it is outstanding, which
    silently costs the split-precision kernels their A-fragment prefetch; the symptom checked is
    the s_waitcnt histogram: exact lgkmcnt(4)/(6) waits must dominate the LDS waits of the f16x2 kernels;
    (2) register spills of the inference kernels stay bounded (252 spilled VGPRs once cost 10 % of the frame)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_report.py")], check=True, capture_output=True,
                         text=True).stdout
    rows = {}
    for line in out.splitlines():
        if line.startswith("#"):
            continue
        name, regs, mix, lgkm, _ = [c.strip() for c in line.split("|")]
        rows[name] = dict(spill=int(regs.split()[1].split("/")[0]), lds_dma=int(mix.split("(")[1].split(")")[0]),
                          lgkm={int(k): int(v) for k, v in (kv.split(":") for kv in lgkm.split())} if lgkm != "-" else {})
    def row(prefix):
        hits = [k for k in rows if k.startswith(prefix)]
        assert len(hits) == 1, (prefix, hits)
        return rows[hits[0]]

    # f16x2 kernels (template arguments: Jacobian kind, precision, dump, action features[, Jacobian-head precision])
    for name in ("void render_kernel<1, 1, 0, false, 1>", "void proposal_kernel<1, false>"):
        r = row(name)
        assert r["lds_dma"] > 0, name                                   # the weight stream is an LDS DMA ...
        exact = r["lgkm"].get(4, 0) + r["lgkm"].get(6, 0)
        assert exact >= 60 and exact > r["lgkm"].get(0, 0), (name, r["lgkm"])   # ... and does not zero the wait counts
    assert row("void render_kernel<1, 1, 0, false, 1>")["spill"] <= 96
    assert row("void proposal_kernel<1, false>")["spill"] == 0
    # the fp6-corrected kernels: LDS-DMA weight stream, counted waits (not all zero), bounded spills
    for name, spill in (("void render_kernel<1, 2, 0, false, 2>", 110), ("void proposal_kernel<2, false>", 0)):
        r = row(name)
        assert r["lds_dma"] > 0 and r["spill"] <= spill, (name, r)
        assert sum(v for k, v in r["lgkm"].items() if k > 0) > r["lgkm"].get(0, 0), (name, r["lgkm"])
    # the plain-fp16 kernels (round 5): no spilled VGPR in any inference instantiation -- the shared tile state of the render kernel
    # (TileShareF16, 16-24 registers) is only instantiated where it fits -- and the weight stream is the LDS DMA here too
    for name in ("void render_kernel<1, 3, 0, false, 3>", "void render_kernel<1, 3, 0, true, 3>", "void render_kernel<2, 3, 0, false, 3>",
                 "void render_kernel<2, 3, 0, true, 3>", "void proposal_kernel<3, false>", "void points_kernel<2, 3, 3, false>"):
        r = row(name)
        assert r["spill"] == 0 and r["lds_dma"] > 0, (name, r)


def test_reference_checkpoint_keys_load_strictly():
    """A reference checkpoint holds `decoder.directional_encoding.tcnn_encoding.params` (tinycudann registers an empty
    Parameter for the spherical-harmonics encoding, action_decoder_jacobian.py:284); it must load with strict=True, both
    into Model and -- `model.`-prefixed -- into ModelWrapper, and must not appear in our own state dict."""
    from neural_jacobian_field_amd import synthetic
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import Model
    from neural_jacobian_field_amd.model_wrapper import ModelWrapper
    for kind, a in (("jacobian_mlp", 8), ("jacobian_transformer", 6), ("flow_mlp", 5)):
        cfg = model_cfg_from_dict({"action_dim": a, "encoder": {"name": "precomputed"},
                                   "rendering": {"num_proposal_samples": [8], "num_nerf_samples": 8}, "action_decoder": {"name": kind}})
        sd = synthetic.seeded_state_dict(synthetic.model_shapes(kind, a, with_encoder=False), seed=0)
        sd["decoder.directional_encoding.tcnn_encoding.params"] = torch.zeros(0)
        model = Model(cfg)
        model.load_state_dict(sd, strict=True)
        assert "decoder.directional_encoding.tcnn_encoding.params" not in model.state_dict()
        wrapper = ModelWrapper("perception", 16, Model(cfg))
        wsd = {"model." + k: v for k, v in sd.items()}
        wsd["depth_sigma"] = torch.tensor([0.001])
        wrapper.load_state_dict(wsd, strict=True)
        bad = dict(sd)
        bad["decoder.directional_encoding.tcnn_encoding.params"] = torch.zeros(3)   # a NON-empty stray tensor is still an error
        with pytest.raises(RuntimeError):
            Model(cfg).load_state_dict(bad, strict=True)


def test_training_step_drives_the_sampler_schedule():
    """ModelWrapper.training_step brackets the forward with step_before_iter / step_after_iter (the reference's
    on_train_batch_start / on_train_batch_end, model_wrapper.py:575-581): the proposal-weight anneal follows global_step."""
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.model import Model
    from neural_jacobian_field_amd.model_wrapper import ModelWrapper
    cfg = model_cfg_from_dict({"action_dim": 8, "encoder": {"name": "precomputed"},
                               "rendering": {"num_proposal_samples": [8], "num_nerf_samples": 8}})
    wrapper = ModelWrapper("perception", 16, Model(cfg))
    seen = []
    wrapper.evaluate_losses = lambda batch: (seen.append((wrapper.model.proposal_sampler._anneal, wrapper.model.proposal_sampler._step)),
                                             {"l": torch.zeros(())})[1]
    for _ in range(3):
        wrapper.training_step({})
        wrapper.optimizer_stepped()
        wrapper.global_step += 499     # jump ahead: 500 optimiser steps per call
    anneals = [a for a, _ in seen]
    assert anneals[0] == 0.0 and 0.0 < anneals[1] < anneals[2] == 1.0     # frac = step / 1000 -> 0, 0.5, 1.0
    assert wrapper.model.proposal_sampler._step == 1000 and wrapper.model.proposal_sampler._steps_since_update >= 1


def test_device_constant_caches_key_on_type_and_index():
    """the device-constant caches key on (type, index), build their constant once, and hand back the same tensor."""
    from neural_jacobian_field_amd import hip
    assert hip.device_key("cpu") == ("cpu", None)
    assert hip.device_key(torch.device("cuda", 3)) == ("cuda", 3) and hip.device_key("cuda:1") == ("cuda", 1)
    cache, built = {}, []

    def make():
        built.append(1)
        return torch.arange(4, dtype=torch.float32)

    key = ("u_base", 4) + hip.device_key("cpu")
    a = hip.cached_device_constant(cache, key, "cpu", make)
    b = hip.cached_device_constant(cache, key, "cpu", make)
    assert a is b and len(built) == 1 and list(cache) == [key]


def test_reserved_stdout_carries_the_json_line_and_nothing_else():
    """launch.reserve_stdout: whatever python OR native code writes to descriptor 1 afterwards lands on stderr; the one line of
    print_line goes to the process's original stdout (RCCL's banner used to precede the bench line, tests/test_rccl_gpu.py)."""
    code = ("import sys, os, ctypes; sys.path.insert(0, %r)\n"
            "from neural_jacobian_field_amd import launch\n"
            "print('before')\n"
            "launch.reserve_stdout(); launch.reserve_stdout()\n"
            "print('python noise'); os.write(1, b'descriptor noise\\n')\n"
            "libc = ctypes.CDLL(None); libc.puts(b'native noise'); libc.fflush(None)\n"
            "launch.print_line({'metric': 'x', 'value': 1})\n"
            "launch.release_stdout(); print('after')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-400:]
    assert r.stdout.splitlines() == ["before", json.dumps({"metric": "x", "value": 1}), "after"]
    for noise in ("python noise", "descriptor noise", "native noise"):
        assert noise in r.stderr


def test_bench_roofline_work_counts_follow_survey_8d():
    """bench.py's algorithmic MAC counts (roofline.achieved numerator): SURVEY 8d's table for the ResnetFC Jacobian head at A = 8,
    their dependence on A, and the folded / reference-formulation counts of the transformer head (bench.mac_jacobian)."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert bench.mac_jacobian("jacobian_mlp", 8) == {"canonical": 174_976, "reference_formulation": 371_584}
    assert bench.mac_jacobian("jacobian_mlp", 6)["canonical"] == 8_064 + 163_840 + 128 * 18
    t8, t6 = bench.mac_jacobian("jacobian_transformer", 8), bench.mac_jacobian("jacobian_transformer", 6)
    assert t8["reference_formulation"] == 284_096 == 575 * 64 + 3 * (64 * 512 + 2 * 512 * 8 + 512 * 64 + 2 * 64 * 64) + 64 * 24
    assert t8["canonical"] == 63 * 64 + 3 * (2 * 64 * 64 + 2 * 64 * 64) + 64 * 24 == 54_720
    assert t6["canonical"] == 63 * 64 + 3 * (2 * 64 * 48 + 2 * 64 * 64) + 64 * 18
    assert t6["reference_formulation"] == 575 * 64 + 3 * (64 * 512 + 2 * 512 * 6 + 512 * 64 + 2 * 64 * 64) + 64 * 18


def test_unifdef_lite_resolves_only_the_named_macros():
    """tools/unifdef_lite.py (what moved the retired experiment macros into tools/probes/*.patch): conditionals on the named macros are
    resolved as undefined -- #ifdef / #ifndef / #if defined() with &&, ||, !, #elif, #else, nested -- everything else stays verbatim."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import unifdef_lite
    src = """a
#ifdef X
x1
#else
x0
#endif
#ifndef X
nx
#endif
#if defined(X) && X == 2
x2
#elif defined(Y)
y
#else
none
#endif
#ifdef KEEP
k
#ifdef X
kx
#endif
#endif
#if defined(X) || defined(KEEP)
mixed
#endif
z
""".splitlines(keepends=True)
    out = "".join(unifdef_lite.run(src, {"X", "Y"}))
    assert out == "a\nx0\nnx\nnone\n#ifdef KEEP\nk\n#endif\n#if defined(X) || defined(KEEP)\nmixed\n#endif\nz\n"
