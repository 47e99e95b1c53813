"""RCCL inside the driver-run suite (VERDICT r04 "next" #3): the benches as subprocesses on ONE GPU with the process group
initialised on the `nccl` backend (= RCCL on ROCm) -- the collective of the multi-GPU contract (train.py:67-79: one rank per GPU
under DDP; here ray shards + ONE all_gather per step, parallel.ShardedFrameStep) runs for real, with a single rank.  Asserted:
ONE JSON line on stdout, rccl.backend == "nccl", world_size == 1, a device record with PCI bus id and uuid, and a frame / loss
digest EQUAL to the one the non-distributed step prints (the kernels are bit-reproducible).  Run with -m gpu."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_line(script, *args, timeout=900):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):   # a plain process: the script brings up its own world
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, script), *args], capture_output=True, text=True, timeout=timeout,
                         cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines      # ONE JSON line on stdout
    return json.loads(lines[0])


def check_rccl(line):
    r = line["rccl"]
    assert r["backend"] == "nccl" and r["world_size"] == 1 and line["n_gpus"] == 1, r
    dev = r["devices"][0]
    assert dev["rank"] == 0 and dev.get("pci_bus_id") and dev.get("uuid") and dev["device"] == "cuda:0", dev


BENCH = ("--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-precisions")


@pytest.fixture(scope="module")
def plain_bench():
    line = run_line("bench.py", *BENCH)
    assert line["rccl"]["backend"] is None and line["frame_digest"]["sha256"]
    return line


@pytest.mark.parametrize("graph", [False, True])
def test_bench_one_rank_with_the_collective_equals_the_plain_step(plain_bench, graph):
    line = run_line("bench.py", *BENCH, "--force-dist", *(["--graph"] if graph else []))
    check_rccl(line)
    assert line["step"]["hip_graph"] is graph
    assert line["metric"] == plain_bench["metric"] and line["config"]["workload"] == plain_bench["config"]["workload"]
    # all_gather_into_tensor of [pixels | 4 scalars] with one rank + the assemble launch: bit for bit the plain step's frame
    assert line["frame_digest"]["sha256"] == plain_bench["frame_digest"]["sha256"]


def test_bench_train_one_rank_with_the_gradient_all_reduce():
    """tools/bench_train.py (config 4's step: forward + backward + ONE flattened gradient all-reduce + Adam) with the bucket
    all-reduce on RCCL, one rank.  ONE seeded step without warm-up: its loss is the first forward's, which equals the plain
    process's to the encoder's run-to-run noise (MIOpen's convolutions are not bit-reproducible across processes; after optimiser
    steps the seeded, far-from-converged head makes the flow loss chaotic -- two steps later the two processes differ by 12 %)."""
    args = ("--gpus", "1", "--steps", "1", "--warmup", "0", "--mode", "action")
    plain = run_line("tools/bench_train.py", *args)
    line = run_line("tools/bench_train.py", *args, "--force-dist")
    check_rccl(line)
    assert plain["rccl"]["backend"] is None
    assert line["gradient_bucket_bytes"] == plain["gradient_bucket_bytes"] > 0
    a, b = line["final_loss"], plain["final_loss"]
    # (six plain processes on one box: 3.103e6 ... 3.124e6, a spread of 0.65 % -- and 1.0 % was seen once between the two processes of
    # this test: the bound is a sanity check of "the same step", the collective is what check_rccl and the bucket size certify)
    assert a == a and abs(a - b) <= 5e-2 * abs(b), (a, b)
