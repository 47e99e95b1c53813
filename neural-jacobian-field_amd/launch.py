"""Self-launch of the one-process-per-GPU benches, and the evidence a multi-GPU line has to carry.

The reference scales with Lightning DDP, which spawns one process per device by itself (``train.py:67-79``:
``Trainer(devices="auto", strategy="ddp_find_unused_parameters_true")``); a user of it never types ``torchrun``.  The benches
here keep that property: ``bench.py --gpus N`` / ``tools/bench_train.py --gpus N`` started as a PLAIN python process spawn
their N ranks themselves (``torch.distributed.run``, rendezvous on 127.0.0.1), and started under a launcher they check that
the world they find is the world they were asked for.  A line whose ``n_gpus`` differs from ``--gpus``, or whose ranks do
not sit on N distinct devices, is refused, not printed (VERDICT r03 "next" #1).

Plain torch + the standard library: usable with the ``gloo`` backend on CPU (tests/test_host_cpu.py drives the whole path
with world size 2).
"""

from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Dict, List, Optional, Sequence

import torch

LAUNCH_MARK = "NJF_SELF_LAUNCHED"


def env_world() -> Optional[int]:
    """World size a launcher (torchrun, or this module) put into the environment; None for a plain process."""
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:
        return int(os.environ["WORLD_SIZE"])
    return None


def ensure_world(gpus: int, script: str, argv: Sequence[str], need_devices: bool = True) -> None:
    """Make the process that continues past this call one rank of a world of exactly ``gpus``.

    * launcher environment present: the world must equal ``gpus`` (SystemExit otherwise -- also for ``--gpus 8`` under a
      one-rank launcher, the case that used to print a one-GPU line);
    * plain process, ``gpus == 1``: continue as the only rank;
    * plain process, ``gpus > 1``: spawn ``gpus`` ranks of the same command line through ``python -m torch.distributed.run``
      (one per device, rendezvous on 127.0.0.1 with a free port), forward their exit status and never return."""
    world = env_world()
    if world is not None:
        if world != gpus:
            raise SystemExit(f"refusing to run: --gpus {gpus} but the launcher's WORLD_SIZE is {world}")
        return
    if gpus == 1:
        return
    if need_devices:
        have = torch.cuda.device_count()
        if have < gpus:
            raise SystemExit(f"refusing to run: --gpus {gpus} but only {have} GPU(s) are visible to this process")
    env = dict(os.environ)
    env[LAUNCH_MARK] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    # --standalone: torchrun's own c10d rendezvous on a port IT binds (no bind-close-reuse race of a port picked here)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={gpus}", script, *argv]
    print("[launch]", " ".join(cmd), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def init_process_group(backend: str, device: Optional[torch.device] = None, reserve: bool = False):
    """torch.distributed for this rank (``nccl`` IS RCCL on ROCm).  A plain single process (``--force-dist``) becomes a world
    of one on 127.0.0.1.  Returns the ``torch.distributed`` module.

    ``reserve``: RCCL prints its version banner on stdout when the first communicator is created (lazily, at the first
    collective).  A caller whose stdout is a protocol (the benches' ONE JSON line) calls ``reserve_stdout()`` itself before this
    function, or passes ``reserve=True``; by default this library helper leaves the process's descriptors alone."""
    import torch.distributed as dist

    kw = {}
    if "MASTER_ADDR" not in os.environ or "RANK" not in os.environ:
        # a plain process: a world of one whose store binds an ephemeral port itself (port 0: nothing to race for)
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        kw = dict(store=dist.TCPStore("127.0.0.1", 0, 1, is_master=True), rank=0, world_size=1)
    if backend == "nccl":
        if reserve:
            reserve_stdout()
        dist.init_process_group("nccl", device_id=device, **kw)
    else:
        dist.init_process_group(backend, **kw)
    return dist


def _device_record(device: torch.device) -> Dict:
    rec = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
           "host": socket.gethostname(), "pid": os.getpid(), "device": str(device)}
    if device.type == "cuda":
        p = torch.cuda.get_device_properties(device)
        bus = None
        if hasattr(p, "pci_bus_id"):
            bus = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}"
        rec.update(name=p.name, pci_bus_id=bus, uuid=str(getattr(p, "uuid", "")) or None,
                   gcn_arch=getattr(p, "gcnArchName", None), compute_units=p.multi_processor_count,
                   hbm_gib=round(p.total_memory / 2 ** 30, 1))
    return rec


def rank_evidence(dist, device: torch.device, local_step_ms: float, graph: bool = False, kernel_ms: Optional[Dict] = None) -> Dict:
    """What every rank contributes to the line rank 0 prints -- gathered THROUGH the process group, so the list can only be
    as long as the world that really exchanged data: device identity per rank (PCI bus id, uuid), its own step time
    (clock stopped after its own device synchronisation, before the closing barrier), backend and library version.

    Raises SystemExit when two ranks report the same physical device (a mis-set LOCAL_RANK / HIP_VISIBLE_DEVICES would
    otherwise produce a well-formed "N-GPU" line measured on fewer GPUs)."""
    rec = _device_record(device)
    rec["step_ms"] = round(float(local_step_ms), 4)
    if kernel_ms is not None:   # this rank's own per-launch kernel durations (HIP events): compute vs collective, from the line alone
        rec["kernel_ms"] = {k: round(float(v), 4) for k, v in kernel_ms.items()}
    if dist is None:
        recs, backend, world = [rec], None, 1
    else:
        world = dist.get_world_size()
        recs: List = [None] * world
        dist.all_gather_object(recs, rec)
        backend = dist.get_backend()
    recs = sorted(recs, key=lambda r: r["rank"])
    if [r["rank"] for r in recs] != list(range(world)):
        raise SystemExit(f"refusing to report: ranks present {[r['rank'] for r in recs]} != 0..{world - 1}")
    if device.type == "cuda":
        # two ranks sit on ONE device only if every hardware identifier they report coincides (bus id AND uuid): a partitioned
        # GPU may share one of them between partitions, and a false refusal would cost the whole multi-GPU measurement
        ids = [(r["host"], r.get("pci_bus_id"), r.get("uuid")) if (r.get("pci_bus_id") or r.get("uuid")) else (r["host"], r["device"])
               for r in recs]
        if len(set(ids)) != world:
            raise SystemExit(f"refusing to report: {world} ranks on {len(set(ids))} distinct device(s): {ids}")
    steps = [r["step_ms"] for r in recs]
    out = {"backend": backend, "world_size": world, "graph": bool(graph), "devices": recs,
           "self_launched": os.environ.get(LAUNCH_MARK) == "1",
           "rank_step_ms": {"min": min(steps), "max": max(steps), "per_rank": steps}}
    if all("kernel_ms" in r for r in recs):
        # per rank: the sum of its kernels' durations next to its step time; step - kernels = launch gaps + the ONE collective + assemble
        ksum = [round(sum(r["kernel_ms"].values()), 4) for r in recs]
        out["rank_kernel_ms"] = {"per_rank": [r["kernel_ms"] for r in recs], "sum_per_rank": ksum,
                                 "step_minus_kernels_ms": [round(s - k, 4) for s, k in zip(steps, ksum)]}
    if backend == "nccl":
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:   # the version query is decoration, never a reason to lose the line
            out["rccl_version"] = None
        out["transport_env"] = {k: os.environ[k] for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "HIP_VISIBLE_DEVICES",
                                                          "ROCR_VISIBLE_DEVICES") if k in os.environ}
    return out


_line_stream = None


def reserve_stdout():
    """Make the process's stdout carry the ONE JSON line and nothing else: the original descriptor is kept for ``print_line``,
    descriptor 1 itself is pointed at stderr, so whatever native libraries print (RCCL's version banner at communicator
    creation -- it appears at NCCL_DEBUG=WARN too, found by tests/test_rccl_gpu.py in round 5 --, MIOpen warnings) cannot
    precede or follow the line.  Idempotent; call before the process group is initialised."""
    global _line_stream
    if _line_stream is None:
        sys.stdout.flush()
        _line_stream = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _line_stream


def release_stdout() -> None:
    """Undo reserve_stdout (tools that call a bench's main() and then print their own report: tools/stamps.py)."""
    global _line_stream
    if _line_stream is not None:
        sys.stdout.flush()
        os.dup2(_line_stream.fileno(), 1)
        _line_stream.close()
        _line_stream = None


def print_line(line: Dict) -> None:
    """The JSON line, on the process's ORIGINAL stdout (see reserve_stdout)."""
    import json
    out = reserve_stdout()
    out.write(json.dumps(line) + "\n")
    out.flush()


def check_line(line: Dict, gpus: int) -> Dict:
    """Last gate in front of ``print``: the line's ``n_gpus`` is the world that ran AND the ``--gpus`` that was asked for."""
    n = line.get("n_gpus")
    ev = line.get("rccl") or {}
    if n != gpus or (ev and ev.get("world_size") != gpus) or (ev and len(ev.get("devices", ())) != gpus):
        raise SystemExit(f"refusing to print a line with n_gpus={n}, evidence of {ev.get('world_size')} rank(s), for --gpus {gpus}")
    return line
