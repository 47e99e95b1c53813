"""Pick the host thread count for bench.py's cpu_baseline leg (run once on the GPU box).  Times the CPU ORACLE itself, so it
needs oracle/ next to the package (the only kind of tool that does: it measures / verifies with the checker)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import parity_harness as ph
case = ph.make_case(1, 256, 256, None, 8, seed=0)
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    sub = dict(case); sub["origins"] = case["origins"][:, :512].contiguous(); sub["directions"] = case["directions"][:, :512].contiguous()
    ph.oracle_forward(sub, 64, 64)
    t0 = time.perf_counter(); ph.oracle_forward(sub, 64, 64); dt = time.perf_counter() - t0
    print(n, "threads:", round(512 / dt, 1), "rays/s", flush=True)
