#!/usr/bin/env python3
"""Launches of ONE steady-state step out of a rocprofv3 kernel trace: everything between the last two launches of a marker kernel
that runs once per step (default: the backward chain).  Prints the launch count, the device-busy time, the span, and the kernels
by total time -- the evidence behind the host-launch figures of DESIGN section 7 (profiles/r06_train_launches_*.txt).

    rocprofv3 --kernel-trace -d gpurun_out/prof -- python tools/bench_train.py ...
    python tools/launches_per_step.py gpurun_out/prof/**/*_kernel_trace.csv [marker-prefix]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "transformer_backward_kernel"
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(marker) or marker in r["Kernel_Name"][:80]]
    if len(idx) < 3:
        raise SystemExit(f"marker {marker!r} seen {len(idx)} times")
    seg = rows[idx[-3]:idx[-2]]
    busy, count = collections.Counter(), collections.Counter()
    for r in seg:
        n = r["Kernel_Name"][:72]
        count[n] += 1
        busy[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    print(f"{path}\nlaunches per step: {len(seg)}   device busy: {sum(busy.values()) / 1e3:.3f} ms   span under the profiler: {span:.3f} ms")
    for n, t in sorted(busy.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {t:9.1f} us  x{count[n]:4d}  {n}")


if __name__ == "__main__":
    main()
