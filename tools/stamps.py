#!/usr/bin/env python3
"""Timeline of ONE wave of the render kernel from in-kernel time stamps (experiment build, not the product):

    hipcc ... -DNJF_STAMPS -shared -fPIC neural-jacobian-field_amd/csrc/njf_kernels.hip -o build/libnjf_stamps.so
    NJF_HIP_LIB=$PWD/build/libnjf_stamps.so python tools/stamps.py            (on the GPU box)

Runs the default bench workload once and prints, for wave 1 of a mid-grid workgroup, where its cycles went: per chunk
the time from `work issued` to `memory drained` (vmcnt(0)), to `barrier passed`, the gathers, and the stage totals."""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-precisions"] + sys.argv[1:]
import bench  # noqa: E402

bench.main()
from neural_jacobian_field_amd import hip, launch  # noqa: E402

launch.release_stdout()   # bench.main() reserved the process's stdout for its JSON line; the report below goes there again

lib = hip.load_library()
buf = (ctypes.c_uint * 1024)()
lib.njf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.njf_debug_read_stamps(buf, 1024)
assert rc == 0, rc
ev = [(w >> 24, w & 0xffffff) for w in buf if w]
print(f"# {len(ev)} stamps")
# unwrap the 24-bit counter
t, prev, base = [], None, 0
for tag, x in ev:
    if prev is not None and x < prev:
        base += 1 << 24
    prev = x
    t.append(base + x)
names = {6: "jac<", 7: "pe>", 14: "sigma", 15: "weights|pdf>", 1: "issued", 2: "drained", 3: "barrier", 4: "gather<", 5: "gather>", 9: "begin", 10: "tile", 11: "density.", 12: "colour.", 13: "end"}
span = collections.defaultdict(int)
count = collections.defaultdict(int)
for i in range(1, len(ev)):
    key = f"{names.get(ev[i - 1][0], ev[i - 1][0])}->{names.get(ev[i][0], ev[i][0])}"
    span[key] += t[i] - t[i - 1]
    count[key] += 1
total = t[-1] - t[0]
print(f"# wave lifetime {total} ticks of s_memtime")
for key, v in sorted(span.items(), key=lambda kv: -kv[1]):
    print(f"{key:24s} {v:9d} ticks  {100.0 * v / total:5.1f} %   x{count[key]:4d}  avg {v / count[key]:8.1f}")
if os.environ.get("NJF_STAMPS_RAW"):
    for (tag, _), tt in zip(ev, t):
        print(names.get(tag, tag), tt - t[0])
