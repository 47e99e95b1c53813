#!/usr/bin/env python3
"""Run bench.py in-process, then print the experiment counters of an experiment library (NJF_HIP_LIB) to stderr.
Round-6 stagger experiment (-DNJF_STAGGER): how many first-generation workgroups took the delayed / undelayed path."""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    from neural_jacobian_field_amd import hip
    lib = hip.load_library()
    if hasattr(lib, "njf_debug_read_stagger"):
        buf = (ctypes.c_uint * 4)()
        rc = lib.njf_debug_read_stagger(buf)
        print(f"[stagger] rc={rc} undelayed={buf[0]} delayed={buf[1]}", file=sys.stderr)
