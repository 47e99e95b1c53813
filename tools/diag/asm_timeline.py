#!/usr/bin/env python3
"""Compressed instruction timeline of one kernel of a `hipcc -save-temps` assembly file (development aid):
G global_load  D buffer_load (LDS DMA)  M mfma  f v_fma_mix  r ds_read  S/L scratch store/load  | s_barrier  w s_waitcnt
    python tools/diag/asm_timeline.py <file.s> <mangled kernel name>"""
import collections
import sys

text = open(sys.argv[1]).read()
i = text.index(sys.argv[2] + ":")
body = text[i:text.index(".end_amdhsa_kernel", i)].splitlines()
CODE = [("global_load", "G"), ("v_mfma", "M"), ("v_fma_mix", "f"), ("scratch_store", "S"), ("scratch_load", "L"), ("buffer_load", "D"),
        ("ds_read", "r"), ("s_barrier", "|"), ("s_waitcnt", "w")]
ops, seq = collections.Counter(), []
for line in body:
    s = line.strip()
    if not s or s[0] in ".;" or s.endswith(":"):
        continue
    op = s.split()[0]
    ops[op] += 1
    seq.append(next((c for p, c in CODE if op.startswith(p)), ""))
print({k: v for k, v in ops.items() if k.startswith(("scratch", "global_load", "v_mfma", "v_fma_mix", "buffer_load", "s_barrier"))})
s = "".join(seq)
for k in range(0, len(s), 200):
    print(s[k:k + 200])
