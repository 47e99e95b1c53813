#!/usr/bin/env python3
"""Basic blocks of one kernel of a `hipcc -save-temps` assembly file with their instruction mix (development aid): which
blocks are loop bodies (a branch back to them), how many MFMA / VALU / SGPR-spill lane moves / memory instructions they hold.
    python tools/diag/asm_blocks.py <file.s> <mangled kernel name> [min instructions to print]"""
import re
import sys

text = open(sys.argv[1]).read()
i = text.index(sys.argv[2] + ":")
body = text[i:text.index(".end_amdhsa_kernel", i)].splitlines()
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 20
blocks, cur = [], dict(label="entry", n=0, mfma=0, valu=0, lane=0, vmem=0, lds=0, wait=0, br=[])
for line in body:
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        blocks.append(cur)
        cur = dict(label=m.group(1), n=0, mfma=0, valu=0, lane=0, vmem=0, lds=0, wait=0, br=[])
        continue
    s = line.strip()
    if not s or s[0] in ".;" or s.endswith(":"):
        continue
    op = s.split()[0]
    cur["n"] += 1
    if op.startswith("v_mfma"):
        cur["mfma"] += 1
    elif op in ("v_readlane_b32", "v_writelane_b32"):
        cur["lane"] += 1
    elif op.startswith("v_"):
        cur["valu"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_")):
        cur["vmem"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith("s_waitcnt"):
        cur["wait"] += 1
    if op.startswith(("s_cbranch", "s_branch")):
        cur["br"].append(s.split()[-1])
blocks.append(cur)
order = {b["label"]: k for k, b in enumerate(blocks)}
for k, b in enumerate(blocks):
    back = [t for t in b["br"] if order.get(t, 1 << 30) <= k]
    if b["n"] >= floor or back:
        print(f"{b['label']:12s} n={b['n']:5d} mfma={b['mfma']:4d} valu={b['valu']:5d} lane={b['lane']:4d} vmem={b['vmem']:4d} lds={b['lds']:4d} "
              f"wait={b['wait']:4d}" + (f"  loops back to {back}" if back else ""))
