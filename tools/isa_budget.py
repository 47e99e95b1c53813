#!/usr/bin/env python3
"""Static per-phase instruction budget of one fused kernel per 32-point tile (no GPU needed; VERDICT r02 "next" #3).

    python tools/isa_budget.py [--kernel render|proposal] [--prec 2] [-DNJF_ABLATE_X ...]

Compiles csrc/njf_kernels.hip for ONE MFMA precision (-DNJF_DEV_ONLY_PREC), takes the inference instantiation of the
kernel, splits the body of its tile loop into basic blocks and weights every block by how often a tile executes it: the
`blk` loops of resnet_tile are real loops in the ISA (the gather block runs 3 times, the 4-chunk block body 5 times per
network).  Prints VALU / MFMA / LDS / VMEM / SALU / s_waitcnt / barrier counts per tile; with ablation defines the
difference to the baseline attributes instructions to the removed phase.  `--classify` additionally buckets the VALU
instructions by mnemonic family (conversions, fma_mix, max, dpp/permute, cndmask, fp32 arithmetic ...)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SRC = os.path.join(entry.CSRC, "njf_kernels.hip")


def compile_asm(prec, extra):
    flags = [f for f in entry.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + [f"-DNJF_DEV_ONLY_PREC={prec}"] + extra
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        res = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *flags, "-S", "--cuda-device-only", SRC, "-o", asm],
                             capture_output=True, text=True)
        if res.returncode:
            sys.exit(res.stderr[-3000:])
        return open(asm).read()


def kernel_body(text, mangled_re):
    m = re.search(r"^(" + mangled_re + r"):.*?\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    if not m:
        sys.exit("kernel not found: " + mangled_re)
    return m.group(2)


def family(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_cvt_scalef32"):
        return "cvt_fp6"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_fma_mix"):
        return "fma_mix"
    if "dpp" in op or op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "v_mov_b32_dpp")):
        return "lane"
    if op.startswith(("v_max", "v_min", "v_pk_max", "v_pk_min", "v_med3")):
        return "minmax"
    if op.startswith(("v_cndmask", "v_cmp", "v_cmpx")):
        return "select/cmp"
    if op.startswith(("v_mov", "v_accvgpr", "v_pk_mov")):
        return "mov"
    if op.startswith(("v_fma", "v_fmac", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_pk_fma", "v_pk_mul", "v_pk_add", "v_mac", "v_mad_f32",
                      "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_rndne", "v_floor", "v_fract", "v_trunc", "v_div", "v_ldexp",
                      "v_frexp", "v_sin", "v_cos")):
        return "fp32 arith"
    return "int/bit"


def analyse(body, classify):
    """Weights: blocks of the tile loop (Depth=1 header) x1; inner loops (Depth=2): the block that contains the gather
    (global_load_dwordx4 without 'lds') x3, the other x5.  Everything outside the tile loop is ignored (prologue/epilogue)."""
    lines = body.splitlines()
    # basic blocks: label -> (lines, annotation)
    blocks, cur, ann = [], [], ""
    for ln in lines:
        if re.match(r"^\.LBB\d+_\d+:", ln):
            blocks.append((ann, cur))
            cur, ann = [], ln
        else:
            cur.append(ln)
    blocks.append((ann, cur))
    # the tile loop = everything from the first "Loop Header: Depth=1" label to the last branch back to it
    hdr = None
    for i, (a, _) in enumerate(blocks):
        m = re.match(r"^\.(LBB\d+_\d+):.*=>This Loop Header: Depth=1", a)
        if m:
            hdr, first = m.group(1), i
            break
    if hdr is None:
        sys.exit("no depth-1 loop header found")
    last = first
    while last + 1 < len(blocks) and "Loop" in blocks[last + 1][0]:   # the blocks of the loop nest follow the header
        last += 1
    if first > 0 and hdr[1:] in blocks[first - 1][0]:                  # the latch may sit in front of the header
        first -= 1
    total = collections.Counter()
    fam = collections.Counter()
    per_block = []
    for i, (a, ls) in enumerate(blocks):
        if i < first or i > last:
            continue
        name = re.match(r"^\.(LBB\d+_\d+)", a).group(1) if a else ""
        ops = [l.split()[0] for l in ls if l.startswith("\t") and not l.startswith(("\t.", "\t;"))]
        # the `blk` loop of resnet_tile is a real loop: the block holding a gather (>= 32 plain global_load_dwordx4) runs 3
        # times per network, the block holding the four 128-wide chunks of a ResNet block (4 barriers) 5 times
        gathers = sum(1 for l in ls if re.match(r"\tglobal_load_dwordx4", l) and " lds" not in l)
        weight = 3 if gathers >= 32 else (5 if ops.count("s_barrier") == 4 else 1)
        c = collections.Counter()
        for op in ops:
            if op.startswith("v_mfma"):
                c["MFMA"] += 1
            elif op.startswith("v_"):
                c["VALU"] += 1
            elif op.startswith("ds_"):
                c["LDS"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["VMEM"] += 1
            elif op == "s_waitcnt":
                c["waitcnt"] += 1
            elif op == "s_barrier":
                c["barrier"] += 1
            elif op == "s_nop":
                c["nop"] += 1
            elif op.startswith("s_"):
                c["SALU"] += 1
            if classify and op.startswith("v_"):
                fam[family(op)] += weight
        for k, v in c.items():
            total[k] += weight * v
        per_block.append((name, weight, dict(c)))
    return total, fam, per_block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="render", choices=["render", "proposal"])
    ap.add_argument("--prec", type=int, default=2)
    ap.add_argument("--classify", action="store_true")
    ap.add_argument("--blocks", action="store_true")
    a, extra = ap.parse_known_args()
    text = compile_asm(a.prec, extra)
    if a.kernel == "render":
        body = kernel_body(text, rf"_Z13render_kernelILi1ELi{a.prec}ELi0ELb0ELi{a.prec}EEv10RenderArgs")
    else:
        body = kernel_body(text, rf"_Z15proposal_kernelILi{a.prec}ELb0EEv12ProposalArgs")
    total, fam, per_block = analyse(body, a.classify)
    print(f"# {a.kernel} kernel, precision {a.prec}, flags {' '.join(extra) or '-'}: instructions per 32-point tile (static, loop-weighted)")
    print(" ".join(f"{k}={total[k]}" for k in ("VALU", "MFMA", "LDS", "VMEM", "SALU", "waitcnt", "barrier", "nop")))
    if a.classify:
        print("VALU by family: " + ", ".join(f"{k} {v}" for k, v in fam.most_common() if k != "mfma"))
    if a.blocks:
        for name, w, c in per_block:
            print(f"  {name:12s} x{w}  {c}")


if __name__ == "__main__":
    main()
