#!/usr/bin/env python3
"""Static report on the compiled kernels (no GPU needed): registers / spills / scratch, instruction mix and the
histogram of s_waitcnt operands per kernel.

    python tools/isa_report.py [extra hipcc flags ...]  > profiles/rNN_isa_report.txt

The wait-count histogram is the cheap detector of the problem that cost the render kernel 3 % in round 1: a FLAT-encoded
instruction that may touch LDS (global_load_lds_*) makes hipcc's wait-count pass force EVERY s_waitcnt to lgkmcnt(0) /
vmcnt(0) while it is outstanding, so a software pipeline of ds_reads waits for the loads it has just issued.  A kernel
whose LDS waits are almost all lgkmcnt(0) although it prefetches is suspect.  Uses the build flags of __graft_entry__."""
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
FLAGS = [f for f in entry.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + sys.argv[1:]
KERNELS = ["render_kernel", "proposal_kernel", "points_kernel", "project_kernel", "solve_action_kernel",
           "scatter_footprint_kernel", "relu_backward_kernel", "upsample_concat_kernel", "resnetfc_backward_kernel"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return dict(zip(names, out.splitlines()))


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        res = subprocess.run([hipcc, *FLAGS, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", SRC, "-o", asm],
                             capture_output=True, text=True)
        if res.returncode:
            sys.exit(res.stderr[-2000:])
        text = open(asm).read()
    usage, cur = collections.defaultdict(dict), None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        for key in ("VGPRs:", "AGPRs:", "ScratchSize", "SGPRs Spill", "VGPRs Spill", "Occupancy"):
            m = re.search(re.escape(key) + r"[^0-9]*(\d+)", line)
            if m and cur:
                usage[cur][key.rstrip(":")] = int(m.group(1))
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    names = demangle(list(bodies))
    print("# static ISA report:", " ".join(FLAGS))
    print("# kernel | VGPRs spill(V/S) scratch B | MFMA VALU LDS VMEM(lds-dma) SALU | s_waitcnt lgkmcnt histogram | vmcnt(0) share")
    for mangled, body in sorted(bodies.items(), key=lambda kv: names[kv[0]]):
        nice = names[mangled]
        if not any(k in nice for k in KERNELS):
            continue
        ops = [ln.split()[0] for ln in body.splitlines() if ln.startswith("\t") and not ln.startswith("\t.") and not ln.startswith("\t;")]
        c = collections.Counter(ops)
        mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        dma = sum(body.count(f"{k} ") for k in ()) + len(re.findall(r"^\t(?:buffer|global)_load\w* .*\blds\b|^\tglobal_load_lds", body, re.M))
        vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
        salu = sum(v for k, v in c.items() if k.startswith("s_") and k not in ("s_waitcnt", "s_nop", "s_barrier"))
        lg = collections.Counter(int(x) for x in re.findall(r"lgkmcnt\((\d+)\)", body))
        vm = collections.Counter(int(x) for x in re.findall(r"vmcnt\((\d+)\)", body))
        u = usage.get(mangled, {})
        lg_txt = " ".join(f"{k}:{v}" for k, v in sorted(lg.items())) or "-"
        vm0 = f"{vm.get(0, 0)}/{sum(vm.values())}" if vm else "-"
        flag = "  <-- LDS waits all zero: check for FLAT LDS-DMA" if lds > 16 and lg and lg.get(0, 0) > 0.9 * sum(lg.values()) else ""
        print(f"{nice[:78]:78s} | {u.get('VGPRs', '?'):>3} {u.get('VGPRs Spill', 0):>3}/{u.get('SGPRs Spill', 0):<3} {u.get('ScratchSize', 0):>4} | "
              f"{mfma:>4} {valu:>5} {lds:>4} {vmem:>4}({dma}) {salu:>5} | {lg_txt} | {vm0}{flag}")


if __name__ == "__main__":
    main()
