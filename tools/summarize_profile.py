#!/usr/bin/env python3
"""Condense a tools/profile_rNN.sh output directory (gpurun_out/prof_rNN) into tracked files under profiles/.

  profiles/rNN_kernel_stats.csv            rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/rNN_pmc_summary.md              per-kernel PMC means + derived figures
  profiles/rNN_render_kernel_hbm_bytes.json   traffic figure bench.py reports in `roofline.traffic`
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
src = f"gpurun_out/prof_{tag}_{prec}"
round_tag = tag
tag = f"{tag}_{prec}"
bench_args = " ".join(sys.argv[3:])   # what the profiled bench ran with besides the defaults (e.g. --height 512 --width 512)
# per MFMA instruction: v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_f16.  The f16f6 kernels issue two kinds (f16 K=16:
# 32 cycles, 32,768 FLOP; fp6 K=64: 32 cycles, 131,072 FLOP) in the ratio 2:1 -> 65,536 FLOP per instruction on average
MFMA_FLOP = {"f32": 4096, "f16x2": 32768, "f16f6": 65536, "f16": 32768}[prec]
MFMA_CYCLES = 64 if prec == "f32" else 32
os.makedirs("profiles", exist_ok=True)
# the traced command may fork helpers, each leaving its own stats file: take the one that holds the fused kernels
stats = max(glob.glob(f"{src}/trace/*/*_kernel_stats.csv"), key=lambda f: open(f).read().count("render_kernel"))
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")

PCODE = {"f32": 0, "f16x2": 1, "f16f6": 2, "f16": 3}[prec]    # template argument of this precision's instantiations
PROP = 1 if prec == "f16f6" else PCODE              # under f16f6 the proposal networks stay on f16x2 (Model.set_precision)
JKIND = 2 if "jacobian_transformer" in bench_args else 1   # round 6: `--decoder jacobian_transformer` profiles (folded attention head)
HEAD = "jacobian_transformer" if JKIND == 2 else "jacobian_mlp"
KERNELS = {f"render_kernel<{JKIND}, {PCODE}": "render", f"proposal_kernel<{PROP},": "proposal",
           ("project_kernel_f32_lds(" if prec == "f32" else "project_kernel_f16x2<"): "project"}   # (rounds 1-4: "project_kernel(")
agg = collections.defaultdict(list)
meta = {}
for f in glob.glob(f"{src}/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for pat, short in KERNELS.items():
            if pat in r["Kernel_Name"]:
                agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
                meta[short] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
mean = {k: sum(v) / len(v) for k, v in agg.items()}
dur = {}
for r in csv.DictReader(open(stats)):
    for pat, short in KERNELS.items():
        if pat in r["Name"]:
            dur[short] = float(r["AverageNs"]) * 1e-9

lines = [f"# {tag}: rocprofv3 PMC summary (MI355X; durations from `rocprofv3 --kernel-trace --stats -- python bench.py --precision {prec} {bench_args} --steps 20 --warmup 3`, counters from `--pmc` runs of the same command with `--steps 2 --warmup 1`, per-launch means)", "",
         f"Collected by `tools/profile_{round_tag.split('_')[0]}.sh {prec} {round_tag}`: kernel-trace/stats and each PMC group in separate runs.", "",
         "| kernel | avg duration (kernel-trace) | launch config |", "|---|---|---|"]
for k in ("project", "proposal", "render"):
    lines.append(f"| {k} | {dur[k]*1e3:.3f} ms | {meta.get(k)} |")
lines += ["", "| kernel | counter | mean per launch |", "|---|---|---|"]
for (k, c), v in sorted(mean.items()):
    lines.append(f"| {k} | {c} | {v:.4g} |")
lines += ["", "## Derived", ""]
out_json = {}
for k in ("proposal", "render"):
    xcd_cycles = mean[(k, "GRBM_GUI_ACTIVE")] / 8.0          # one GRBM per XCD, summed by rocprofv3
    clock = xcd_cycles / dur[k]
    mfma_util = mean[(k, "SQ_VALU_MFMA_BUSY_CYCLES")] / (xcd_cycles * 1024)   # 1024 SIMDs
    flop = mean[(k, "SQ_INSTS_MFMA")] * MFMA_FLOP
    mfma_util_insts = mean[(k, "SQ_INSTS_MFMA")] * MFMA_CYCLES / (1024 * dur[k] * 2.4e9)   # at the 2.4 GHz peak clock
    fetch, write = mean[(k, "FETCH_SIZE")] * 1024, mean[(k, "WRITE_SIZE")] * 1024
    hbm = 2 * fetch + write   # gfx950: FETCH_SIZE reports half of wide (16 B/lane) reads (MI355X_MICROARCH.md, HBM)
    l2 = (mean[(k, "TCC_HIT_sum")] / (mean[(k, "TCC_HIT_sum")] + mean[(k, "TCC_MISS_sum")])) if (k, "TCC_HIT_sum") in mean else float("nan")
    wc = mean[(k, "SQ_WAVE_CYCLES")]
    lines += [f"### {k}",
              f"* effective clock {clock/1e9:.2f} GHz; issued MFMA work {flop/1e12:.3f} TFLOP (incl. zero padding) "
              f"= {flop/dur[k]/1e12:.1f} TFLOP/s",
              f"* MFMA pipe utilisation: {100*mfma_util:.1f} % from SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE/8) (the GRBM "
              f"figure is taken in a slower counter-collection run, so this ratio and the 'effective clock' are indicative only); "
              f"SQ_INSTS_MFMA x {MFMA_CYCLES} cycles / (1024 SIMDs x kernel-trace duration x 2.4 GHz) = {100*mfma_util_insts:.1f} % is "
              f"an UPPER bound since round 2: the instruction count mixes {MFMA_CYCLES}-cycle products with the 8-cycle 4x4x1 "
              f"MFMAs of the f16f6 gather (6 x 256 per 32-point tile) and the 64-cycle fp32 bias MFMAs",
              f"* wave time: issue-stall {100*mean[(k,'SQ_WAIT_INST_ANY')]/wc:.0f} %, waitcnt/barrier "
              f"{100*mean[(k,'SQ_WAIT_ANY')]/wc:.0f} %, issuing {100*mean[(k,'SQ_ACTIVE_INST_ANY')]/wc:.0f} % "
              f"(VALU {100*mean.get((k,'SQ_ACTIVE_INST_VALU'),0)/wc:.0f} %, LDS {100*mean.get((k,'SQ_ACTIVE_INST_LDS'),0)/wc:.0f} %, "
              f"VMEM {100*mean.get((k,'SQ_ACTIVE_INST_VMEM'),0)/wc:.0f} %); LDS-issue stall {100*mean.get((k,'SQ_WAIT_INST_LDS'),0)/wc:.0f} %",
              f"* instructions per launch: MFMA {mean[(k,'SQ_INSTS_MFMA')]:.3g}, VALU {mean[(k,'SQ_INSTS_VALU')]:.3g}, "
              f"LDS {mean.get((k,'SQ_INSTS_LDS'),0):.3g}, VMEM {mean.get((k,'SQ_INSTS_VMEM'),0):.3g}",
              f"* LDS bank-conflict cycles {mean[(k,'SQ_LDS_BANK_CONFLICT')]:.3g} of {mean[(k,'SQ_LDS_IDX_ACTIVE')]:.3g} active",
              f"* L2 hit rate {100*l2:.1f} %; memory-side traffic: FETCH_SIZE {fetch/1e9:.2f} GB (x2 correction -> "
              f"{2*fetch/1e9:.2f} GB), WRITE_SIZE {write/1e9:.2f} GB -> {hbm/1e9:.2f} GB per launch "
              f"({hbm/dur[k]/1e12:.2f} TB/s; includes Infinity-Cache hits and scratch spill traffic)", ""]
    if (k, "TA_TA_BUSY_sum") in mean:   # texture-addresser occupancy: summed over the TA instances (one per CU)
        ta = mean[(k, "TA_TA_BUSY_sum")]
        lines += [f"* texture addresser (vector-memory issue path: gathers + the weight stream's LDS-DMA): TA_TA_BUSY_sum {ta:.3g} cycles "
                  f"over 256 CUs = {100 * ta / 256 / xcd_cycles:.0f} % of the kernel's cycles per CU"
                  + (f"; TA_BUSY_avr {mean[(k, 'TA_BUSY_avr')]:.3g}, TA_BUSY_max {mean[(k, 'TA_BUSY_max')]:.3g}" if (k, "TA_BUSY_avr") in mean else ""), ""]
    if k == "render":
        out_json = {"kernel": f"render_kernel<{HEAD}, {prec}>", "hbm_bytes_per_launch": hbm, "fetch_size_bytes_raw": fetch,
                    "write_size_bytes": write, "note": "2*FETCH_SIZE + WRITE_SIZE (gfx950 correction), rocprofv3 --pmc, "
                    "separate passes; counts L2 memory-side requests incl. Infinity-Cache hits and scratch traffic",
                    "avg_duration_s": dur[k], "mfma_util": mfma_util_insts, "mfma_util_counter_ratio": mfma_util}
open(f"profiles/{tag}_pmc_summary.md", "w").write("\n".join(lines) + "\n")
json.dump(out_json, open(f"profiles/{round_tag}_render_kernel_hbm_bytes_{prec}.json", "w"), indent=1)
print("\n".join(lines[-16:]))
