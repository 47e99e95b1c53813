#!/usr/bin/env python3
"""One-time tool of round 6 (VERDICT r05 "next" #6): move the retired experiment macros of the fused kernels out of the product
sources.  Reads csrc/ as of the round-5 commit (git show <rev>:...), resolves every conditional of the macro groups below as
"undefined" (tools/unifdef_lite.py), applies the stamp-macro refactor, and writes

    the cleaned sources                      -> neural-jacobian-field_amd/csrc/   (with --write-clean)
    one patch per experiment group           -> tools/probes/<group>.patch        (cleaned source -> source with that experiment)

so that `patch -p1 < tools/probes/<group>.patch` + `-D<MACRO>` rebuilds an experiment library.  The table of which line of
profiles/*ablate*.txt retired each group is in tools/README.md."""
import difflib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import unifdef_lite  # noqa: E402

REV = "6539b58"   # round 5's last commit
FILES = ["neural-jacobian-field_amd/csrc/njf_device.h", "neural-jacobian-field_amd/csrc/njf_kernels.hip"]
GROUPS = {
    "async_stream_8wave": ["NJF_ASYNC_STREAM", "NJF_WAVES"],
    "f16_ride": ["NJF_F16_RIDE", "NJF_F16_RIDE_DEPTH"],
    "f16_gather_pin_depth": ["NJF_F16_GATHER_PIN", "NJF_F16_GATHER_DEPTH"],
    "f16_prefetch2": ["NJF_F16_PREFETCH2"],
    "gather_dephase": ["NJF_GATHER_DEPHASE"],
    "f16_dma_place": ["NJF_F16_DMA_PLACE"],
    "f16_setprio": ["NJF_F16_SETPRIO"],
    "bias_form": ["NJF_BIAS_VALU", "NJF_BIAS_MFMA_ALL"],
    "project_direct": ["NJF_PROJECT_DIRECT"],
    "upsample_per_texel": ["NJF_UPSAMPLE_PER_TEXEL"],
    "sin_polynomial": ["NJF_SIN_POLYNOMIAL"],
    "footprint_ieee_div": ["NJF_FOOTPRINT_IEEE_DIV"],
    "gather_form": ["NJF_GATHER_ALWAYS_HALF", "NJF_GATHER_ALWAYS_QUAD"],
    "train_no_af": ["NJF_TRAIN_NO_AF"],
    "f16_share_switch": ["NJF_F16_SHARE_D"],
    "ablate_split_cvt6_afrag": ["NJF_ABLATE_SPLIT", "NJF_ABLATE_CVT6", "NJF_ABLATE_AFRAG", "NJF_ABLATE_AFRAG_HALF"],
    "ablate_gather_detail": ["NJF_ABLATE_GATHER_ADDR", "NJF_ABLATE_GATHER_HALFBYTES", "NJF_ABLATE_GATHER_NOLOAD",
                             "NJF_ABLATE_GATHER_NOFOLD"],
    "ablate_one_wg_per_cu": ["NJF_ABLATE_ONE_WG_PER_CU"],
}


def stamp_refactor(path: str, text: str) -> str:
    """the NJF_STAMP_* macro forms of round 6 (csrc/njf_device.h) instead of #ifdef NJF_STAMPS blocks at every site"""
    with open(os.path.join(ROOT, "tools", "probes", "stamp_refactor.json")) as f:
        table = json.load(f)
    for old, new in table[os.path.basename(path)]:
        if old not in text:
            raise SystemExit(f"stamp refactor: anchor not found in {path}: {old[:60]!r}")
        text = text.replace(old, new)
    return text


# definitions that only an experiment macro's code calls: wrapped in that macro first, so that they leave with it
WRAP = {
    "njf_kernels.hip": [("NJF_PROJECT_DIRECT", "// One wave: 32 texels x 128 channels, K = 512 swept two at a time straight from global memory",
                         "// Split-precision variant (PREC_F16X2): one wave = 64 texels")],
    "njf_device.h": [("NJF_F16_RIDE", "// PREC_F16, fc_0 chunk of a block WITH the next block's gather riding on it",
                      "// Which form a network uses follows its MFMA precision")],
}


def variant(path: str, undef) -> str:
    src = subprocess.run(["git", "show", f"{REV}:{path}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout
    for macro, first, after in WRAP[os.path.basename(path)]:
        i, j = src.index(first), src.index(after)
        src = src[:i] + f"#ifdef {macro}\n" + src[i:j].rstrip("\n") + f"\n#endif\n\n" + src[j:]
    return stamp_refactor(path, "".join(unifdef_lite.run(src.splitlines(keepends=True), set(undef))))


def main():
    every = [m for g in GROUPS.values() for m in g]
    clean = {p: variant(p, every) for p in FILES}
    if "--write-clean" in sys.argv:
        for p, t in clean.items():
            with open(os.path.join(ROOT, p), "w") as f:
                f.write(t)
    os.makedirs(os.path.join(ROOT, "tools", "probes"), exist_ok=True)
    for name, macros in GROUPS.items():
        keep = [m for m in every if m not in macros]
        chunks = []
        for p in FILES:
            v = variant(p, keep)
            chunks += list(difflib.unified_diff(clean[p].splitlines(keepends=True), v.splitlines(keepends=True), "a/" + p, "b/" + p))
        with open(os.path.join(ROOT, "tools", "probes", name + ".patch"), "w") as f:
            f.write(f"# experiment group {name}: {' '.join('-D' + m for m in macros)} (round-5 sources {REV}; tools/make_probe_patches.py)\n")
            f.writelines(chunks)
        print(name, sum(1 for c in chunks if c.startswith("+") and not c.startswith("+++")), "added lines")


if __name__ == "__main__":
    main()
