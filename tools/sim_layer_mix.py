#!/usr/bin/env python3
"""Per-LAYER precision mixes for the proposal network (VERDICT r02 "next" #4), simulated exactly on the CPU like
tools/sim_split_precision.py: which of the 11 128-wide layers of a ResnetFC (lin_in + 5 x (fc_0, fc_1)) could take the
fp6-corrected product (issue factor 1.5) instead of the 2 x f16 split (issue factor 3) before the density error moves the
resampled bins too far?  Output = |error| of the density PRE-ACTIVATION against float64 (sigma = exp(pre - 1): an absolute
error of the pre-activation is a relative error of sigma).

Yardsticks (measured on the GPU, profiles/r02_ab_variants.txt section 3): all 11 layers on fp6 -> proposal pass 3.16 instead
of 3.80 ms, depth 3.0e-4 against a bound of 1.0e-4 and optical flow 7.8e-4 against 3.6e-4: the density error has to come
down by 2-3x, i.e. to <= 1.7-2.5e-6 rms here.  Projected time = 3.80 - 0.64 * (fp6 layers / 11) ms (chunks are equal)."""
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import sim_split_precision as S  # noqa: E402
from neural_jacobian_field_amd import synthetic  # noqa: E402

D = torch.float64
NAMES = ["lin_in"] + [f"blocks.{i}.fc_{j}" for i in range(5) for j in range(2)] + ["lin_out"]


def matmul(w, x, mode):
    if mode in ("f64", "f32", "f16", "f16x2", "f16+f6", "f16+f8"):
        return S.matmul(w, x, mode)
    wh, wl = S.split(w)
    xh, xl = S.split(x)
    main = xh @ wh.t()
    pad = (-w.shape[1]) % 32
    z = (lambda t: torch.nn.functional.pad(t, (0, pad))) if pad else (lambda t: t)
    q = lambda t: S.quant_block(z(t), -1, **S.FP6)
    if mode == "wlo6":   # W_lo * x_hi in fp6, W_hi * x_lo in f16 (issue factor 2.25)
        return main + xl @ wh.t() + q(xh) @ q(wl).t()
    if mode == "xlo6":   # W_hi * x_lo in fp6, W_lo * x_hi in f16
        return main + q(xl) @ q(wh).t() + xh @ wl.t()
    raise ValueError(mode)


def net(p, z, x, modes):
    md = dict(zip(NAMES, modes))
    lin = lambda n, v: matmul(p[n + ".weight"].to(D), v, md[n]) + p[n + ".bias"].to(D)
    h = lin("lin_in", x)
    for i in range(5):
        if i < 3:
            h = h + (z @ p[f"lin_z.{i}.weight"].to(D).t() + p[f"lin_z.{i}.bias"].to(D))
        n_ = lin(f"blocks.{i}.fc_0", torch.relu(h))
        h = h + lin(f"blocks.{i}.fc_1", torch.relu(n_))
    return lin("lin_out", torch.relu(h))


def main():
    torch.manual_seed(0)
    p = {k[4:]: v for k, v in synthetic.seeded_state_dict(synthetic.resnet_fc_shapes("net.", 63, 512, 1), seed=1).items()}
    pts = 8192
    z = torch.randn(pts, 512, dtype=D)
    xyz = torch.rand(pts, 3, dtype=D) * 2 - 1
    s = (2 * math.pi * xyz)[..., None] * (2.0 ** torch.arange(10, dtype=D))
    x = torch.cat([torch.sin(s).reshape(pts, -1), torch.cos(s).reshape(pts, -1), xyz], -1)
    ref = net(p, z, x, ["f64"] * 12)

    def rep(label, modes, fp6_layers, factor=None):
        e = (net(p, z, x, modes) - ref).abs()
        ms = 3.80 - 0.64 * fp6_layers / 11 if factor is None else factor
        print(f"{label:46s} max {e.max().item():.2e}  rms {e.pow(2).mean().sqrt().item():.2e}   projected proposal pass {ms:.2f} ms")

    print(f"# density pre-activation error vs float64, {pts} points, seeded N(0, 0.05) weights; |pre| max {ref.abs().max().item():.2f}")
    rep("f16x2 on every layer (shipped)", ["f16x2"] * 12, 0)
    rep("fp6 corrections on all 11 wide layers", ["f16+f6"] * 11 + ["f16x2"], 11)
    for k in (1, 3, 5, 7, 9):
        rep(f"fp6 on the FIRST {k} wide layers", ["f16+f6"] * k + ["f16x2"] * (12 - k), k)
    for k in (1, 3, 5, 7, 9):
        rep(f"fp6 on the LAST {k} wide layers", ["f16x2"] * (11 - k) + ["f16+f6"] * k + ["f16x2"], k)
    rep("only W_lo*x_hi in fp6 (W_hi*x_lo in f16), 11", ["wlo6"] * 11 + ["f16x2"], 0, 3.80 - 0.64 * 0.5)
    rep("only W_hi*x_lo in fp6 (W_lo*x_hi in f16), 11", ["xlo6"] * 11 + ["f16x2"], 0, 3.80 - 0.64 * 0.5)
    rep("hi*hi only (plain f16)", ["f16"] * 12, 0, float("nan"))


if __name__ == "__main__":
    main()
