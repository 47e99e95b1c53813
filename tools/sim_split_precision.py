#!/usr/bin/env python3
"""CPU study for the next kernel round: how accurate is a ResnetFC whose matrix products are evaluated as

    f32      exact fp32 products, fp32 accumulation                       (v_mfma_f32_32x32x2_f32)
    f16x2    hi*hi + hi*lo + lo*hi, all operands fp16                     (the shipped default, 3 f16 MFMAs per block)
    f16+f8   hi*hi in fp16, the two correction products in block-scaled fp8 e4m3 (32-element blocks, power-of-two scales:
             v_mfma_scale_f32_32x32x64_f8f6f4 computes both corrections of TWO K-steps in the time of two f16 MFMAs)
    f16+f6   the same with fp6 e2m3 corrections (half the time again)
    f16      hi*hi only (what a plain fp16 kernel would do)

against a float64 evaluation of the same network?  Operand rounding is simulated exactly; accumulation is done in
float64, so the figures isolate what the operand formats cost (fp32 accumulation adds ~1e-7 to each).  Runs anywhere
(no GPU).  Not part of the product path."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_jacobian_field_amd import synthetic  # noqa: E402

torch.manual_seed(0)
D = torch.float64


def f16(x):
    return x.to(torch.float32).to(torch.float16).to(D)


def split(x):
    hi = f16(x)
    return hi, f16(x - hi)


def quant_block(x, axis, mant_bits, emax, emin):
    """Block-scaled minifloat along `axis` in blocks of 32: scale = 2^(floor(log2(max|x|)) - emax), element = nearest
    value with `mant_bits` mantissa bits and exponent range [emin, emax] (subnormals below emin)."""
    x = x.movedim(axis, -1)
    shape = x.shape
    xb = x.reshape(*shape[:-1], -1, 32)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    y = xb / scale
    e = torch.floor(torch.log2(y.abs().clamp_min(2.0 ** (emin - 20)))).clamp(min=emin, max=emax)
    q = torch.exp2(e - mant_bits)
    y = torch.clamp(torch.round(y / q) * q, -(2.0 - 2.0 ** -mant_bits) * 2.0 ** emax, (2.0 - 2.0 ** -mant_bits) * 2.0 ** emax)
    return (y * scale).reshape(shape).movedim(-1, axis)


FP8 = dict(mant_bits=3, emax=8, emin=-6)     # e4m3 (OCP: max 448 = 1.75 * 2^8)
FP6 = dict(mant_bits=3, emax=2, emin=0)      # e2m3 (max 7.5)


def matmul(w, x, mode):
    """y[P,out] = x[P,in] @ w[out,in]^T with the operand formats of `mode`."""
    if mode == "f64":
        return x @ w.t()
    if mode == "f32":
        return x.to(torch.float32).to(D) @ w.to(torch.float32).to(D).t()
    wh, wl = split(w)
    xh, xl = split(x)
    main = xh @ wh.t()
    if mode == "f16":
        return main
    if mode == "f16x2":
        return main + xl @ wh.t() + xh @ wl.t()
    fmt = FP8 if mode == "f16+f8" else FP6
    pad = (-w.shape[1]) % 32
    if pad:
        z = lambda t: torch.nn.functional.pad(t, (0, pad))
        wh, wl, xh, xl = z(wh), z(wl), z(xh), z(xl)
    q = lambda t: quant_block(t, -1, **fmt)       # blocks run along K for both operands
    return main + q(xl) @ q(wh).t() + q(xh) @ q(wl).t()


def resnet_fc(p, z, x, mode):
    lin = lambda n, v: matmul(p[n + ".weight"].to(D), v, mode) + p[n + ".bias"].to(D)
    h = lin("lin_in", x)
    for i in range(5):
        if i < 3:
            h = h + (z @ p[f"lin_z.{i}.weight"].to(D).t() + p[f"lin_z.{i}.bias"].to(D))   # hoisted: an fp32 map, exact here
        net = lin(f"blocks.{i}.fc_0", torch.relu(h))
        h = h + lin(f"blocks.{i}.fc_1", torch.relu(net))
    return lin("lin_out", torch.relu(h))


def main():
    shapes = synthetic.resnet_fc_shapes("net.", 63, 512, 24)
    for label, std in (("seeded N(0, 0.05) weights (parity-test scale)", 0.05), ("reference init of the Jacobian head, N(0, 1e-4)", None)):
        p = {k[4:]: v for k, v in synthetic.seeded_state_dict(shapes, seed=1).items()}
        if std is None:
            g = torch.Generator().manual_seed(3)
            p = {k: torch.randn(v.shape, generator=g) * 1e-4 for k, v in p.items()}
        pts = 4096
        z = torch.randn(pts, 512, dtype=D)
        xyz = torch.rand(pts, 3, dtype=D) * 2 - 1
        freqs = 2.0 ** torch.arange(10, dtype=D)
        s = (2 * math.pi * xyz)[..., None] * freqs
        x = torch.cat([torch.sin(s).reshape(pts, -1), torch.cos(s).reshape(pts, -1), xyz], -1)
        ref = resnet_fc(p, z, x, "f64")
        print(label)
        for mode in ("f32", "f16x2", "f16+f8", "f16+f6", "f16"):
            out = resnet_fc(p, z, x, mode)
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            rms = ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            print(f"  {mode:7s} max-norm error {err:.2e}   rms {rms:.2e}")


if __name__ == "__main__":
    main()
