"""Action-mode training support: gradients of the rendered optical flow w.r.t. the Jacobian head.

Reference contract: in ``dataset.mode == "action"`` everything except the Jacobian head is frozen
(``models/model_wrapper.py:75-85``) and the loss is ``0.01 * mse(optical_flow, target)`` (``:148-160``), so density,
weights and sample placement are constants of the backward pass (build step 6).

Forward: the fused HIP kernels, with the final pass additionally dumping the ReLU'd input of every layer of the
Jacobian ``ResnetFC`` (``njf_render_forward`` with ``jac_act``/``jac_pe``/``foot_*`` outputs).
Backward: the data-gradient chain of each ``ResnetFC`` (11 transposed-weight products with ReLU masks and residual adds)
is ONE fused HIP launch, ``njf_resnetfc_backward``, that keeps the gradient in MFMA accumulators and emits per layer the
matrix its weight gradient contracts with -- tested against the GEMM-by-GEMM form and against autograd of the CPU oracle;
the texel scatter of the ``lin_z`` / feature gradients (grid_sample's input gradient) is ``njf_scatter_footprint``.
The weight gradients are sums over ALL points of outer products, i.e. GEMMs with K = points: ONE batched library GEMM on
the emitted matrices (a per-workgroup accumulation would need 11 x 64 KB of partial sums per tile).  The ``jacobian_transformer`` head is
differentiated by recomputing it (original parameterisation, library ops) on the dumped encoding + footprint.
Perception mode (every parameter trains; rgb / depth / per-level weights carry the graph) is ``FieldFunction``.
"""

from __future__ import annotations

import math
from typing import Callable, Dict, List, Sequence

import os

import torch

from . import hip
from .encoder import FeaturePyramid

JACOBIAN_PARAM_ORDER: List[str] = (
    ["lin_in.weight", "lin_in.bias"]
    + [f"blocks.{b}.{fc}.{wb}" for b in range(5) for fc in ("fc_0", "fc_1") for wb in ("weight", "bias")]
    + [f"lin_z.{i}.{wb}" for i in range(3) for wb in ("weight", "bias")]
    + ["lin_out.weight", "lin_out.bias"]
)

# positional-encoding slot -> reference channel (csrc/njf_kernels.hip::pack_source, kind 1); slot 63 is the bias
_PE_SLOT_TO_CHANNEL = list(range(30)) + [60, 61] + list(range(30, 60)) + [62]
_slot_index_cache: Dict[torch.device, torch.Tensor] = {}


def _slot_to_channel(device: torch.device) -> torch.Tensor:
    """Device copy of _PE_SLOT_TO_CHANNEL (made once per device: a host->device copy per backward pass would stall)."""
    if device not in _slot_index_cache:
        _slot_index_cache[device] = torch.tensor(_PE_SLOT_TO_CHANNEL, device=device)
    return _slot_index_cache[device]


def _tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a^T b for tall-skinny operands (a [K,M], b [K,N], K = points or texels >> M, N <= 512): the weight-gradient
    GEMMs.  The output has only a handful of macro-tiles, so a plain GEMM call serialises the whole K loop on a few
    workgroups; splitting K into batches fills the chip, and the partial products are summed afterwards (measured on
    an MI355X at the reference batch shape: action-mode step 15.5 -> 10.0 ms, perception-mode step 30.3 -> 21.2 ms)."""
    k = a.shape[0]
    groups = 128
    while groups > 1 and (k % groups or k // groups < 256):
        groups //= 2
    if groups == 1:
        return a.t() @ b
    return torch.bmm(a.reshape(groups, k // groups, -1).transpose(1, 2), b.reshape(groups, k // groups, -1)).sum(0)


def _colsum(x: torch.Tensor) -> torch.Tensor:
    """Column sums of a tall matrix [K, n] with n small (bias gradients: K = points).  torch reduces the leading dimension of a
    [459 k, 3] tensor on FOUR workgroups (153 us on an MI355X); in two stages the first one fills the chip."""
    k = x.shape[0]
    groups = 1024
    while groups > 1 and (k % groups or k // groups < 64):
        groups //= 2
    if groups == 1 or x.dim() != 2:
        return x.sum(0)
    return x.reshape(groups, k // groups, x.shape[1]).sum(1).sum(0)


_LAYER_NAMES = [f"blocks.{l // 2}.fc_{l % 2}" for l in range(10)]   # the layer whose ReLU'd input is act[l]


def _tn_batched(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a[l]^T b[l] for every layer l at once (a, b [L,K,128], K = points): ONE batched GEMM for all weight gradients of a
    ResnetFC.  K is split into groups like `_tn` (the [128,128] outputs are single macro-tiles: without the split each
    product would serialise its K loop on one workgroup) and the partial products are summed."""
    layers, k, n = a.shape
    groups = 64
    while groups > 1 and (k % groups or k // groups < 256):
        groups //= 2
    out = torch.bmm(a.reshape(layers * groups, k // groups, n).transpose(1, 2), b.reshape(layers * groups, k // groups, -1))
    return out.reshape(layers, groups, n, -1).sum(1)


# How the reference itself trains: ``torch.set_float32_matmul_precision("high")`` (train.py:64-65) -- every GEMM of its training step,
# forward and backward, runs on TF32 products (10 mantissa bits per operand).  The same process-wide switch selects the TF32-class
# forms of THIS backward pass: with "auto" (the default of both settings below) they are used exactly when the caller has relaxed
# torch's matmul precision the way train.py does AND the network's forward pass runs in a split precision (i.e. inside fp16's
# range: a network forced to exact fp32 products, by hand or by the range guard of Model._maybe_check_range, keeps the exact
# backward).  With torch's own default ("highest") nothing changes: exact fp32 products, fp32 storage.
def reference_allows_reduced_products() -> bool:
    return torch.get_float32_matmul_precision() != "highest"


def _auto(setting: str, reduced: str, forward_precision) -> str:
    if setting != "auto":
        return setting
    return reduced if (reference_allows_reduced_products() and forward_precision != "f32") else "f32"


# Product form of the fused backward chain (njf_resnetfc_backward): "f32" = exact fp32 MFMA products (what every gradient row of
# the test-suite is held to its fp64 floor with), "f16x2" = split fp16 products on power-of-two-scaled gradients (fp32-class,
# 2^-22 per product where the reference's own training runs TF32 products; measured against the exact chain in
# tests/test_training_gpu.py::test_backward_chain_f16x2_against_exact), "auto" (default) = see above.  Process-wide, like the
# reference's torch.set_float32_matmul_precision.
_BACKWARD_PRECISION = os.environ.get("NJF_BACKWARD_PRECISION", "auto")


def set_backward_precision(name: str) -> None:
    global _BACKWARD_PRECISION
    if name != "auto" and name not in hip.BACKWARD_PRECISIONS:
        raise ValueError(f"backward precision must be 'auto' or one of {hip.BACKWARD_PRECISIONS} (got {name!r})")
    _BACKWARD_PRECISION = name


def backward_precision(forward_precision=None) -> str:
    """The chain's product form for a network whose forward pass ran in ``forward_precision`` (None: unknown, taken as split)."""
    return _auto(_BACKWARD_PRECISION, "f16x2", forward_precision)


# Storage of what the weight-gradient GEMMs read (the K = points contractions deltas[l+1]^T act[l]: 2.95 GB per network and operand
# on SURVEY's C4 shard).  "f32": fp32 activations from the training forward, fp32 deltas from the chain, fp32 GEMMs.
# "f16": the forward dumps the activations as fp16, the chain writes deltas x 2^k as fp16 (k from max|d_out|; its masks make
# it independent of the activations' format), and the GEMMs run with fp16 operands and FP32 ACCUMULATION (torch.bmm(..., out_dtype=
# float32)) -- 10 mantissa bits per operand, which is what the reference's own training arithmetic keeps (TF32: train.py:64-65).
# "auto" (default): see above.  Stated tolerance: tests/test_training_gpu.py::test_f16_training_storage_against_fp32_storage.
_STORAGE_PRECISION = os.environ.get("NJF_TRAINING_STORAGE", "auto")


def set_storage_precision(name: str) -> None:
    global _STORAGE_PRECISION
    if name not in ("auto", "f32", "f16"):
        raise ValueError(f"training storage must be 'auto', 'f32' or 'f16' (got {name!r})")
    _STORAGE_PRECISION = name


def storage_precision(forward_precision=None) -> str:
    return _auto(_STORAGE_PRECISION, "f16", forward_precision)


def activation_dump_dtype(forward_precision=None) -> torch.dtype:
    return torch.float16 if storage_precision(forward_precision) == "f16" else torch.float32


# Safety net of the TF32-class forms (fp16 has a range, TF32 does not): the chain runs on gradients scaled so that max|d_out| = 64
# (2^9 of head room) and the 16-bit storage holds deltas x 2^k as halves -- a network whose transposed weights amplify a gradient
# by more than that overflows to inf / NaN where the exact backward would not.  Every reduced backward call folds "were its results
# finite?" into ONE device scalar (no host synchronisation); Model._maybe_check_range reads it at its periodic check (which
# synchronises anyway), warns, and switches both settings to exact fp32 for the rest of the process.
_reduced_flags: Dict[tuple, torch.Tensor] = {}


def _note_reduced_results(*tensors: torch.Tensor) -> None:
    dev = tensors[0].device
    key = hip.device_key(dev)
    flag = _reduced_flags.get(key)
    if flag is None:
        flag = _reduced_flags[key] = torch.zeros((), dtype=torch.float32, device=dev)
    total = tensors[0].sum()
    for t in tensors[1:]:
        total = total + t.sum()
    flag.add_(total * 0.0)          # inf * 0 = NaN * 0 = NaN: the flag turns NaN once and stays


def reduced_backward_overflowed(device=None, reset: bool = True) -> bool:
    """Did any TF32-class backward call since the last query produce a non-finite gradient?  (One host synchronisation.)"""
    hit = False
    for key, flag in list(_reduced_flags.items()):
        if device is not None and key != hip.device_key(torch.device(device)):
            continue
        if bool(torch.isnan(flag).item()):
            hit = True
            if reset:
                flag.zero_()
    return hit


def _tn_batched_f16(a16: torch.Tensor, b16: torch.Tensor) -> torch.Tensor:
    """a[l]^T b[l] for fp16 operands [L,K,128] with fp32 accumulation and fp32 results (K split into groups as in _tn_batched)."""
    layers, k, n = a16.shape
    groups = 64
    while groups > 1 and (k % groups or k // groups < 256):
        groups //= 2
    out = torch.bmm(a16.reshape(layers * groups, k // groups, n).transpose(1, 2), b16.reshape(layers * groups, k // groups, -1),
                    out_dtype=torch.float32)
    return out.reshape(layers, groups, n, -1).sum(1)


def resnetfc_backward_chain(p: Dict[str, torch.Tensor], d_out: torch.Tensor, act: torch.Tensor, want_colsum: bool = False,
                            mask: torch.Tensor = None, forward_precision=None):
    """The data-gradient chain of one ResnetFC as ONE fused launch (njf_resnetfc_backward): deltas [11,P,128], see
    include/njf_hip.h for the meaning of each slice (with ``want_colsum`` also their column sums [11,128], from the
    kernel's per-tile partial sums).  The transposed weights are packed per call (eleven small launches: the weights
    change with every optimiser step)."""
    w_t = torch.empty(hip.RESNET_BACKWARD_W_FLOATS, dtype=torch.float32, device=d_out.device)
    chain = backward_precision(forward_precision)
    hip.pack_resnetfc_backward(p, "", w_t, precision=chain)
    return hip.resnetfc_backward(d_out, act, w_t, want_colsum=want_colsum, mask=mask, precision=chain)


def resnetfc_backward(p: Dict[str, torch.Tensor], d_out: torch.Tensor, act: torch.Tensor, pe: torch.Tensor,
                      foot_idx: torch.Tensor, foot_w: torch.Tensor, feats_flat: torch.Tensor,
                      d_feats: torch.Tensor = None, samples_per_ray: int = 1, mask: torch.Tensor = None,
                      latent_constants: torch.Tensor = None, forward_precision=None) -> Dict[str, torch.Tensor]:
    """Backward pass of one ResnetFC (resnet_fc.py:130-154) from the activations the HIP forward dumped.  ``mask`` [11,P,4] int32:
    the ReLU masks the same forward dumped -- the fused data-gradient chain then reads 16 instead of 512 bytes per point and layer
    (the weight-gradient GEMMs below are what still reads ``act``).

    ``p``: the net's parameters by reference name; ``d_out`` [P, d_out]; ``act`` [11,P,128] (ReLU'd layer inputs),
    ``pe`` [P,64] (slot order), ``foot_idx``/``foot_w`` [P,4] bilinear footprint on the flattened texel grid,
    ``feats_flat`` [T,512] encoder features, channels last.  Returns the parameter gradients; when ``d_feats`` [T,512]
    is given, the gradient w.r.t. the encoder features is accumulated into it (grid_sample's input gradient followed
    by lin_z's, in hoisted order: scatter the [P,128] latent gradient onto the texels, then one GEMM per lin_z).
    ``latent_constants`` [B,A]: latent columns beyond the encoder channels that are constant per batch element (the robot
    action of ``flow_mlp``, action_decoder_flow.py:168-172; points are batch-major) -- ``lin_z.*.weight`` is then [128, C + A]
    and its last A columns receive sum_b a[b] (x) sum_{p in b} delta[p].

    The data-gradient chain (transposed-weight products, ReLU masks, residual adds of all 11 layers) is one HIP launch
    that keeps the gradient in MFMA accumulators and emits, per layer, the matrix that layer's weight gradient contracts
    with; the weight gradients themselves are sums over ALL points of outer products (K = points) and are one batched
    library GEMM on those matrices; the bias gradients one column-sum reduction."""
    grads: Dict[str, torch.Tensor] = {}
    if act.dtype == torch.float16:
        # 16-bit training storage: fp16 activations (dumped by the forward) x fp16 deltas (scaled by 2^k), fp32 accumulation
        if mask is None:
            raise ValueError("the 16-bit training storage needs the forward's ReLU masks")
        w_t = torch.empty(hip.RESNET_BACKWARD_W_FLOATS, dtype=torch.float32, device=d_out.device)
        chain = backward_precision(forward_precision)
        hip.pack_resnetfc_backward(p, "", w_t, precision=chain)
        latent, deltas16, sums, unscale = hip.resnetfc_backward_f16_storage(d_out, w_t, mask, precision=chain)
        w_grads = _tn_batched_f16(deltas16[1:11], act[0:10]) * unscale
        deltas_latent = latent                                        # [3,P,128]: gradients w.r.t. the three hoisted latents
        delta0 = latent[0]
        grads["lin_out.weight"] = _tn(d_out, act[10].float())
        _note_reduced_results(w_grads, sums)
    else:
        deltas, sums = resnetfc_backward_chain(p, d_out, act, want_colsum=True, mask=mask,
                                               forward_precision=forward_precision)   # sums [11,128]: deltas[l+1] <-> bias of layer l
        w_grads = _tn_batched(deltas[1:11], act[0:10])                   # [10,128,128]
        if backward_precision(forward_precision) != "f32":
            _note_reduced_results(sums)
        deltas_latent = deltas[0:6:2]
        delta0 = deltas[0]
        grads["lin_out.weight"] = _tn(d_out, act[10])
    for l, name in enumerate(_LAYER_NAMES):
        grads[name + ".weight"] = w_grads[l]
        grads[name + ".bias"] = sums[l + 1]
    grads["lin_out.bias"] = _colsum(d_out)
    # lin_z[blk](bilinear(F)) was added to h in front of block blk: its gradient is deltas[2 blk].  grid_sample's input
    # gradient of all three latents is ONE launch into [T,384] (points are ray-major: neighbouring samples share texels),
    # the three weight gradients one GEMM against the channels-last features
    d_g = torch.zeros(feats_flat.shape[0], 3 * 128, dtype=torch.float32, device=d_out.device)
    hip.scatter_footprint(deltas_latent, foot_idx, foot_w, d_g, run_length=samples_per_ray)
    wz_grad = _tn(d_g, feats_flat)                                   # [384,512]
    if latent_constants is not None:
        nb = latent_constants.shape[0]
        per_image = deltas_latent.float().reshape(3, nb, -1, 128).sum(2)                     # [3,B,128]
        const_grad = torch.einsum("lbf,ba->lfa", per_image, latent_constants.to(per_image.dtype))   # [3,128,A]
    n_feat = feats_flat.shape[1]
    for blk in range(3):
        w_blk = wz_grad[128 * blk:128 * (blk + 1)]
        grads[f"lin_z.{blk}.weight"] = w_blk if latent_constants is None else torch.cat([w_blk, const_grad[blk]], dim=1)
        grads[f"lin_z.{blk}.bias"] = sums[2 * blk]
    if d_feats is not None:
        d_feats.addmm_(d_g, torch.cat([p[f"lin_z.{blk}.weight"][:, :n_feat] for blk in range(3)]))
    d_in = _tn(delta0, pe)  # [128, 64] in slot order
    grads["lin_in.weight"] = d_in.new_zeros(d_in.shape[0], 63).index_copy_(
        1, _slot_to_channel(d_in.device), d_in[:, :63])
    grads["lin_in.bias"] = d_in[:, 63].clone()
    return grads


def resnetfc_backward_reference_chain(p: Dict[str, torch.Tensor], d_out: torch.Tensor, act: torch.Tensor) -> torch.Tensor:
    """The same deltas as ``resnetfc_backward_chain`` from library GEMMs + njf_relu_backward (the round-1 form of the
    backward pass): kept as the comparator of the fused kernel's GPU test."""
    deltas = torch.empty_like(act)
    delta, _ = hip.relu_backward(d_out @ p["lin_out.weight"], act[10], want_colsum=False)
    deltas[10] = delta
    for blk in range(4, -1, -1):
        d_net, _ = hip.relu_backward(delta @ p[f"blocks.{blk}.fc_1.weight"], act[2 * blk + 1], want_colsum=False)
        delta, _ = hip.relu_backward(d_net @ p[f"blocks.{blk}.fc_0.weight"], act[2 * blk], residual=delta, want_colsum=False)
        deltas[2 * blk + 1], deltas[2 * blk] = d_net, delta
    return deltas


def _flat_features(features) -> torch.Tensor:
    """[T,512] channels-last view of the encoder output.  A FeaturePyramid (the un-concatenated latents the forward pass
    hoisted from) is expanded here, once, by njf_upsample_concat -- only the backward pass needs the 512-channel map."""
    if isinstance(features, FeaturePyramid):
        return hip.upsample_concat(features.levels)
    return features.permute(0, 2, 3, 1).reshape(-1, features.shape[1])


def transformer_head(p: Dict[str, torch.Tensor], xyz_features: torch.Tensor, pixel_features: torch.Tensor,
                     heads: int = 8) -> torch.Tensor:
    """ActionDecoderJacobianTransformer.compute_jacobian (action_decoder_jacobian.py:418-446 with
    model_components/transformer.py:38-135) in its ORIGINAL, un-folded parameterisation, as differentiable torch ops.
    Used only by the backward pass, which recomputes the head (~0.3 MMAC/point) on the dumped inputs so that autograd
    yields gradients w.r.t. the reference's parameters directly instead of the folded matrices the kernel evaluates.
    ``p``: parameters by their name relative to the decoder; xyz_features [P,63], pixel_features [P,512] -> [P,3A]."""
    x = torch.nn.functional.linear(torch.cat([xyz_features, pixel_features], dim=-1),
                                   p["jacobian_query_mlp.weight"], p["jacobian_query_mlp.bias"])
    z = p["jacobian_index_embedding"][0]                               # [A, 64] learned tokens, shared by all points
    n_pts, n_tok = x.shape[0], z.shape[0]
    layer = 0
    while f"jacobian_attn_decoder.layers.{layer}.0.norm.weight" in p:
        pre = f"jacobian_attn_decoder.layers.{layer}."
        n = torch.nn.functional.layer_norm(x, x.shape[-1:], p[pre + "0.norm.weight"], p[pre + "0.norm.bias"])
        q = (n @ p[pre + "0.fn.to_q.weight"].t()).reshape(n_pts, heads, -1)
        k, v = (z @ p[pre + "0.fn.to_kv.weight"].t()).chunk(2, dim=-1)
        k, v = k.reshape(n_tok, heads, -1), v.reshape(n_tok, heads, -1)
        attn = torch.softmax(torch.einsum("phd,ahd->pha", q, k) * q.shape[-1] ** -0.5, dim=-1)
        o = torch.einsum("pha,ahd->phd", attn, v).reshape(n_pts, -1)
        x = x + torch.nn.functional.linear(o, p[pre + "0.fn.to_out.0.weight"], p[pre + "0.fn.to_out.0.bias"])
        n = torch.nn.functional.layer_norm(x, x.shape[-1:], p[pre + "1.norm.weight"], p[pre + "1.norm.bias"])
        hid = torch.nn.functional.gelu(torch.nn.functional.linear(n, p[pre + "1.fn.net.0.weight"], p[pre + "1.fn.net.0.bias"]))
        x = x + torch.nn.functional.linear(hid, p[pre + "1.fn.net.3.weight"], p[pre + "1.fn.net.3.bias"])
        layer += 1
    return torch.nn.functional.linear(x, p["jacobian_head.weight"], p["jacobian_head.bias"])


def folded_transformer(p: Dict[str, torch.Tensor], heads: int = 8) -> Dict[str, torch.Tensor]:
    """The FOLDED form of the transformer head that the fused kernels evaluate (decoder.ActionDecoderJacobianTransformer.
    _pack_regular_jacobian restated as differentiable torch ops on the reference's parameters, names relative to the decoder):
    ``mats`` [3,4,64,64] = (Mqk, Nov, W1', W2) per layer, ``biases`` [3,4,64] = (bqk, bo, b1', b2).  Keys / values depend only on
    the learned index embedding, so to_q . K^T / sqrt(d) and V . to_out are 64 x 64 matrices (rows of Mqk / columns of Nov ordered
    head * 8 + key, unused key slots zero); the LayerNorm affines are folded into Mqk / W1'.  The fold's own autograd graph is what
    carries the gradients of the folded matrices (njf_transformer_backward) back to the reference's parameters."""
    z = p["jacobian_index_embedding"][0]                                  # [A, 64]
    a = z.shape[0]
    depth = 0
    while f"jacobian_attn_decoder.layers.{depth}.0.norm.weight" in p:
        depth += 1
    # all layers at once (every step of an action-mode run goes through this graph forwards and backwards: ~3 x fewer launches
    # than a loop over the layers, and the step with this head is bound by its host launches)
    st = lambda suffix: torch.stack([p[f"jacobian_attn_decoder.layers.{l}.{suffix}"] for l in range(depth)])
    g1, be1, g2, be2 = st("0.norm.weight"), st("0.norm.bias"), st("1.norm.weight"), st("1.norm.bias")        # [L, 64]
    to_q, to_kv, to_out, bo = st("0.fn.to_q.weight"), st("0.fn.to_kv.weight"), st("0.fn.to_out.0.weight"), st("0.fn.to_out.0.bias")
    w1, b1, w2, b2 = st("1.fn.net.0.weight"), st("1.fn.net.0.bias"), st("1.fn.net.3.weight"), st("1.fn.net.3.bias")
    inner = to_kv.shape[1] // 2
    dh = inner // heads
    c = to_q.shape[-1]
    kv = torch.einsum("ac,lkc->lak", z, to_kv)                            # [L, A, 2 * H * dh]
    k = kv[..., :inner].reshape(depth, a, heads, dh)                      # [L, A, H, dh]
    v = kv[..., inner:].reshape(depth, a, heads, dh)
    mqk = (dh ** -0.5) * torch.einsum("lahd,lhdc->lhac", k, to_q.reshape(depth, heads, dh, c))          # [L, H, A, 64]
    mqk = torch.nn.functional.pad(mqk, (0, 0, 0, 8 - a)).reshape(depth, heads * 8, c)                   # rows head * 8 + key
    nov = torch.einsum("lchd,lahd->lcha", to_out.reshape(depth, -1, heads, dh), v)                      # [L, 64, H, A]
    nov = torch.nn.functional.pad(nov, (0, 8 - a)).reshape(depth, -1, heads * 8)                        # columns head * 8 + key
    mats = torch.stack([mqk * g1[:, None, :], nov, w1 * g2[:, None, :], w2], dim=1)                     # [L, 4, 64, 64]
    biases = torch.stack([torch.einsum("lrc,lc->lr", mqk, be1), bo, torch.einsum("lrc,lc->lr", w1, be2) + b1, b2], dim=1)
    return {"mats": mats, "biases": biases}


def flat_double(tensors: Sequence[torch.Tensor], requires_grad: bool = False):
    """(flat, views): ONE float64 copy of a list of small tensors -- a cat of the flattened tensors, one conversion, and per-tensor
    views of the result (three launches instead of one conversion per tensor: the transformer head has 41 parameter tensors and a
    step with it is bound by its host launches).  With ``requires_grad`` the FLAT tensor is the autograd leaf."""
    flat = torch.cat([t.detach().reshape(-1) for t in tensors]).double()
    if requires_grad:
        flat.requires_grad_(True)
    with torch.set_grad_enabled(requires_grad):     # (callers run under no_grad: the views must be recorded all the same)
        views = [v.reshape(t.shape) for v, t in zip(flat.split([t.numel() for t in tensors]), tensors)]
    return flat, views


# The fold of the CURRENT parameter values with its autograd graph, built by whoever needs it first: the forward pass's pack
# (decoder.ActionDecoderJacobianTransformer._pack_regular_jacobian, after every optimiser step) or the backward pass.  One entry,
# keyed by the parameters' (storage, version) pairs; the backward pass consumes it (its graph is freed by the differentiation).
_fold_cache: Dict[str, object] = {}


def _fold_key(names: Sequence[str], params: Sequence[torch.Tensor]):
    return tuple(names), tuple((t.data_ptr(), t._version) for t in params)


def transformer_fold(names: Sequence[str], params: Sequence[torch.Tensor], consume: bool = False, heads: int = 8):
    """(flat float64 leaf, folded) for the transformer head's parameters (names relative to the decoder).  ``folded`` carries the
    autograd graph back to ``flat`` when any parameter trains.  ``consume``: the caller differentiates the graph (it is dropped
    from the cache); otherwise the result is kept for the backward pass of the same step."""
    key = _fold_key(names, params)
    hit = _fold_cache.get("entry")
    if hit is not None and hit[0] == key:
        if consume:
            _fold_cache.clear()
        return hit[1], hit[2]
    want_graph = any(t.requires_grad for t in params)
    flat, views = flat_double(params, requires_grad=want_graph)
    with torch.set_grad_enabled(want_graph):
        folded = folded_transformer(dict(zip(names, views)), heads)
    _fold_cache.clear()
    if want_graph and not consume:
        _fold_cache["entry"] = (key, flat, folded)
    return flat, folded


def transformer_head_backward(names: Sequence[str], params: Sequence[torch.Tensor], d_j: torch.Tensor, x: torch.Tensor,
                              pe: torch.Tensor, foot_idx: torch.Tensor, foot_w: torch.Tensor, feats_flat: torch.Tensor,
                              samples_per_ray: int = 1, forward_precision=None):
    """Gradients of the transformer Jacobian head's parameters (names relative to the decoder, any order) from d_j [P,3A], the
    residual stream x [4,P,64] the training forward dumped, the dumped encoding pe [P,64] (slot order) and footprint.
    ONE fused launch for the data-gradient chain (njf_transformer_backward, exact fp32 MFMA on the folded head), one batched
    library GEMM for the K = points weight gradients of the twelve folded matrices, the footprint scatter for the hoisted query
    features, and the fold's autograd graph (64 x 64 matrices) back to the reference's parameterisation.  The chain is exact fp32
    unless ``backward_precision`` selects the split-fp16 form (as for the ResnetFC chain); under the 16-bit training storage (``storage_precision``: "f16", or "auto" with the reference's matmul
    precision "high") the (X, dY) pairs are written as halves and contracted with fp32 accumulation."""
    sizes = [t.numel() for t in params]
    flat, folded = transformer_fold(names, params, consume=True)      # (the forward pass's pack has usually built it already)
    if not flat.requires_grad:                                        # every parameter frozen: nothing to differentiate
        return tuple(torch.zeros_like(t) for t in params)
    mats32, biases32 = folded["mats"].detach().float(), folded["biases"].detach().float()
    head_w = params[list(names).index("jacobian_head.weight")].detach().float()
    dev = d_j.device
    w_t = torch.empty(hip.TRANSFORMER_BACKWARD_W_FLOATS, dtype=torch.float32, device=dev)
    b_t = torch.empty(hip.TRANSFORMER_BACKWARD_B_FLOATS, dtype=torch.float32, device=dev)
    chain = backward_precision(forward_precision)            # exact fp32, or (TF32-class, opt-in / "auto" under matmul precision "high") f16x2
    hip.pack_transformer_backward(mats32, biases32[:, :3].contiguous(), head_w, w_t, b_t, precision=chain)
    keys = params[list(names).index("jacobian_index_embedding")].shape[1]
    half = storage_precision(forward_precision) == "f16"     # 16-bit training storage of what the weight-gradient GEMM reads
    wg_x, wg_dy, dx0, dy_sums, unscale = hip.transformer_backward(x, d_j, keys, w_t, b_t, half_storage=half, precision=chain)
    if chain != "f32":
        _note_reduced_results(dx0, dy_sums)
    if half:
        g_mats = (_tn_batched_f16(wg_dy, wg_x) * unscale).reshape(3, 4, 64, 64)
        _note_reduced_results(g_mats)
    else:
        g_mats = _tn_batched(wg_dy, wg_x).reshape(3, 4, 64, 64)             # dY^T X per folded matrix: [out, in]
    g_biases = dy_sums.reshape(3, 4, 64)                                    # (bqk, bo, b1', b2): column sums of the dY (per-tile partials)
    with torch.enable_grad():
        (g_flat,) = torch.autograd.grad([folded["mats"], folded["biases"]], [flat], [g_mats.double(), g_biases.double()])
    # one conversion for all parameters; the gradients returned are views of this buffer
    g32 = g_flat.float()
    out = dict(zip(names, g32.split(sizes)))
    # output Linear: J = Wj x3 + bj
    out["jacobian_head.weight"].copy_(_tn(d_j, x[3]).reshape(-1))
    torch.sum(d_j, 0, out=out["jacobian_head.bias"])
    # query MLP: x0 = Wq [pe | bilinear(features)] + bq  (action_decoder_jacobian.py:421-427)
    d_q = _tn(dx0, pe)                                                       # [64, 64 slots]; slot 63 is the bias
    d_g = torch.zeros(feats_flat.shape[0], 64, dtype=torch.float32, device=dev)
    hip.scatter_footprint(dx0, foot_idx, foot_w, d_g, run_length=samples_per_ray)
    gq = out["jacobian_query_mlp.weight"].reshape(64, -1)
    gq[:, :63].index_copy_(1, _slot_to_channel(dev), d_q[:, :63])            # (every encoding channel has a slot)
    gq[:, 63:] = _tn(d_g, feats_flat)
    out["jacobian_query_mlp.bias"].copy_(d_q[:, 63])
    return tuple(out[n].to(t.dtype).reshape(t.shape) for n, t in zip(names, params))


class ActionFlowFunction(torch.autograd.Function):
    """optical_flow = f(Jacobian-head parameters); every other input is a constant captured by ``run``.
    ``names``: the parameters' names relative to the decoder, ``kind``: ``jacobian_mlp`` | ``jacobian_transformer`` | ``flow_mlp``
    (a ResnetFC that predicts the scene flow itself from cat[features, action], action_decoder_flow.py:156-176: the kernel
    contracts its three outputs with the constant 1 and the action is a per-image constant of its latent input)."""

    @staticmethod
    def forward(ctx, run: Callable[[], Dict[str, torch.Tensor]], project: Callable, action: torch.Tensor,
                features: torch.Tensor, names: Sequence[str], kind: str, *jparams: torch.Tensor):
        outs = run()
        ctx.outs = outs
        ctx.project = project
        ctx.action = action
        ctx.features = features
        ctx.names, ctx.kind = list(names), kind
        ctx.save_for_backward(*jparams)
        ctx.set_materialize_grads(False)
        return outs["flow"]

    @staticmethod
    def backward(ctx, g_flow):
        lead = (None,) * 6
        if g_flow is None:
            return lead + (None,) * len(ctx.saved_tensors)
        outs, action, features = ctx.outs, ctx.action, ctx.features
        weights = outs["weights"]  # [B,R,S]
        b, r, s = weights.shape
        a = action.shape[-1]
        # d flow2d / d (sum_s w (x + flow_s)): pinhole projection of the warped mean position (model.py:300-312)
        with torch.enable_grad():
            xw = outs["pos_warped"].detach().requires_grad_(True)
            (g_xw,) = torch.autograd.grad(ctx.project(xw), xw, g_flow.contiguous())
        feats_flat = _flat_features(features)
        if ctx.kind == "flow_mlp":
            # flow_s IS the head's output (action_decoder_flow.py:156-176)  =>  d flow[s,c] = w_s g_xw[c]
            d_j = torch.einsum("brs,brc->brsc", weights, g_xw).reshape(b * r * s, 3)
        else:
            # flow_s = sum_a J[a,:] act[a]  (action_decoder_jacobian.py:128-145)  =>  dJ[s,a,c] = w_s act[a] g_xw[c]
            d_j = torch.einsum("brs,ba,brc->brsac", weights, action, g_xw).reshape(b * r * s, 3 * a)
        if ctx.kind in ("jacobian_mlp", "flow_mlp"):   # a ResnetFC head: ``jacobian_head.*`` / ``jacobian_head_arm.*`` / ``flow_head*.*``
            cut = ctx.names[0].index(".") + 1
            p = {n[cut:]: t for n, t in zip(ctx.names, ctx.saved_tensors)}
            grads = resnetfc_backward(p, d_j, outs["jac_act"], outs["jac_pe"], outs["foot_idx"], outs["foot_w"], feats_flat,
                                      samples_per_ray=s, mask=outs.get("jac_mask"),
                                      latent_constants=action if ctx.kind == "flow_mlp" else None,
                                      forward_precision=outs.get("jac_forward_precision"))
            result = tuple(grads[n[cut:]] for n in ctx.names)
        elif outs.get("jac_act") is not None and os.environ.get("NJF_TRANSFORMER_BACKWARD", "hip") != "torch":
            # jacobian_transformer: the fused chain on the residual stream the forward dumped (round 6; the recomputation in
            # library ops below was 54 ms of a 62 ms action step on SURVEY's C4 shard and stays as the comparator of the tests)
            result = transformer_head_backward(ctx.names, ctx.saved_tensors, d_j, outs["jac_act"], outs["jac_pe"],
                                               outs["foot_idx"], outs["foot_w"], feats_flat, samples_per_ray=s,
                                               forward_precision=outs.get("jac_forward_precision"))
        else:  # jacobian_transformer: recompute the head on the dumped inputs, autograd to the original parameters
            pe = outs["jac_pe"]
            xyz_features = pe.new_empty(pe.shape[0], 63)
            xyz_features[:, _slot_to_channel(pe.device)] = pe[:, :63]
            idx, fw = outs["foot_idx"].long(), outs["foot_w"]
            pixel_features = sum(feats_flat[idx[:, c]] * fw[:, c:c + 1] for c in range(4))   # bilinear, border-clamped
            leaves = [t.detach().requires_grad_(True) for t in ctx.saved_tensors]
            with torch.enable_grad():
                jac = transformer_head(dict(zip(ctx.names, leaves)), xyz_features, pixel_features)
                result = torch.autograd.grad(jac, leaves, d_j)
        ctx.outs = None
        return lead + tuple(result)


COLOR_PARAM_ORDER: List[str] = [f"{i}.{wb}" for i in (0, 2, 4) for wb in ("weight", "bias")]


def color_head_backward(p: Dict[str, torch.Tensor], d_rgb: torch.Tensor, rgb: torch.Tensor, col_in: torch.Tensor,
                        col_act: torch.Tensor):
    """Backward of sigmoid(L4(relu(L2(relu(L0(cat[geo15, sh16])))))) (action_decoder_jacobian.py:315-322) from the
    dumped ``col_in`` [P,32] = [geo 15 | 1 | sh 16] and ``col_act`` [2,P,64].  Returns (grads, d_geo [P,15])."""
    grads: Dict[str, torch.Tensor] = {}
    d3 = d_rgb * rgb * (1.0 - rgb)
    grads["4.weight"] = _tn(d3, col_act[1])
    grads["4.bias"] = _colsum(d3)
    d2, d2_sum = hip.relu_backward(d3 @ p["4.weight"], col_act[1])
    grads["2.weight"] = _tn(d2, col_act[0])
    grads["2.bias"] = d2_sum
    d1, _ = hip.relu_backward(d2 @ p["2.weight"], col_act[0], want_colsum=False)
    d_w0 = _tn(d1, col_in)                     # [64, 32]: columns 0..14 geo, 15 the folded bias, 16..31 sh
    grads["0.weight"] = torch.cat([d_w0[:, :15], d_w0[:, 16:]], dim=1)
    grads["0.bias"] = d_w0[:, 15].clone()
    return grads, d1 @ p["0.weight"][:, :15]


def trunc_exp_backward_factor(sigma: torch.Tensor) -> torch.Tensor:
    """d density / d pre-activation of ``trunc_exp(x - 1)`` from the density itself (activations.py:24-29): the reference
    back-propagates g * exp(clamp(x - 1, -15, 15)); with sigma = exp(x - 1) that is sigma clamped to [e^-15, e^15]."""
    return sigma.clamp(min=math.exp(-15.0), max=math.exp(15.0))


class FieldFunction(torch.autograd.Function):
    """Perception-mode training (model_wrapper.py:117-146: every parameter trains, the losses read rgb, depth and the
    per-level weights).  Outputs the per-sample fields the compositing consumes -- final density [B,R,S,1], colour
    [B,R,S,3] and each proposal level's density [B,R,S_l,1] -- as functions of (encoder features, density head, colour
    head, proposal nets).  Sample placement is a constant: the reference detaches the resampled bins
    (ray_samplers.py:446).  Forward = the fused HIP kernels with activation dumps; backward = library GEMMs on the
    dumps; the compositing between these fields and the losses is left to autograd (it is O(B R S) elementwise work)."""

    @staticmethod
    def forward(ctx, run: Callable[[], Dict[str, torch.Tensor]], n_levels: int, n_prop: int, *tensors):
        """``tensors`` = the encoder output -- ONE [B,512,Hf,Wf] feature tensor (n_levels = 1) or the encoder's un-concatenated
        latents (n_levels = 4: the forward pass hoists from them, njf_project_pyramid; the 512-channel matrix and the
        gradients of the latents are formed in the backward pass by njf_upsample_concat and its adjoint, round 3: this
        replaces ATen's three upsample_bilinear2d forward + backward launches per step) -- followed by the parameters."""
        outs = run()
        ctx.outs = outs
        ctx.n_levels = n_levels
        ctx.n_prop = n_prop
        ctx.save_for_backward(*tensors)
        ctx.set_materialize_grads(False)
        b, r, s = outs["weights"].shape
        # detached aliases: the returned tensors must not be the objects ``ctx.outs`` holds (reference cycle)
        fields = [outs["density"].detach().reshape(b, r, s, 1), outs["color"].detach().reshape(b, r, s, 3)]
        fields += [d["density"].detach()[..., None] for d in outs["proposal_dumps"]]
        return tuple(fields)

    @staticmethod
    def backward(ctx, g_sigma, g_color, *g_prop):
        levels, params = ctx.saved_tensors[:ctx.n_levels], ctx.saved_tensors[ctx.n_levels:]
        n = len(JACOBIAN_PARAM_ORDER)
        outs = ctx.outs
        features = levels[0] if ctx.n_levels == 1 else FeaturePyramid([lv.detach() for lv in levels])
        feats_flat = _flat_features(features).detach()
        d_feats = torch.zeros_like(feats_flat) if any(ctx.needs_input_grad[3:3 + ctx.n_levels]) else None
        out_grads = [None] * len(params)

        clamp_exp = trunc_exp_backward_factor

        if g_sigma is not None or g_color is not None:
            pts = outs["density"].numel()
            den = dict(zip(JACOBIAN_PARAM_ORDER, params[:n]))
            col = dict(zip(COLOR_PARAM_ORDER, params[n:n + 6]))
            d_out = torch.zeros(pts, 16, dtype=torch.float32, device=feats_flat.device)
            if g_color is not None:
                cgrads, d_geo = color_head_backward(col, g_color.reshape(pts, 3), outs["color"].reshape(pts, 3),
                                                    outs["col_in"], outs["col_act"])
                d_out[:, :15] = d_geo
                for i, k in enumerate(COLOR_PARAM_ORDER):
                    out_grads[n + i] = cgrads[k]
            if g_sigma is not None:
                d_out[:, 15] = g_sigma.reshape(pts) * clamp_exp(outs["density"].reshape(pts))
            grads = resnetfc_backward(den, d_out, outs["den_act"], outs["jac_pe"], outs["foot_idx"], outs["foot_w"],
                                      feats_flat, d_feats, samples_per_ray=outs["weights"].shape[-1], mask=outs.get("den_mask"),
                                      forward_precision=outs.get("den_forward_precision"))
            for i, k in enumerate(JACOBIAN_PARAM_ORDER):
                out_grads[i] = grads[k]
        for lvl in range(ctx.n_prop):
            g = g_prop[lvl]
            if g is None:
                continue
            d = outs["proposal_dumps"][lvl]
            net = dict(zip(JACOBIAN_PARAM_ORDER, params[n + 6 + lvl * n:n + 6 + (lvl + 1) * n]))
            d_out = (g.reshape(-1) * clamp_exp(d["density"].reshape(-1)))[:, None]
            grads = resnetfc_backward(net, d_out, d["act"], d["pe"], d["foot_idx"], d["foot_w"], feats_flat, d_feats,
                                      samples_per_ray=d["density"].shape[-1], mask=d.get("mask"),
                                      forward_precision=d.get("forward_precision"))
            for i, k in enumerate(JACOBIAN_PARAM_ORDER):
                out_grads[n + 6 + lvl * n + i] = grads[k]
        ctx.outs = None
        g_levels = [None] * ctx.n_levels
        if d_feats is not None:
            if ctx.n_levels == 1:
                bsz, c, hf, wf = features.shape
                g_levels = [d_feats.reshape(bsz, hf, wf, c).permute(0, 3, 1, 2)]
            else:   # adjoint of the encoder tail: channels-last [T,512] -> one NCHW gradient per latent
                g_levels = hip.upsample_concat_backward(d_feats, [tuple(lv.shape) for lv in levels])
        return (None, None, None) + tuple(g_levels) + tuple(out_grads)


class CompositeFunction(torch.autograd.Function):
    """Alpha compositing of one sampling level as an autograd node (perception-mode training): weights = get_weights(sigma)
    (ray_samplers.py:77-101), rgb = sum_s w c (model.py:257-270), depth = sum_s w t / (sum_s w + 1e-10) BEFORE the clip
    (model.py:271-276).  The VALUES are the ones the fused forward kernels already composited (``values``: a dict holding
    ``weights`` [B,R,S] and, for the final level, ``rgb`` [B,R,3] / ``depth`` [B,R,1]); the backward pass is one HIP launch,
    njf_composite_backward -- autograd's ~25 element-wise / scan launches per level otherwise.  ``steps`` / ``color`` are
    None for a proposal level (only its weights are read by the losses)."""

    @staticmethod
    def forward(ctx, deltas, steps, sigma, color, values):
        ctx.save_for_backward(deltas, sigma, *([] if steps is None else [steps]), *([] if color is None else [color]))
        ctx.has = (steps is not None, color is not None)
        ctx.set_materialize_grads(False)
        w = values["weights"].detach().reshape(sigma.shape)
        if color is None:
            return w
        return w, values["rgb"].detach().view_as(values["rgb"]), values["depth"].detach().view_as(values["depth"])

    @staticmethod
    def backward(ctx, g_w, g_rgb=None, g_depth=None):
        saved = list(ctx.saved_tensors)
        deltas, sigma = saved[0], saved[1]
        steps = saved[2] if ctx.has[0] else None
        color = saved[-1] if ctx.has[1] else None
        if g_w is None and g_rgb is None and g_depth is None:
            return None, None, None, None, None
        n_s = sigma.shape[-2]                                          # per-sample fields are [..., S, 1] / [..., S, 3]
        per_sample = lambda t: None if t is None else t.reshape(-1, n_s).contiguous()
        want_color = color is not None and ctx.needs_input_grad[3]
        g_sigma, g_color = hip.composite_backward(
            per_sample(deltas), per_sample(steps), per_sample(sigma),
            None if color is None else color.reshape(-1, n_s, 3).contiguous(), per_sample(g_w),
            None if g_rgb is None else g_rgb.reshape(-1, 3).contiguous(),
            None if (g_depth is None or steps is None) else g_depth.reshape(-1).contiguous(), want_color=want_color)
        return None, None, g_sigma.reshape(sigma.shape), (g_color.reshape(color.shape) if g_color is not None else None), None


def perception_params(model) -> List[torch.Tensor]:
    """Parameters FieldFunction differentiates, in its order: density head | colour head | proposal nets."""
    den = dict(model.decoder.density_head.named_parameters())
    col = dict(model.decoder.color_head.named_parameters())
    out = [den[k] for k in JACOBIAN_PARAM_ORDER] + [col[k] for k in COLOR_PARAM_ORDER]
    for net in model.proposal_networks:
        head = dict(net.density_head.named_parameters())
        out += [head[k] for k in JACOBIAN_PARAM_ORDER]
    return out


class RefuseBackward(torch.autograd.Function):
    """Identity whose backward raises: attached to the outputs of a forward pass whose trainable set the fused path
    cannot differentiate, so inference-style calls work and any attempt to back-propagate fails loudly."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, anchor: torch.Tensor, message: str):
        ctx.message = message
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError(ctx.message)


def is_action_mode(model) -> bool:
    """The reference's action-mode trainable set (ModelWrapper.freeze_parameters, model_wrapper.py:75-85 with
    ActionDecoder*.freeze_non_action_parameters): only decoder parameters whose name contains the decoder's action pattern
    ("jacobian" for the Jacobian decoders, "flow_head" for flow_mlp, action_decoder_flow.py:67, :281-288)."""
    names = trainable_names(model)
    pattern = "decoder." + model.decoder.action_param_glob_pattern      # "jacobian_head" | "jacobian" | "flow_head"
    return bool(names) and all(n.startswith(pattern) for n in names)


PERCEPTION_MESSAGE = ("optical_flow is differentiable only in the reference's action mode (ModelWrapper.freeze_parameters: "
                      "just the jacobian_mlp head trainable); with any other trainable set the fused path differentiates "
                      "rgb, depth and the per-level weights (perception mode) and optical_flow is a value only")


def trainable_names(module: torch.nn.Module) -> List[str]:
    return [n for n, q in module.named_parameters() if q.requires_grad]


def action_params(model):
    """(names relative to the decoder, tensors) of the Jacobian head in the order ActionFlowFunction uses: the
    ResnetFC layer order for ``jacobian_mlp``, registration order for ``jacobian_transformer`` (index embedding, query
    MLP, attention decoder, output Linear).  Frozen members are included (they simply receive unused gradients)."""
    dec = dict(model.decoder.named_parameters())
    prefix = model.decoder.active_head_prefix     # "jacobian_head." | "jacobian_head_arm." | "flow_head." | "flow_head_arm." | ""
    if prefix:
        names = [prefix + k for k in JACOBIAN_PARAM_ORDER]
    else:
        names = [n for n in dec if n.startswith("jacobian") and not n.startswith("jacobian_head_arm.")]
    return names, [dec[n] for n in names]


def action_kind(model) -> str:
    """Which backward ActionFlowFunction runs: the ResnetFC chain for a ResnetFC head (jacobian_mlp, and the arm head of either
    decoder), the recomputed transformer head otherwise."""
    if model.cfg.action_decoder.name == "flow_mlp":
        return "flow_mlp"
    return "jacobian_mlp" if model.decoder.active_head_prefix else model.cfg.action_decoder.name
