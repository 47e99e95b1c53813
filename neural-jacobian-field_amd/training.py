"""Action-mode training support: gradients of the rendered optical flow w.r.t. the Jacobian head.

Reference contract: in ``dataset.mode == "action"`` everything except the Jacobian head is frozen
(``models/model_wrapper.py:75-85``) and the loss is ``0.01 * mse(optical_flow, target)`` (``:148-160``), so density,
weights and sample placement are constants of the backward pass (build step 6).

Forward: the fused HIP kernels, with the final pass additionally dumping the ReLU'd input of every layer of the
Jacobian ``ResnetFC`` (``njf_render_forward`` with ``jac_act``/``jac_pe``/``foot_*`` outputs).
Backward (round-1 form): the layer-by-layer chain on the dumped ``[P,128]`` matrices as plain library GEMMs
(rocBLAS through ``torch.matmul``) plus ReLU masks -- exact, tested against autograd of the CPU oracle.  Fusing this
chain into a HIP kernel is the next step of SURVEY.md section 8f #2; ``jacobian_transformer`` and perception-mode
(full-model) gradients are not implemented and raise.
"""

from __future__ import annotations

from typing import Callable, Dict, List, Sequence

import torch

JACOBIAN_PARAM_ORDER: List[str] = (
    ["lin_in.weight", "lin_in.bias"]
    + [f"blocks.{b}.{fc}.{wb}" for b in range(5) for fc in ("fc_0", "fc_1") for wb in ("weight", "bias")]
    + [f"lin_z.{i}.{wb}" for i in range(3) for wb in ("weight", "bias")]
    + ["lin_out.weight", "lin_out.bias"]
)

# positional-encoding slot -> reference channel (csrc/njf_kernels.hip::pack_source, kind 1); slot 63 is the bias
_PE_SLOT_TO_CHANNEL = list(range(30)) + [60, 61] + list(range(30, 60)) + [62]


class ActionFlowFunction(torch.autograd.Function):
    """optical_flow = f(jacobian_head parameters); every other input is a constant captured by ``run``."""

    @staticmethod
    def forward(ctx, run: Callable[[], Dict[str, torch.Tensor]], project: Callable, action: torch.Tensor,
                features: torch.Tensor, *jparams: torch.Tensor):
        outs = run()
        ctx.outs = outs
        ctx.project = project
        ctx.action = action
        ctx.features = features
        ctx.save_for_backward(*jparams)
        ctx.set_materialize_grads(False)
        return outs["flow"]

    @staticmethod
    def backward(ctx, g_flow):
        if g_flow is None:
            return (None,) * (4 + len(ctx.saved_tensors))
        p = dict(zip(JACOBIAN_PARAM_ORDER, ctx.saved_tensors))
        outs, action, features = ctx.outs, ctx.action, ctx.features
        weights = outs["weights"]  # [B,R,S]
        b, r, s = weights.shape
        a = action.shape[-1]
        # d flow2d / d (sum_s w (x + flow_s)): pinhole projection of the warped mean position (model.py:300-312)
        with torch.enable_grad():
            xw = outs["pos_warped"].detach().requires_grad_(True)
            (g_xw,) = torch.autograd.grad(ctx.project(xw), xw, g_flow.contiguous())
        # flow_s = sum_a J[a,:] act[a]  (action_decoder_jacobian.py:128-145)  =>  dJ[s,a,c] = w_s act[a] g_xw[c]
        d_j = torch.einsum("brs,ba,brc->brsac", weights, action, g_xw).reshape(b * r * s, 3 * a)
        act = outs["jac_act"]  # [11, P, 128]
        grads: Dict[str, torch.Tensor] = {}
        r_out = act[10]
        grads["lin_out.weight"] = d_j.t() @ r_out
        grads["lin_out.bias"] = d_j.sum(0)
        delta = (d_j @ p["lin_out.weight"]) * (r_out > 0)
        feats_flat = None
        for blk in range(4, -1, -1):
            r0, r1 = act[2 * blk], act[2 * blk + 1]
            grads[f"blocks.{blk}.fc_1.weight"] = delta.t() @ r1
            grads[f"blocks.{blk}.fc_1.bias"] = delta.sum(0)
            d_net = (delta @ p[f"blocks.{blk}.fc_1.weight"]) * (r1 > 0)
            grads[f"blocks.{blk}.fc_0.weight"] = d_net.t() @ r0
            grads[f"blocks.{blk}.fc_0.bias"] = d_net.sum(0)
            delta = delta + (d_net @ p[f"blocks.{blk}.fc_0.weight"]) * (r0 > 0)
            if blk < 3:  # lin_z[blk](bilinear(F)) was added here: scatter the gradient onto the texels, then one GEMM
                if feats_flat is None:
                    feats_flat = features.permute(0, 2, 3, 1).reshape(-1, features.shape[1])
                    idx = outs["foot_idx"].long()
                    fw = outs["foot_w"]
                d_g = torch.zeros(feats_flat.shape[0], delta.shape[1], dtype=delta.dtype, device=delta.device)
                for c in range(4):
                    d_g.index_add_(0, idx[:, c], delta * fw[:, c:c + 1])
                grads[f"lin_z.{blk}.weight"] = d_g.t() @ feats_flat
                grads[f"lin_z.{blk}.bias"] = delta.sum(0)
        d_in = delta.t() @ outs["jac_pe"]  # [128, 64] in slot order
        grads["lin_in.weight"] = d_in[:, :63].new_zeros(d_in.shape[0], 63).index_copy_(
            1, torch.tensor(_PE_SLOT_TO_CHANNEL, device=d_in.device), d_in[:, :63])
        grads["lin_in.bias"] = d_in[:, 63].clone()
        ctx.outs = None
        return (None, None, None, None) + tuple(grads[k] for k in JACOBIAN_PARAM_ORDER)


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
    names = trainable_names(model)
    return (bool(names) and all(n.startswith("decoder.jacobian_head.") for n in names)
            and model.cfg.action_decoder.name == "jacobian_mlp")


PERCEPTION_MESSAGE = ("the fused HIP path differentiates only the Jacobian head of a jacobian_mlp decoder (reference action "
                      "mode, ModelWrapper.freeze_parameters); gradients w.r.t. other parameters (perception mode, "
                      "jacobian_transformer) are not implemented -- SURVEY.md section 8f #2")


def trainable_names(module: torch.nn.Module) -> List[str]:
    return [n for n, q in module.named_parameters() if q.requires_grad]


def check_action_mode(model) -> Sequence[torch.Tensor]:
    """Returns the Jacobian-head parameters in JACOBIAN_PARAM_ORDER, or raises if the trainable set is not the
    reference's action mode (models/model_wrapper.py:75-85) on a ``jacobian_mlp`` decoder."""
    names = trainable_names(model)
    if any(not n.startswith("decoder.jacobian_head.") for n in names):
        other = [n for n in names if not n.startswith("decoder.jacobian_head.")][:3]
        raise NotImplementedError(
            "the fused path differentiates only the Jacobian head (reference action mode: "
            f"ModelWrapper.freeze_parameters); also trainable here: {other} ... (perception-mode backward is "
            "SURVEY.md section 8f #2)")
    if model.cfg.action_decoder.name != "jacobian_mlp":
        raise NotImplementedError("backward is implemented for the jacobian_mlp decoder only")
    head = dict(model.decoder.jacobian_head.named_parameters())
    return [head[k] for k in JACOBIAN_PARAM_ORDER]
