"""Tensor-op restatement of the product's Levenberg-Marquardt inverse-dynamics solve (njf_solve_action) -- TEST
INFRASTRUCTURE, like the rest of oracle/: only tests/ may import it.  The reference has no such solver (its notebook
runs Adam through Model.infer_optical_flow, restated in njf_oracle.infer_optical_flow); this file is the checker of
the HIP kernel's arithmetic: same algorithm, batched torch ops, any device."""

from typing import Optional

import torch


def _solve_spd(h: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """x = h^-1 g for small damped normal matrices h [B,A,A] (symmetric positive definite), g [B,A]: un-pivoted
    Gauss-Jordan in batched tensor ops.  torch.linalg.solve would do, but its LAPACK-style back ends synchronise
    with the host (error check / MAGMA), which rules out HIP-graph capture of the control step."""
    a = h.shape[-1]
    m = torch.cat([h, g[..., None]], dim=-1)                                   # [B, A, A+1]
    rows = torch.arange(a, device=h.device)
    for k in range(a):
        pivot_row = m[:, k:k + 1, :] / m[:, k:k + 1, k:k + 1]
        factor = torch.where((rows == k)[None, :, None], torch.zeros_like(m[:, :, k:k + 1]), m[:, :, k:k + 1])
        m = torch.where((rows == k)[None, :, None], pivot_row, m - factor * pivot_row)
    return m[..., -1]



@torch.no_grad()
def lm_solve_action(lin, target_flow: torch.Tensor, init_action: Optional[torch.Tensor] = None,
                           iterations: int = 20, damping: float = 1e-3,
                           visible_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Levenberg-Marquardt on ``|| optical_flow(a) - target_flow ||^2`` (pixels), as batched tensor ops.

    target_flow [B,R,2], visible_mask [B,R] (the notebook masks the loss with the tracker's visibility) -> [B,A].
    The only non-linearity is the perspective divide, so for the few-pixel flows of a control step a handful of
    iterations reach the minimum; each is a [2R x A] normal-equation solve per batch element.  A step is kept only
    where it lowers the cost (per batch element, no host synchronisation), which keeps large-flow problems stable."""
    b, r = target_flow.shape[:2]
    a_dim = lin.jacobian.shape[-1]
    dev = target_flow.device
    action = torch.zeros(b, a_dim, dtype=torch.float32, device=dev) if init_action is None else init_action.clone().float()
    w = torch.ones(b, r, 1, device=dev) if visible_mask is None else visible_mask[..., None].float()
    proj = (lin.trgt_intrinsics @ torch.linalg.inv(lin.trgt_extrinsics)[:, :3, :])[:, None]                                   # [B,1,3,4]
    p_lin, p_off = proj[..., :3].expand(b, r, 3, 3), proj[..., 3]
    uv0 = (torch.einsum("brij,brj->bri", p_lin, lin.mean_position) + p_off)
    uv0 = uv0[..., :2] / (uv0[..., 2:] + 1e-9)

    def evaluate(act):
        x = lin.mean_position + torch.einsum("brca,ba->brc", lin.jacobian, act)
        xyw = torch.einsum("brij,brj->bri", p_lin, x) + p_off
        depth = xyw[..., 2:] + 1e-9
        uv = xyw[..., :2] / depth
        res = ((uv - uv0) - target_flow) * w                                   # [B,R,2]
        return uv, depth, res, res.square().sum((1, 2))

    lam = torch.full((b, 1, 1), damping, device=dev)
    uv, depth, res, cost = evaluate(action)
    for _ in range(iterations):
        duv_dx = (proj[..., :2, :3] - uv[..., None] * proj[..., 2:3, :3]) / depth[..., None]     # [B,R,2,3]
        jac = ((duv_dx @ lin.jacobian) * w[..., None]).reshape(b, 2 * r, a_dim)
        h = jac.transpose(1, 2) @ jac
        diag = torch.diag_embed(torch.diagonal(h, dim1=1, dim2=2).clamp_min(1e-12))
        step = _solve_spd(h + lam * diag, (jac.transpose(1, 2) @ res.reshape(b, 2 * r, 1))[..., 0])
        cand = action - step
        uv_c, depth_c, res_c, cost_c = evaluate(cand)
        better = cost_c < cost                                                # NaN (point behind the camera) -> rejected
        sel = better[:, None]
        action = torch.where(sel, cand, action)
        uv, depth, res = (torch.where(sel[..., None], n, o) for n, o in ((uv_c, uv), (depth_c, depth), (res_c, res)))
        cost = torch.where(better, cost_c, cost)
        lam = torch.where(better[:, None, None], lam / 3.0, lam * 4.0).clamp(1e-9, 1e9)
    return action


