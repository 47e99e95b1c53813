"""Inverse dynamics on a rendered Jacobian field: find the robot command that produces a desired optical flow.

Reference: notebooks/real_world/2_inverse_dynamics.ipynb (cells 26-29) encodes the image once
(``Model.encode_image``) and then runs 100 Adam steps through ``Model.infer_optical_flow``; the authors note the
loop becomes real-time "if a least square solver is used" (1_visualize_jacobian_fields.ipynb:448).  Because the scene
flow is linear in the command (``flow_s = J_s a``), the composited warped point is ``x_bar + M a`` with
``M = sum_s w_s J_s`` -- exactly the ``action_features`` the fused render kernel already composites -- so one fused
render of the tracked rays yields everything a Gauss-Newton / least-squares solve needs; the per-iteration work is a
[2R x A] least-squares problem.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .model import CameraInput, Model, RenderingInput, RobotInput


@dataclass
class FlowLinearization:
    """Per-ray composited quantities of one fused render: optical_flow(a) = proj(x_bar + M a) - proj(x_bar)."""
    mean_position: torch.Tensor  # [B,R,3]  sum_s w_s x_s
    jacobian: torch.Tensor       # [B,R,3,A] sum_s w_s J_s, spatial-major (J viewed (action, spatial) in the reference)
    trgt_extrinsics: torch.Tensor
    trgt_intrinsics: torch.Tensor

    def optical_flow(self, action: torch.Tensor) -> torch.Tensor:
        warped = self.mean_position + torch.einsum("brca,ba->brc", self.jacobian, action)
        return (Model._project(warped, self.trgt_extrinsics, self.trgt_intrinsics)
                - Model._project(self.mean_position, self.trgt_extrinsics, self.trgt_intrinsics))


@torch.no_grad()
def linearize_flow(model: Model, camera_input: CameraInput, rendering_input: RenderingInput,
                   action_dim: Optional[int] = None) -> FlowLinearization:
    """One fused render (any command: the composited Jacobian does not depend on it)."""
    a = action_dim or model.cfg.action_dim
    b = rendering_input.origins.shape[0]
    zero = torch.zeros(b, a, dtype=torch.float32, device=rendering_input.origins.device)
    was_training = model.training
    model.eval()
    try:
        out = model._forward_inference(camera_input, rendering_input, RobotInput(zero), compute_vis_features=True)
    finally:
        model.train(was_training)
    feat = out.vis_output.action_features  # [B,R,3A], (action, spatial) order
    jac = feat.reshape(*feat.shape[:2], a, 3).transpose(-1, -2).contiguous()
    return FlowLinearization(out.vis_output.ray_positions, jac, camera_input.trgt_extrinsics, camera_input.trgt_intrinsics)


def _projection_matrix(lin: FlowLinearization) -> torch.Tensor:
    """[B,3,4] world -> homogeneous pixel matrix K . inv(E)[:3]."""
    from . import hip
    return lin.trgt_intrinsics @ hip.inverse(lin.trgt_extrinsics)[:, :3, :]


@torch.no_grad()
def solve_action(lin: FlowLinearization, target_flow: torch.Tensor, init_action: Optional[torch.Tensor] = None,
                 iterations: int = 20, damping: float = 1e-3, visible_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Levenberg-Marquardt on ``|| optical_flow(a) - target_flow ||^2`` (pixels): all iterations in ONE HIP launch
    (``njf_solve_action``: one workgroup per batch element, deterministic reductions, no host synchronisation).

    target_flow [B,R,2], visible_mask [B,R] (the notebook masks the loss with the tracker's visibility) -> [B,A].
    The only non-linearity is the perspective divide, so for the few-pixel flows of a control step a handful of
    iterations reach the minimum; a step is kept only where it lowers the cost, which keeps large-flow problems
    stable.  GPU tensors only (no CPU path); the tensor-op restatement used by the tests lives in oracle/."""
    from . import hip
    b = target_flow.shape[0]
    out = torch.empty(b, lin.jacobian.shape[-1], dtype=torch.float32, device=target_flow.device)
    f = lambda t: None if t is None else t.float().contiguous()
    hip.solve_action(f(lin.mean_position), f(lin.jacobian), f(_projection_matrix(lin)), f(target_flow), f(visible_mask),
                     f(init_action), iterations, damping, out)
    return out


class GraphedLinearizer:
    """``linearize_flow`` for a fixed camera rig and ray set, captured once into a HIP graph and replayed per frame:
    at control-loop sizes (a few hundred tracked rays) the fused render is launch-bound, so replaying one graph
    (encoder + lin_z hoist + proposal pass + final pass) removes the per-launch host cost.  ``__call__(image)`` copies
    the new context image into the captured input buffer and replays; the returned tensors are the graph's static
    outputs (overwritten by the next call)."""

    def __init__(self, model: Model, camera_input: CameraInput, rendering_input: RenderingInput,
                 action_dim: Optional[int] = None, warmup: int = 2):
        self._image = camera_input.input_image.clone()
        cam = CameraInput(self._image, camera_input.ctxt_extrinsics, camera_input.ctxt_intrinsics,
                          camera_input.trgt_extrinsics, camera_input.trgt_intrinsics)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up off the default stream: packs weights, loads MIOpen kernels
            for _ in range(warmup):
                linearize_flow(model, cam, rendering_input, action_dim)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._out = linearize_flow(model, cam, rendering_input, action_dim)

    def __call__(self, image: torch.Tensor) -> FlowLinearization:
        self._image.copy_(image)
        self._graph.replay()
        return self._out


class GraphedInverseDynamics:
    """The whole control step -- encoder, lin_z hoist, proposal pass, final pass, ``iterations`` Levenberg-Marquardt
    steps -- as ONE replayed HIP graph: ``action = controller(image, target_flow[, init_action, visible_mask])``.
    Camera rig, tracked rays and iteration count are fixed at capture time; nothing in the step synchronises with the
    host, so the per-frame host cost is three small copies and one graph launch."""

    def __init__(self, model: Model, camera_input: CameraInput, rendering_input: RenderingInput, iterations: int = 8,
                 damping: float = 1e-3, action_dim: Optional[int] = None, warmup: int = 2):
        a = action_dim or model.cfg.action_dim
        b, r = rendering_input.origins.shape[:2]
        dev = rendering_input.origins.device
        self._image = camera_input.input_image.clone()
        self._target = torch.zeros(b, r, 2, device=dev)
        self._init = torch.zeros(b, a, device=dev)
        self._mask = torch.ones(b, r, device=dev)
        cam = CameraInput(self._image, camera_input.ctxt_extrinsics, camera_input.ctxt_intrinsics,
                          camera_input.trgt_extrinsics, camera_input.trgt_intrinsics)

        def step():
            lin = linearize_flow(model, cam, rendering_input, a)
            return solve_action(lin, self._target, self._init, iterations, damping, self._mask)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._action = step()

    def __call__(self, image: torch.Tensor, target_flow: torch.Tensor, init_action: Optional[torch.Tensor] = None,
                 visible_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._image.copy_(image)
        self._target.copy_(target_flow)
        if init_action is None:
            self._init.zero_()
        else:
            self._init.copy_(init_action)
        if visible_mask is None:
            self._mask.fill_(1.0)
        else:
            self._mask.copy_(visible_mask)
        self._graph.replay()
        return self._action
