"""ctypes binding of ``libnjf_hip.so`` (C ABI in ``include/njf_hip.h``).

PyTorch is plumbing here: it owns device memory and the stream; every compute call goes through
the C ABI.  There is **no CPU fallback**: if the library is missing or a tensor is not a
contiguous fp32 device tensor the call raises.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NJF_HIP_LIB", os.path.join(_HERE, "libnjf_hip.so"))  # override: kernel A/B experiments only

MAX_ACTION_DIM = 10
ZDIM = 384
RESNET_W_FLOATS = 22 * 8192
RESNET_B_FLOATS = 10 * 128 + 32
COLOR_W_FLOATS = 8192
COLOR_B_FLOATS = 96
TRANSFORMER_W_FLOATS = 7 * 8192
TRANSFORMER_B_FLOATS = 3 * 256 + 32
QDIM = 64
JACOBIAN_NONE, JACOBIAN_MLP, JACOBIAN_TRANSFORMER = 0, 1, 2
# MFMA precision of the fused MLPs (include/njf_hip.h: NJF_PRECISION_*).  "f16x2" = fp32 operands split into two
# fp16 (hi+lo), three f16 MFMAs per product block, fp32 accumulation: fp32-class accuracy, ~5x less matrix time.
# "f16f6" = the same hi*hi product, the two 2^-11-sized correction products in block-scaled fp6 (4x the f16 MFMA rate).
# "f16" = PLAIN fp16 products (round 5; BASELINE config 5's "fp16 MFMA fused-MLP"): weights and layer inputs rounded to fp16, fp32
# accumulation, hoisted maps stored in fp16 -- a reduced-precision inference mode with its own stated tolerance (DESIGN.md
# section 5), selectable, never the default.
PRECISIONS = {"f32": 0, "f16x2": 1, "f16f6": 2, "f16": 3}
REDUCED_PRECISIONS = ("f16",)   # modes that are NOT held to the fp32 parity bound
# Default: "f16f6" for the final pass with the proposal networks on "f16x2" (Model.set_precision's policy).  It passes the
# same parity suite as the exact-fp32 path with the same bounds (profiles/r02_parity_margins.json).
DEFAULT_PRECISION = os.environ.get("NJF_PRECISION", "f16f6")
if DEFAULT_PRECISION not in PRECISIONS or DEFAULT_PRECISION in REDUCED_PRECISIONS:
    # the environment may pick among the modes that meet the fp32 parity bound; a reduced mode is chosen per model
    # (Model.set_precision("f16")), by code that knows its tolerance -- never by a variable that every process inherits
    raise ValueError(f"NJF_PRECISION={DEFAULT_PRECISION!r}: the package default must be one of "
                     f"{sorted(set(PRECISIONS) - set(REDUCED_PRECISIONS))}")


def proposal_precision_for(precision: str) -> str:
    """The proposal networks' precision that goes with a decoder precision: the same, except that "f16f6" keeps sample
    PLACEMENT on "f16x2" (an error of 1e-5 in the proposal weights moves samples enough to show up as 3e-4 in depth / flow
    through the positional encoding's 2*pi*512 gain; inside the final pass the same error stays 1e-5).  Re-measured in
    round 2 with the proposal pass forced to "f16f6": depth 3.0e-4 against a bound of 1.0e-4, optical flow 7.8e-4 against
    3.6e-4 (parity cases 1 and 4) for a proposal pass of 3.16 instead of 3.80 ms -- not adopted."""
    return "f16x2" if precision == "f16f6" else precision


def precision_code(precision: Optional[str], jacobian_precision: Optional[str] = None) -> int:
    """include/njf_hip.h: NJF_PRECISION_* (and NJF_PRECISION_MIXED when the Jacobian head runs in another precision)."""
    name = DEFAULT_PRECISION if precision is None else precision
    if name not in PRECISIONS:
        raise ValueError(f"njf_hip: unknown precision {name!r}; choose from {sorted(PRECISIONS)}")
    code = PRECISIONS[name]
    if jacobian_precision is not None and jacobian_precision != name:
        if jacobian_precision not in PRECISIONS:
            raise ValueError(f"njf_hip: unknown precision {jacobian_precision!r}; choose from {sorted(PRECISIONS)}")
        if "f32" in (name, jacobian_precision) or "f16" in (name, jacobian_precision):
            raise ValueError("njf_hip: mixed decoder precisions exist for the two split-precision modes only")
        code |= (PRECISIONS[jacobian_precision] + 1) << 4
    return code

_vp = C.c_void_p


class ResnetFcWeights(C.Structure):
    _fields_ = [
        ("lin_in_w", _vp), ("lin_in_b", _vp),
        ("fc0_w", _vp * 5), ("fc0_b", _vp * 5), ("fc1_w", _vp * 5), ("fc1_b", _vp * 5),
        ("lin_z_w", _vp * 3), ("lin_z_b", _vp * 3),
        ("lin_out_w", _vp), ("lin_out_b", _vp), ("d_out", C.c_int),
    ]


class ColorHeadWeights(C.Structure):
    _fields_ = [("w0", _vp), ("b0", _vp), ("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp)]


class Cameras(C.Structure):
    _fields_ = [
        ("ctxt_w2c", _vp), ("ctxt_k", _vp), ("trgt_w2c", _vp), ("trgt_k", _vp),
        ("z_near", _vp), ("z_far", _vp), ("action", _vp), ("batch", C.c_int), ("action_dim", C.c_int),
    ]


class FeatureMap(C.Structure):
    _fields_ = [("data", _vp), ("height", C.c_int), ("width", C.c_int), ("stride", C.c_int)]


class RenderOutputs(C.Structure):
    _fields_ = [(n, _vp) for n in (
        "rgb", "depth", "step_minmax", "flow", "pos", "pos_warped", "action_features",
        "weights", "density", "color", "sample_flow", "jacobian", "jac_act", "jac_pe", "foot_idx", "foot_w",
        "den_act", "col_in", "col_act", "frame_partials", "trgt_rgb", "trgt_flow", "jac_mask", "den_mask")] + [("dump_f16", C.c_int)]


class PyramidLevel(C.Structure):
    _fields_ = [("feats", _vp), ("channels", C.c_int), ("height", C.c_int), ("width", C.c_int)]


class ActivationDump(C.Structure):
    _fields_ = [("act", _vp), ("pe", _vp), ("foot_idx", _vp), ("foot_w", _vp), ("mask", _vp), ("act_f16", C.c_int)]


_lib = None

_SIGNATURES = {
    "njf_abi_version": ([], C.c_int),
    "njf_rays_per_workgroup": ([], C.c_int),
    "njf_error_string": ([C.c_int], C.c_char_p),
    "njf_pack_resnetfc": ([C.POINTER(ResnetFcWeights), _vp, _vp, _vp, _vp, C.c_int, _vp], C.c_int),
    "njf_pack_resnetfc_ld": ([C.POINTER(ResnetFcWeights), _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp], C.c_int),
    "njf_pack_color_head": ([C.POINTER(ColorHeadWeights), _vp, _vp, C.c_int, _vp], C.c_int),
    "njf_pack_linear": ([_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp], C.c_int),
    "njf_project_features": ([_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp], C.c_int),
    "njf_project_features_ld": ([_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp], C.c_int),
    "njf_project_pyramid": ([C.POINTER(PyramidLevel), C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp],
                            C.c_int),
    "njf_hoisted_channel": ([C.c_int, C.c_int, C.c_int], C.c_int),
    "njf_upsample_concat": ([C.POINTER(PyramidLevel), C.c_int, C.c_int, _vp, _vp], C.c_int),
    "njf_upsample_concat_backward": ([_vp, C.POINTER(PyramidLevel), C.c_int, C.c_int, _vp], C.c_int),
    "njf_solve_action": ([_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp, _vp], C.c_int),
    "njf_invert_4x4": ([_vp, C.c_int, _vp, _vp], C.c_int),
    "njf_generate_rays": ([_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp], C.c_int),
    "njf_proposal_forward": ([_vp, _vp, C.c_int, C.POINTER(Cameras), C.POINTER(FeatureMap), C.c_int, _vp, _vp,
                              _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp, C.POINTER(ActivationDump), C.c_int, _vp],
                             C.c_int),
    "njf_render_forward": ([_vp, _vp, C.c_int, C.POINTER(Cameras), C.POINTER(FeatureMap), C.c_int, C.c_int, C.c_int,
                            _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.POINTER(RenderOutputs), C.c_int, _vp], C.c_int),
    "njf_points_forward": ([_vp, _vp, C.c_int, C.POINTER(Cameras), C.POINTER(FeatureMap), C.c_int, C.c_int, C.c_int,
                            C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp], C.c_int),
    "njf_pack_resnetfc_backward": ([C.POINTER(ResnetFcWeights), _vp, C.c_int, _vp], C.c_int),
    "njf_resnetfc_backward": ([_vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp], C.c_int),
    "njf_pack_transformer_backward": ([_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp], C.c_int),
    "njf_transformer_backward": ([_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp], C.c_int),
    "njf_scatter_footprint": ([_vp, C.c_int, C.c_longlong, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp], C.c_int),
    "njf_bn_act": ([_vp, _vp, _vp, _vp, _vp, C.c_float, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp], C.c_int),
    "njf_relu_backward": ([_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp], C.c_int),
    "njf_reduce_frame_partials": ([_vp, C.c_int, _vp, _vp], C.c_int),
    "njf_assemble_frame": ([_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _vp, _vp, _vp], C.c_int),
    "njf_composite_backward": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp], C.c_int),
    "njf_alpha_weights": ([_vp, _vp, C.c_int, C.c_int, _vp, _vp], C.c_int),
    "njf_pdf_resample": ([_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp, _vp], C.c_int),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load_library() -> C.CDLL:
    """Load ``libnjf_hip.so`` (built by ``__graft_entry__.build()``); raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
                "g.build()'`).  There is no CPU fallback for the rendering hot path.")
        lib = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = lib
    return _lib


def _check(code: int) -> None:
    if code == 0:
        return
    msg = load_library().njf_error_string(code).decode()
    if code < 0:
        raise ValueError(f"njf_hip: {msg} (code {code})")
    raise RuntimeError(f"njf_hip: {msg} (hipError_t {code})")


import threading

_call = threading.local()  # per-thread: the device of the tensors handed to the entry point being assembled


def _note_device(t, name: str) -> None:
    """`t`: a tensor, or a Cameras / FeatureMap record (which remembers the device of the tensors it points to)."""
    new = t.device if torch.is_tensor(t) else getattr(t, "_device", None)
    if new is None:
        return
    dev = getattr(_call, "device", None)
    if dev is None:
        _call.device = new
    elif dev != new:
        _call.device = None   # the half-assembled call is abandoned: its device must not leak into the next entry point
        raise ValueError(f"njf_hip: {name} lives on {new} but other arguments of this call live on {dev}; all tensors "
                         "of one call must be on the same GPU")


class _RecordScope:
    """Pointers gathered inside the scope belong to a record built ahead of its call (Cameras, FeatureMap): their device
    is stored on the record instead of leaking into whichever entry point is assembled next."""

    def __enter__(self):
        self.prev = getattr(_call, "device", None)
        _call.device = None
        return self

    def close(self, record):
        record._device = getattr(_call, "device", None)
        return record

    def __exit__(self, *exc):
        _call.device = self.prev
        return False


def device_key(device) -> tuple:
    """Cache key of a device: (type, index) with an index-less ``torch.device('cuda')`` resolved to the CURRENT device -- a
    plain ``str(device)`` would give every GPU of a process the same key."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        return ("cuda", torch.cuda.current_device())
    return (d.type, d.index)


def cached_device_constant(cache: dict, key: tuple, device, make) -> torch.Tensor:
    """``cache[key]`` (key must end with ``device_key(device)``), built by ``make()`` -- a host tensor -- and uploaded ONCE.  A
    first use inside a HIP-graph capture would be a pageable host-to-device copy inside the capture: refused with a message that
    says what to do (every capture path here warms up with eager steps first, so this only fires on a mis-ordered caller)."""
    t = cache.get(key)
    if t is None:
        d = torch.device(device)
        if d.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("njf: a device constant would be uploaded inside a HIP-graph capture; run one eager step of the same "
                               "shape before capturing (the constant caches are filled on first use)")
        t = make().to(d)
        cache[key] = t
    return t


def map_dtype(precision: Optional[str]) -> torch.dtype:
    """Element type of the hoisted map a network of MFMA ``precision`` reads (include/njf_hip.h: NJF_PRECISION_F16)."""
    return torch.float16 if (DEFAULT_PRECISION if precision is None else precision) == "f16" else torch.float32


def _check_map_dtype(fmap, *precisions: Optional[str]) -> None:
    """The forward kernels read the hoisted map with the element type of their MFMA precision (fp16 maps for "f16", fp32 maps
    otherwise) and the C side cannot tell what an allocation holds: a half map read as fp32 is an out-of-bounds read of twice the
    allocation, an fp32 map read as halves is silent garbage.  Checked here, where the tensor is still known."""
    keep = getattr(fmap, "_keep", None)
    if keep is None:
        return
    for p in precisions:
        if keep.dtype != map_dtype(p):
            raise ValueError(f"njf_hip: the hoisted map is {keep.dtype} but a network of precision "
                             f"{DEFAULT_PRECISION if p is None else p!r} reads {map_dtype(p)} maps (project it with that precision)")


def _ptr(t: Optional[torch.Tensor], name: str = "tensor", dtype: torch.dtype = torch.float32) -> Optional[int]:
    if t is None:
        return None
    problem = None
    if not t.is_cuda:
        problem = f"njf_hip: {name} must live on the GPU (got {t.device}); there is no CPU path"
    elif t.dtype != dtype:
        problem = f"njf_hip: {name} must be {str(dtype).replace('torch.', '')} (got {t.dtype})"
    elif not t.is_contiguous():
        problem = f"njf_hip: {name} must be contiguous"
    if problem is not None:
        _call.device = None   # abandon the half-assembled call (a stale device used to leak into the next one)
        raise ValueError(problem)
    _note_device(t, name)
    return t.data_ptr()


def inverse(m: torch.Tensor) -> torch.Tensor:
    """Batched small-matrix inverse (camera matrices: torch.inverse in rendering/geometry.py:52,64).  fp32 [...,4,4]
    device tensors go through njf_invert_4x4 (one launch); anything else through torch.linalg.inv_ex, which unlike
    torch.linalg.inv does not synchronise with the host for its error check."""
    if m.is_cuda and m.dtype == torch.float32 and m.shape[-2:] == (4, 4) and m.numel() > 0 and not m.requires_grad:
        src = m.contiguous()
        out = torch.empty_like(src)
        _launch("njf_invert_4x4", load_library().njf_invert_4x4, _ptr(src, "matrices"), src.numel() // 16, _ptr(out, "out"))
        return out
    return torch.linalg.inv_ex(m).inverse


def _stream() -> int:
    """The torch stream of the device the call's tensors live on (NOT of the current device: a model loaded on cuda:N
    without torch.cuda.set_device(N) would otherwise launch on device 0's stream with device-N pointers)."""
    dev = getattr(_call, "device", None)
    return torch.cuda.current_stream(dev).cuda_stream


# Optional per-launch timing for bench.py: a list that receives (name, start_event, end_event) for every fused launch,
# recorded on the stream the kernel is launched on.  None (default) = no events, no overhead.
_profile_sink = None


def set_profile_sink(sink) -> None:
    global _profile_sink
    _profile_sink = sink


def _launch(name: str, fn, *args) -> None:
    """Run one C-ABI entry point on the device (and stream) of its tensor arguments, which `_ptr` collected while the
    argument list was built."""
    dev = getattr(_call, "device", None)
    _call.device = None
    if dev is None:
        raise ValueError(f"njf_hip: {name} called without any device tensor")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        if _profile_sink is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            _check(fn(*args, stream.cuda_stream))
            e1.record(stream)
            _profile_sink.append((name, e0, e1))
        else:
            _check(fn(*args, stream.cuda_stream))


def make_cameras(ctxt_w2c, ctxt_k, z_near, z_far, trgt_w2c=None, trgt_k=None, action=None, action_dim=None) -> Cameras:
    batch = ctxt_w2c.shape[0]
    a_dim = (0 if action is None else action.shape[-1]) if action_dim is None else action_dim
    with _RecordScope() as scope:
        cams = Cameras(_ptr(ctxt_w2c, "ctxt_w2c"), _ptr(ctxt_k, "ctxt_k"), _ptr(trgt_w2c, "trgt_w2c"),
                       _ptr(trgt_k, "trgt_k"), _ptr(z_near, "z_near"), _ptr(z_far, "z_far"), _ptr(action, "action"),
                       batch, a_dim)
        scope.close(cams)
    cams._keep = (ctxt_w2c, ctxt_k, z_near, z_far, trgt_w2c, trgt_k, action)  # keep tensors alive
    return cams


def make_feature_map(gmap: torch.Tensor) -> FeatureMap:
    """gmap: [B, Hf, Wf, C] channels-last hoisted map (float32; float16 for the networks of the plain-fp16 mode -- the
    precision of the forward call that reads it must agree, which the callers below guarantee by construction)."""
    if gmap.dtype not in (torch.float32, torch.float16):
        raise ValueError(f"njf_hip: gmap must be float32 or float16 (got {gmap.dtype})")
    with _RecordScope() as scope:
        fm = FeatureMap(_ptr(gmap, "gmap", gmap.dtype), gmap.shape[1], gmap.shape[2], gmap.shape[3])
        scope.close(fm)
    fm._keep = gmap
    return fm


# --------------------------------------------------------------------------------------
# packing
# --------------------------------------------------------------------------------------
def pack_resnetfc(params: Dict[str, torch.Tensor], prefix: str, w_out: torch.Tensor, b_out: torch.Tensor,
                  wz: Optional[torch.Tensor] = None, wz_col: int = 0, bz: Optional[torch.Tensor] = None,
                  precision: Optional[str] = None) -> None:
    """``params[prefix + 'lin_in.weight']`` ... (reference ResnetFC names) -> packed blobs.

    ``wz`` [512, ld] / ``bz`` [ld] receive this net's three lin_z layers at columns ``wz_col .. wz_col+383``."""
    def p(name):
        return _ptr(params[prefix + name].detach(), prefix + name)

    src = ResnetFcWeights()
    src.lin_in_w, src.lin_in_b = p("lin_in.weight"), p("lin_in.bias")
    for i in range(5):
        src.fc0_w[i], src.fc0_b[i] = p(f"blocks.{i}.fc_0.weight"), p(f"blocks.{i}.fc_0.bias")
        src.fc1_w[i], src.fc1_b[i] = p(f"blocks.{i}.fc_1.weight"), p(f"blocks.{i}.fc_1.bias")
    for i in range(3):
        src.lin_z_w[i], src.lin_z_b[i] = p(f"lin_z.{i}.weight"), p(f"lin_z.{i}.bias")
    src.lin_out_w, src.lin_out_b = p("lin_out.weight"), p("lin_out.bias")
    src.d_out = params[prefix + "lin_out.weight"].shape[0]
    wz_ptr = bz_ptr = None
    ld = ZDIM
    if wz is not None:
        ld = wz.shape[1]
        if wz.shape[0] != 512 or wz_col + ZDIM > ld or bz is None or bz.numel() != ld:
            raise ValueError("njf_hip: wz must be [512, ld] with wz_col + 384 <= ld and bz [ld]")
        wz_ptr = _ptr(wz, "wz") + 4 * wz_col
        bz_ptr = _ptr(bz, "bz") + 4 * wz_col
    _launch("njf_pack_resnetfc_ld", load_library().njf_pack_resnetfc_ld, C.byref(src), _ptr(w_out), _ptr(b_out), wz_ptr, ld, bz_ptr,
                                               precision_code(precision))


def pack_color_head(params: Dict[str, torch.Tensor], prefix: str, w_out: torch.Tensor, b_out: torch.Tensor,
                    precision: Optional[str] = None) -> None:
    def p(name):
        return _ptr(params[prefix + name].detach(), prefix + name)

    src = ColorHeadWeights(p("0.weight"), p("0.bias"), p("2.weight"), p("2.bias"), p("4.weight"), p("4.bias"))
    _launch("njf_pack_color_head", load_library().njf_pack_color_head, C.byref(src), _ptr(w_out), _ptr(b_out), precision_code(precision))


def pack_linear(weight: torch.Tensor, bias: Optional[torch.Tensor], kind: int, w_out: torch.Tensor,
                b_out: Optional[torch.Tensor] = None, precision: Optional[str] = None) -> None:
    """One Linear [d_out, d_in] -> fragment-major block (see include/njf_hip.h: njf_pack_linear)."""
    d_out, d_in = weight.shape
    _launch("njf_pack_linear", load_library().njf_pack_linear, _ptr(weight.detach().contiguous(), "weight"),
                                          _ptr(None if bias is None else bias.detach().contiguous(), "bias"), d_out, d_in,
                                          kind, _ptr(w_out, "w_out"), _ptr(b_out, "b_out"), precision_code(precision))


def project_features(feats: torch.Tensor, wz: torch.Tensor, bz: torch.Tensor, out: torch.Tensor,
                     precision: Optional[str] = None) -> None:
    """feats [B,512,Hf,Wf]; wz [512,N]; bz [N]; out [B,Hf,Wf,N] (float16 for precision "f16": ``map_dtype``)."""
    b, k, hf, wf = feats.shape
    n = wz.shape[1]
    if k != 512 or wz.shape[0] != 512 or tuple(out.shape) != (b, hf, wf, n):
        raise ValueError("njf_hip: project_features shape mismatch")
    _launch("njf_project_features_ld", load_library().njf_project_features_ld, _ptr(feats), _ptr(wz), n, _ptr(bz), b, hf * wf, n,
            _ptr(out, "out", map_dtype(precision)), precision_code(precision))


def project_pyramid(levels, wz: torch.Tensor, bz: torch.Tensor, out: torch.Tensor, precision: Optional[str] = None) -> None:
    """levels: the encoder's latents [B,C_l,H_l,W_l] in concatenation order (level 0 at the output resolution, channel
    counts summing to 512); wz [512,N]; bz [N]; out [B,H_0,W_0,N] -- the hoisted map of the concatenated, up-sampled
    feature map, without forming it (njf_project_pyramid)."""
    b = levels[0].shape[0]
    n = wz.shape[1]
    if tuple(out.shape) != (b, levels[0].shape[2], levels[0].shape[3], n) or wz.shape[0] != 512:
        raise ValueError("njf_hip: project_pyramid shape mismatch")
    arr = (PyramidLevel * len(levels))()
    keep = []
    ws_floats = 0
    for i, lv in enumerate(levels):
        lv = lv.contiguous()
        keep.append(lv)
        arr[i] = PyramidLevel(_ptr(lv), lv.shape[1], lv.shape[2], lv.shape[3])
        if i > 0 or (map_dtype(precision) == torch.float16 and len(levels) > 1):
            # (plain-fp16 map of a pyramid: the levels are summed in fp32 in the workspace and rounded once)
            ws_floats += b * lv.shape[2] * lv.shape[3] * n
    workspace = torch.empty(max(ws_floats, 1), dtype=torch.float32, device=out.device)
    _launch("njf_project_pyramid", load_library().njf_project_pyramid, arr, len(levels), _ptr(wz), n, _ptr(bz), b, n,
            _ptr(out, "out", map_dtype(precision)), _ptr(workspace), precision_code(precision))


_hoist_order_cache: Dict[tuple, torch.Tensor] = {}


def hoisted_channel_order(block_channels: int, device, precision: str) -> torch.Tensor:
    """pos[f] = position of logical feature f inside a block of ``block_channels`` hoisted-map channels of a network
    packed for MFMA ``precision`` (njf_hoisted_channel: the single definition of that order), as an index tensor on
    ``device``."""
    key = (block_channels, device_key(device), precision)
    if key not in _hoist_order_cache:
        lib = load_library()
        code = PRECISIONS[precision]
        pos = [lib.njf_hoisted_channel(f, block_channels, code) for f in range(block_channels)]
        if min(pos) < 0:
            _check(min(pos))
        _hoist_order_cache[key] = torch.tensor(pos, dtype=torch.long, device=device)
    return _hoist_order_cache[key]


def upsample_concat(levels) -> torch.Tensor:
    """The encoder output cat_l(upsample_l(latent_l)) as a channels-last matrix [B*H_0*W_0, sum C_l] (njf_upsample_concat):
    what the lin_z weight gradients contract against; levels as for project_pyramid."""
    b, _, h0, w0 = levels[0].shape
    arr = (PyramidLevel * len(levels))()
    keep = []
    for i, lv in enumerate(levels):
        lv = lv.contiguous()
        keep.append(lv)
        arr[i] = PyramidLevel(_ptr(lv), lv.shape[1], lv.shape[2], lv.shape[3])
    out = torch.empty(b * h0 * w0, sum(lv.shape[1] for lv in levels), dtype=torch.float32, device=levels[0].device)
    _launch("njf_upsample_concat", load_library().njf_upsample_concat, arr, len(levels), b, _ptr(out))
    return out


def upsample_concat_backward(grad: torch.Tensor, shapes) -> list:
    """Adjoint of ``upsample_concat``: grad [B*H_0*W_0, sum C_l] channels-last -> [gradient of latent l, NCHW] for the
    latents of shapes ``shapes`` = [(B,C_l,H_l,W_l), ...] (njf_upsample_concat_backward)."""
    b, _, h0, w0 = shapes[0]
    n = sum(sh[1] for sh in shapes)
    if tuple(grad.shape) != (b * h0 * w0, n):
        raise ValueError("njf_hip: upsample_concat_backward shape mismatch")
    outs = [torch.empty(tuple(sh), dtype=torch.float32, device=grad.device) for sh in shapes]
    arr = (PyramidLevel * len(shapes))()
    for i, (o, sh) in enumerate(zip(outs, shapes)):
        arr[i] = PyramidLevel(_ptr(o, "latent gradient"), sh[1], sh[2], sh[3])
    _launch("njf_upsample_concat_backward", load_library().njf_upsample_concat_backward, _ptr(grad, "grad"), arr, len(shapes), b)
    return outs


# --------------------------------------------------------------------------------------
# ops
# --------------------------------------------------------------------------------------
def generate_rays(coords, height, width, k_inv, c2w, origins, directions, z) -> None:
    batch, rays = origins.shape[0], origins.shape[1]
    _launch("njf_generate_rays", load_library().njf_generate_rays, _ptr(coords), height, width, _ptr(k_inv), _ptr(c2w), batch, rays,
                                            _ptr(origins), _ptr(directions), _ptr(z))


def _int_ptr(t: torch.Tensor) -> int:
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        _call.device = None
        raise ValueError("njf_hip: foot_idx / mask tensors must be contiguous int32 device tensors")
    _note_device(t, "foot_idx")
    return t.data_ptr()


def proposal_forward(origins, directions, cams: Cameras, fmap: FeatureMap, gmap_offset: int, w_pack, b_pack,
                     bins_in, s_in: int, u, s_out: int, anneal: float, bins_out, weights_out=None,
                     density_out=None, precision: Optional[str] = None,
                     dump: Optional[Dict[str, torch.Tensor]] = None) -> None:
    """``dump`` (training forward): tensors act [11,P,128], pe [P,64], foot_idx [P,4] int32, foot_w [P,4]."""
    rays_per_batch = origins.shape[1]
    dump_ref = None
    if dump is not None:
        if dump["act"].dtype not in (torch.float32, torch.float16):
            raise ValueError(f"njf_hip: activation dumps are float32 or float16 (got {dump['act'].dtype})")
        half = dump["act"].dtype == torch.float16   # 16-bit training storage (training.set_storage_precision)
        dump_ref = C.byref(ActivationDump(_ptr(dump["act"], "act", dump["act"].dtype), _ptr(dump["pe"]), _int_ptr(dump["foot_idx"]),
                                          _ptr(dump["foot_w"]), _int_ptr(dump["mask"]) if dump.get("mask") is not None else None,
                                          int(half)))
    _note_device(cams, "cameras")
    _note_device(fmap, "feature map")
    _check_map_dtype(fmap, precision)
    _launch("njf_proposal_forward", load_library().njf_proposal_forward, 
        _ptr(origins), _ptr(directions), rays_per_batch, C.byref(cams), C.byref(fmap), gmap_offset,
        _ptr(w_pack), _ptr(b_pack), _ptr(bins_in), int(bins_in.dim() > 1), s_in, _ptr(u), int(u.dim() > 1), s_out,
        float(anneal), _ptr(bins_out), _ptr(weights_out), _ptr(density_out), dump_ref, precision_code(precision))


def render_forward(origins, directions, cams: Cameras, fmap: FeatureMap, goff_density: int, goff_jacobian: int,
                   w_all: torch.Tensor, b_density, b_color, b_jacobian, bins, samples: int, outputs: Dict[str, torch.Tensor],
                   jacobian_kind: int = JACOBIAN_MLP, precision: Optional[str] = None,
                   jacobian_precision: Optional[str] = None) -> None:
    """``w_all`` is the single allocation [density | colour | jacobian head] of packed weights; ``jacobian_precision``
    (default: ``precision``) is the MFMA precision the Jacobian head's blob was packed for."""
    rays_per_batch = origins.shape[1]
    out = RenderOutputs()
    dump_f16 = 0
    for name, _ in RenderOutputs._fields_:
        if name == "dump_f16":
            continue
        t = outputs.get(name)
        if name in ("foot_idx", "jac_mask", "den_mask") and t is not None:  # the int32 outputs
            setattr(out, name, _int_ptr(t))
        elif name in ("jac_act", "den_act") and t is not None and t.dtype == torch.float16:   # 16-bit training storage
            setattr(out, name, _ptr(t, name, torch.float16))
            dump_f16 = 1
        else:
            setattr(out, name, _ptr(t, name))
    out.dump_f16 = dump_f16
    base = _ptr(w_all, "w_all")
    w_c = base + 4 * RESNET_W_FLOATS
    with_j = jacobian_kind != JACOBIAN_NONE
    w_j = w_c + 4 * COLOR_W_FLOATS if with_j else None
    _note_device(cams, "cameras")
    _note_device(fmap, "feature map")
    _check_map_dtype(fmap, precision, precision if jacobian_precision is None else jacobian_precision)
    _launch("njf_render_forward", load_library().njf_render_forward, 
        _ptr(origins), _ptr(directions), rays_per_batch, C.byref(cams), C.byref(fmap), goff_density, goff_jacobian,
        jacobian_kind, base, _ptr(b_density), w_c, _ptr(b_color), w_j, _ptr(b_jacobian) if with_j else None,
        _ptr(bins), samples, C.byref(out), precision_code(precision, jacobian_precision))


def points_forward(xyz, dirs, cams: Cameras, fmap: FeatureMap, goff_density: int, goff_jacobian: int, mode: int,
                   w_all, b_density, b_color=None, b_jacobian=None, jacobian_kind: int = JACOBIAN_NONE, density=None,
                   color=None, flow=None, jacobian=None, geo=None, precision: Optional[str] = None,
                   jacobian_precision: Optional[str] = None, features=None) -> None:
    """``features`` [5, B*N, 128] (ABI v18): the ResnetFC head's residual stream after each block (include/njf_hip.h)."""
    points_per_batch = xyz.shape[1]
    if features is not None and tuple(features.shape) != (5, xyz.shape[0] * points_per_batch, 128):
        raise ValueError(f"njf_hip: features must be [5, {xyz.shape[0] * points_per_batch}, 128] (got {tuple(features.shape)})")
    base = _ptr(w_all, "w_all")
    w_c = base + 4 * RESNET_W_FLOATS if mode == 1 else None
    with_j = mode == 1 and jacobian_kind != JACOBIAN_NONE
    w_j = (w_c + 4 * COLOR_W_FLOATS) if with_j else None
    _note_device(cams, "cameras")
    _note_device(fmap, "feature map")
    _check_map_dtype(fmap, precision, precision if jacobian_precision is None else jacobian_precision)
    _launch("njf_points_forward", load_library().njf_points_forward, 
        _ptr(xyz), _ptr(dirs), points_per_batch, C.byref(cams), C.byref(fmap), goff_density, goff_jacobian, mode,
        jacobian_kind if mode == 1 else JACOBIAN_NONE, base, _ptr(b_density), w_c, _ptr(b_color), w_j,
        _ptr(b_jacobian) if with_j else None, _ptr(density), _ptr(color), _ptr(flow), _ptr(jacobian), _ptr(geo),
        _ptr(features, "features"), precision_code(precision, jacobian_precision))


def solve_action(mean_position, jacobian, projection, target_flow, visible_mask, init_action, iterations: int,
                 damping: float, action) -> None:
    """mean_position [B,R,3], jacobian [B,R,3,A], projection [B,3,4], target_flow [B,R,2] -> action [B,A]."""
    b, r = target_flow.shape[:2]
    _launch("njf_solve_action", load_library().njf_solve_action, _ptr(mean_position), _ptr(jacobian), _ptr(projection), _ptr(target_flow),
                                           _ptr(visible_mask), _ptr(init_action), b, r, jacobian.shape[-1], int(iterations),
                                           float(damping), _ptr(action))


def scatter_footprint(grad, foot_idx, foot_w, out, run_length: int = 1) -> None:
    """out [T,S*C] += bilinear-footprint scatter of grad [P,C] or [S,P,C] (foot_idx [P,4] int32, foot_w [P,4]): the input
    gradient of the pixel-aligned sampling, see include/njf_hip.h.  A 3-D ``grad`` may be a strided view along its first
    axis (e.g. ``deltas[0:6:2]``): S gradients of the same points land side by side in ``out``'s columns, one launch.
    ``run_length``: samples per ray (consecutive points that mostly share texels are merged in registers before the
    atomics)."""
    if grad.dim() == 2:
        grad = grad.unsqueeze(0)
    slices, points, channels = grad.shape
    if grad.stride(2) != 1 or grad.stride(1) != channels:
        raise ValueError("njf_hip: scatter_footprint needs [P,C]-contiguous gradient slices")
    if tuple(foot_idx.shape) != (points, 4) or tuple(foot_w.shape) != (points, 4) or out.shape[1] != slices * channels:
        raise ValueError("njf_hip: scatter_footprint shape mismatch")
    if not out.is_contiguous():
        raise ValueError("njf_hip: scatter_footprint output must be contiguous")
    if not grad.is_cuda or grad.dtype != torch.float32:
        raise ValueError("njf_hip: grad must be a float32 GPU tensor; there is no CPU path")
    _note_device(grad, "grad")
    _launch("njf_scatter_footprint", load_library().njf_scatter_footprint, grad.data_ptr(), slices,
            grad.stride(0) if slices > 1 else 0, _int_ptr(foot_idx), _ptr(foot_w, "foot_w"), points, channels, out.shape[0],
            int(run_length), _ptr(out, "out"))


RESNET_BACKWARD_W_FLOATS = 21 * 8192
TRANSFORMER_BACKWARD_W_FLOATS = 13 * 8192    # NJF_TRANSFORMER_BACKWARD_CHUNKS
TRANSFORMER_BACKWARD_B_FLOATS = 3 * 192


def pack_transformer_backward(mats: torch.Tensor, biases: torch.Tensor, head_w: torch.Tensor, w_out: torch.Tensor,
                              b_out: torch.Tensor, precision: str = "f32") -> None:
    """Weights of njf_transformer_backward from the FOLDED head: mats [3,4,64,64] = (Mqk, Nov, W1', W2) per layer ([out, in]),
    biases [3,3,64] = (bqk, bo, b1'), head_w [3A,64] (include/njf_hip.h), packed for the chain's product form (``precision``:
    "f32" exact, or "f16x2")."""
    if precision not in BACKWARD_PRECISIONS:
        raise ValueError(f"njf_hip: backward precision must be one of {BACKWARD_PRECISIONS} (got {precision!r})")
    if tuple(mats.shape) != (3, 4, 64, 64) or tuple(biases.shape) != (3, 3, 64) or head_w.dim() != 2 or head_w.shape[1] != 64:
        raise ValueError("njf_hip: pack_transformer_backward shape mismatch")
    if w_out.numel() != TRANSFORMER_BACKWARD_W_FLOATS or b_out.numel() != TRANSFORMER_BACKWARD_B_FLOATS:
        raise ValueError("njf_hip: pack_transformer_backward output size mismatch")
    _launch("njf_pack_transformer_backward", load_library().njf_pack_transformer_backward, _ptr(mats.contiguous(), "mats"),
            _ptr(biases.contiguous(), "biases"), _ptr(head_w.contiguous(), "head_w"), head_w.shape[0], _ptr(w_out, "w_out"),
            _ptr(b_out, "b_out"), PRECISIONS[precision])


def transformer_backward(x: torch.Tensor, d_out: torch.Tensor, keys: int, w_backward: torch.Tensor, b_backward: torch.Tensor,
                         half_storage: bool = False, precision: str = "f32"):
    """The folded transformer head's data-gradient chain (include/njf_hip.h: njf_transformer_backward): x [4,P,64] (the residual
    stream the training forward dumped), d_out [P,3A] -> (wg_x [12,P,64], wg_dy [12,P,64], dx0 [P,64], column sums of the dY
    [12,64], unscale).  ``half_storage``: the pairs are fp16, the dY scaled by 2^k; ``unscale`` = 2^-k as a device scalar (else 1).
    ``precision``: the product form the weights were packed for ("f32" exact, "f16x2" split fp16 on d_out x 2^k)."""
    if precision not in BACKWARD_PRECISIONS:
        raise ValueError(f"njf_hip: backward precision must be one of {BACKWARD_PRECISIONS} (got {precision!r})")
    points = d_out.shape[0]
    if tuple(x.shape) != (4, points, 64) or w_backward.numel() != TRANSFORMER_BACKWARD_W_FLOATS \
            or b_backward.numel() != TRANSFORMER_BACKWARD_B_FLOATS:
        raise ValueError("njf_hip: transformer_backward shape mismatch")
    dev = d_out.device
    pair_dtype = torch.float16 if half_storage else torch.float32
    wg_x = torch.empty(12, points, 64, dtype=pair_dtype, device=dev)
    wg_dy = torch.empty(12, points, 64, dtype=pair_dtype, device=dev)
    dx0 = torch.empty(points, 64, dtype=torch.float32, device=dev)
    partial = torch.empty((points + 31) // 32, 12, 64, dtype=torch.float32, device=dev)
    d_out = d_out.contiguous()
    absmax = d_out.abs().amax().reshape(1) if (half_storage or precision != "f32") else None
    _launch("njf_transformer_backward", load_library().njf_transformer_backward, _ptr(x, "x"), _ptr(d_out, "d_out"), d_out.shape[1],
            int(keys), points, _ptr(w_backward, "w_backward"), _ptr(b_backward, "b_backward"), _ptr(wg_x, "wg_x", pair_dtype),
            _ptr(wg_dy, "wg_dy", pair_dtype), _ptr(dx0, "dx0"), _ptr(partial, "colsum_partial"), int(half_storage),
            _ptr(absmax, "d_out_absmax"), PRECISIONS[precision])
    unscale = power_of_two_unscale(absmax) if half_storage else None
    return wg_x, wg_dy, dx0, partial.sum(0), unscale


BACKWARD_PRECISIONS = ("f32", "f16x2")


def pack_resnetfc_backward(params: Dict[str, torch.Tensor], prefix: str, w_out: torch.Tensor, precision: str = "f32") -> None:
    """Transposed weights of one ResnetFC in the chunk order njf_resnetfc_backward streams them (include/njf_hip.h), packed for
    the chain's product form (``precision``: "f32" exact, or "f16x2")."""
    if precision not in BACKWARD_PRECISIONS:
        raise ValueError(f"njf_hip: backward precision must be one of {BACKWARD_PRECISIONS} (got {precision!r})")
    def p(name):
        return _ptr(params[prefix + name].detach().contiguous(), prefix + name)

    src = ResnetFcWeights()
    for i in range(5):
        src.fc0_w[i], src.fc1_w[i] = p(f"blocks.{i}.fc_0.weight"), p(f"blocks.{i}.fc_1.weight")
    src.lin_out_w = p("lin_out.weight")
    src.d_out = params[prefix + "lin_out.weight"].shape[0]
    _launch("njf_pack_resnetfc_backward", load_library().njf_pack_resnetfc_backward, C.byref(src), _ptr(w_out, "w_out"),
            PRECISIONS[precision])


def power_of_two_unscale(absmax: torch.Tensor) -> torch.Tensor:
    """2^-k for the k = 6 - exponent(max|d_out|) that njf_resnetfc_backward scales by (device-side, no host round trip)."""
    _, e = torch.frexp(absmax)                      # absmax = f * 2^e, f in [0.5, 1): the kernel's frexpf
    k = torch.clamp(6 - e, -120, 120)
    ok = (absmax > 0) & (absmax < 3.0e38)
    return torch.where(ok, torch.ldexp(torch.ones_like(absmax), -k), torch.ones_like(absmax))


def resnetfc_backward_f16_storage(d_out: torch.Tensor, w_backward: torch.Tensor, mask: torch.Tensor, precision: str = "f32"):
    """The chain under the 16-bit training storage: -> (latent deltas [3,P,128] fp32 = slices 0, 2, 4, deltas16 [11,P,128] fp16 =
    deltas x 2^k, column sums [11,128] fp32, 2^-k as a device scalar)."""
    points = d_out.shape[0]
    if tuple(mask.shape) != (11, points, 4) or w_backward.numel() != RESNET_BACKWARD_W_FLOATS:
        raise ValueError("njf_hip: resnetfc_backward shape mismatch")
    if precision not in BACKWARD_PRECISIONS:
        raise ValueError(f"njf_hip: backward precision must be one of {BACKWARD_PRECISIONS} (got {precision!r})")
    dev = d_out.device
    latent = torch.empty(3, points, 128, dtype=torch.float32, device=dev)
    deltas16 = torch.empty(11, points, 128, dtype=torch.float16, device=dev)
    partial = torch.empty((points + 31) // 32, 11, 128, dtype=torch.float32, device=dev)
    d_out = d_out.contiguous()
    absmax = d_out.abs().amax().reshape(1)
    _launch("njf_resnetfc_backward", load_library().njf_resnetfc_backward, _ptr(d_out, "d_out"), d_out.shape[1], None,
            _ptr(w_backward, "w_backward"), points, _ptr(latent, "deltas"), _ptr(partial, "colsum_partial"), _int_ptr(mask),
            PRECISIONS[precision], _ptr(absmax, "d_out_absmax"), _ptr(deltas16, "deltas16", torch.float16))
    return latent, deltas16, partial.sum(0), power_of_two_unscale(absmax)


def resnetfc_backward(d_out: torch.Tensor, act: torch.Tensor, w_backward: torch.Tensor, want_colsum: bool = False,
                      mask: Optional[torch.Tensor] = None, precision: str = "f32"):
    """deltas [11,P,128] of one ResnetFC's backward pass (include/njf_hip.h: njf_resnetfc_backward): d_out [P,d_out],
    act [11,P,128] (dumped ReLU'd layer inputs), w_backward from pack_resnetfc_backward.  ``want_colsum``: also return
    the column sums [11,128] of every deltas slice (the bias gradients), reduced from the kernel's per-tile partials.
    ``mask`` [11,P,4] int32: the ReLU masks the same forward dumped (ABI v17) -- the chain then reads them instead of ``act``."""
    points = d_out.shape[0]
    if tuple(act.shape) != (11, points, 128) or w_backward.numel() != RESNET_BACKWARD_W_FLOATS:
        raise ValueError("njf_hip: resnetfc_backward shape mismatch")
    if mask is not None and tuple(mask.shape) != (11, points, 4):
        raise ValueError("njf_hip: resnetfc_backward mask must be [11, P, 4] int32")
    if precision not in BACKWARD_PRECISIONS:
        raise ValueError(f"njf_hip: backward precision must be one of {BACKWARD_PRECISIONS} (got {precision!r})")
    deltas = torch.empty_like(act)
    partial = torch.empty((points + 31) // 32, 11, 128, dtype=torch.float32, device=act.device) if want_colsum else None
    d_out = d_out.contiguous()
    # "f16x2": the chain runs on d_out scaled by a power of two taken from its largest entry (one reduction, no host round trip)
    absmax = d_out.abs().amax().reshape(1) if precision == "f16x2" else None
    _launch("njf_resnetfc_backward", load_library().njf_resnetfc_backward, _ptr(d_out, "d_out"), d_out.shape[1],
            _ptr(act, "act"), _ptr(w_backward, "w_backward"), points, _ptr(deltas, "deltas"), _ptr(partial, "colsum_partial"),
            _int_ptr(mask) if mask is not None else None, PRECISIONS[precision], _ptr(absmax, "d_out_absmax"), None)
    return (deltas, partial.sum(0)) if want_colsum else deltas


RELU_BACKWARD_ROWS = 512  # rows per workgroup of njf_relu_backward (one partial column-sum row each)


def relu_backward(upstream: torch.Tensor, act: torch.Tensor, residual: Optional[torch.Tensor] = None, want_colsum: bool = True):
    """(residual + upstream * [act > 0], its column sums) -- one layer step of the ResnetFC backward chain in one launch
    (include/njf_hip.h: njf_relu_backward).  All tensors [P,C] contiguous fp32 on the device."""
    points, channels = upstream.shape
    if act.shape != upstream.shape or (residual is not None and residual.shape != upstream.shape):
        raise ValueError("njf_hip: relu_backward shape mismatch")
    out = torch.empty_like(upstream)
    blocks = (points + RELU_BACKWARD_ROWS - 1) // RELU_BACKWARD_ROWS
    partial = torch.empty(blocks, channels, dtype=torch.float32, device=upstream.device) if want_colsum else None
    _launch("njf_relu_backward", load_library().njf_relu_backward, _ptr(upstream, "upstream"), _ptr(act, "act"), _ptr(residual, "residual"), points,
                                            channels, RELU_BACKWARD_ROWS, _ptr(out, "out"), _ptr(partial, "partial"))
    return out, (partial.sum(0) if want_colsum else None)


def bn_act(x: torch.Tensor, bn: "torch.nn.BatchNorm2d", skip: Optional[torch.Tensor] = None, relu: bool = True) -> torch.Tensor:
    """[relu](batch_norm_eval(x) [+ skip]) IN PLACE on x ([B,C,H,W] contiguous fp32, a convolution's fresh output): the epilogue of
    a convolution of the frozen encoder trunk as one launch (include/njf_hip.h: njf_bn_act).  ``bn``: an affine BatchNorm2d with
    running statistics, evaluated on them (eval mode)."""
    b, c, h, w = x.shape
    if skip is not None and (skip.shape != x.shape or not skip.is_contiguous() or skip.dtype != torch.float32):
        raise ValueError("njf_hip: bn_act skip must be a contiguous float32 tensor of x's shape")
    if not x.is_contiguous() or bn.running_mean is None or bn.weight is None or bn.weight.numel() != c:
        raise ValueError("njf_hip: bn_act needs a contiguous NCHW tensor and an affine BatchNorm2d with running statistics")
    _launch("njf_bn_act", load_library().njf_bn_act, _ptr(x, "x"), _ptr(bn.weight.detach(), "gamma"), _ptr(bn.bias.detach(), "beta"),
            _ptr(bn.running_mean, "running_mean"), _ptr(bn.running_var, "running_var"), float(bn.eps), _ptr(skip, "skip"), int(relu),
            b, c, h * w, _ptr(x, "out"))
    return x


def frame_partial_groups(total_rays: int) -> int:
    """Rows of NjfRenderOutputs.frame_partials for a launch of ``total_rays`` rays (one per workgroup)."""
    per = load_library().njf_rays_per_workgroup()
    return (total_rays + per - 1) // per


def reduce_frame_partials(partials: torch.Tensor, out4: torch.Tensor) -> None:
    """partials [groups,4] -> out4 [4] = (min_t, max_t, sum (rgb - trgt)^2, sum (flow - trgt)^2), fixed order."""
    _launch("njf_reduce_frame_partials", load_library().njf_reduce_frame_partials, _ptr(partials, "partials"),
            partials.shape[0], _ptr(out4, "out4"))


def assemble_frame(packets: torch.Tensor, batch: int, rays_per_batch: int, frame: torch.Tensor, scalars6: torch.Tensor,
                   rgb_scale: float = 0.0, flow_scale: float = 0.0) -> None:
    """packets [world, packet_floats] (all-gathered, include/njf_hip.h: njf_assemble_frame) -> frame [B,R,6] with the
    depth clipped to the global bounds, scalars6 = (min, max, S_rgb, S_flow, S_rgb * rgb_scale, S_flow * flow_scale)."""
    world, packet_floats = packets.shape
    if tuple(frame.shape) != (batch, rays_per_batch, 6) or scalars6.numel() != 6:
        raise ValueError("njf_hip: assemble_frame shape mismatch")
    _launch("njf_assemble_frame", load_library().njf_assemble_frame, _ptr(packets, "packets"), world, packet_floats, batch,
            rays_per_batch, float(rgb_scale), float(flow_scale), _ptr(frame, "frame"), _ptr(scalars6, "scalars6"))


def composite_backward(deltas, steps, sigma, color=None, g_weights=None, g_rgb=None, g_depth=None, want_color: bool = True):
    """(g_sigma [rays,S], g_color [rays,S,3] | None) of alpha compositing (njf_composite_backward): deltas / steps / sigma /
    g_weights [rays,S], color [rays,S,3], g_rgb [rays,3], g_depth [rays], contiguous fp32."""
    if deltas.dim() != 2:
        raise ValueError("njf_hip: composite_backward takes [rays, samples] fields")
    rays, samples = deltas.shape
    for name, t, shape in (("steps", steps, (rays, samples)), ("sigma", sigma, (rays, samples)), ("g_weights", g_weights, (rays, samples)),
                           ("color", color, (rays, samples, 3)), ("g_rgb", g_rgb, (rays, 3)), ("g_depth", g_depth, (rays,))):
        if t is not None and tuple(t.shape) != shape:
            raise ValueError(f"njf_hip: composite_backward: {name} has shape {tuple(t.shape)}, expected {shape}")
    g_sigma = torch.empty(rays, samples, dtype=torch.float32, device=deltas.device)
    g_color = torch.empty(rays, samples, 3, dtype=torch.float32, device=deltas.device) if (color is not None and want_color) else None
    _launch("njf_composite_backward", load_library().njf_composite_backward, _ptr(deltas, "deltas"), _ptr(steps, "steps"),
            _ptr(sigma, "sigma"), _ptr(color, "color"), _ptr(g_weights, "g_weights"), _ptr(g_rgb, "g_rgb"),
            _ptr(g_depth, "g_depth"), rays, samples, _ptr(g_sigma, "g_sigma"), _ptr(g_color, "g_color"))
    return g_sigma, g_color


def alpha_weights(deltas, densities, weights) -> None:
    samples = deltas.shape[-1]
    rays = deltas.numel() // samples
    _launch("njf_alpha_weights", load_library().njf_alpha_weights, _ptr(deltas), _ptr(densities), rays, samples, _ptr(weights))


def pdf_resample(weights, bins_in, u, s_out: int, anneal: float, bins_out) -> None:
    s_in = weights.shape[-1]
    rays = weights.numel() // s_in
    _launch("njf_pdf_resample", load_library().njf_pdf_resample, _ptr(weights), _ptr(bins_in), int(bins_in.dim() > 1), s_in, _ptr(u),
                                           int(u.dim() > 1), s_out, float(anneal), rays, _ptr(bins_out))
