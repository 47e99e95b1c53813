"""Ray bundles, samples and samplers -- host-side mirror of ``rendering/ray_samplers.py``.

The containers keep the reference's field names and shapes.  ``RaySamples.get_weights`` and
``PDFSampler`` run in HIP (``njf_alpha_weights`` / ``njf_pdf_resample``); ``ProposalNetworkSampler``
has two routes with identical results: the reference's generic callback loop (any ``density_fns``)
and the fused route used by ``Model`` (``njf_proposal_forward``: sampling + proposal MLP + weights +
inverse-CDF in one kernel per level).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import hip
from .renderer import pdf_u_eval, uniform_bins


@dataclass
class RaySamples:
    """ray_samplers.py:28-45."""
    origins: torch.Tensor        # [..., 1, 3]
    directions: torch.Tensor     # [..., 1, 3]
    starts: torch.Tensor         # [..., S, 1]
    ends: torch.Tensor           # [..., S, 1]
    deltas: Optional[torch.Tensor] = None
    spacing_starts: Optional[torch.Tensor] = None
    spacing_ends: Optional[torch.Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None

    def get_positions(self) -> torch.Tensor:
        """ray_samplers.py:48-55."""
        return self.origins + self.directions * (self.starts + self.ends) / 2

    @torch.no_grad()
    def get_weights(self, densities: torch.Tensor) -> torch.Tensor:
        """ray_samplers.py:77-101 (alpha * transmittance) via the HIP scan kernel."""
        deltas = self.deltas.expand_as(densities).contiguous()
        out = torch.empty_like(deltas)
        hip.alpha_weights(deltas[..., 0].contiguous(), densities[..., 0].contiguous(), out[..., 0])
        return out

    def spacing_bins(self) -> torch.Tensor:
        """[..., S+1] bin edges in the spacing domain (ray_samplers.py:418-424)."""
        return torch.cat([self.spacing_starts[..., 0], self.spacing_ends[..., -1:, 0]], dim=-1)


@dataclass
class RayBundle:
    """ray_samplers.py:104-147."""
    origins: torch.Tensor     # [..., 3]
    directions: torch.Tensor  # [..., 3]
    nears: Optional[torch.Tensor] = None  # [..., 1]
    fars: Optional[torch.Tensor] = None

    def __len__(self):
        return torch.numel(self.origins) // self.origins.shape[-1]

    def get_ray_samples(self, bin_starts, bin_ends, spacing_starts=None, spacing_ends=None,
                        spacing_to_euclidean_fn=None) -> RaySamples:
        return RaySamples(origins=self.origins[..., None, :], directions=self.directions[..., None, :], starts=bin_starts,
                          ends=bin_ends, deltas=bin_ends - bin_starts, spacing_starts=spacing_starts,
                          spacing_ends=spacing_ends, spacing_to_euclidean_fn=spacing_to_euclidean_fn)

    def samples_from_bins(self, bins: torch.Tensor) -> RaySamples:
        """Uniform spacing (identity spacing_fn): t = b*far + (1-b)*near (ray_samplers.py:240-252)."""
        near, far = self.nears, self.fars
        fn = lambda x: x * far + (1 - x) * near
        e = fn(bins)
        return self.get_ray_samples(e[..., :-1, None], e[..., 1:, None], bins[..., :-1, None], bins[..., 1:, None], fn)


class Sampler(nn.Module):
    def __init__(self, num_samples: Optional[int] = None):
        super().__init__()
        self.num_samples = num_samples

    def forward(self, *args, **kwargs):
        return self.generate_ray_samples(*args, **kwargs)


class UniformSampler(Sampler):
    """ray_samplers.py:197-276 (SpacedSampler with identity spacing)."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False):
        super().__init__(num_samples)
        self.train_stratified, self.single_jitter = train_stratified, single_jitter

    def spacing_bins(self, ray_bundle: RayBundle, num_samples: int, shared_ok: bool = False) -> torch.Tensor:
        """Bin edges [..., S+1]; in eval mode they are ray-independent and, when ``shared_ok``, returned as [S+1]."""
        dev = ray_bundle.origins.device
        shape = ray_bundle.origins.shape[:-1]
        bins = uniform_bins(num_samples, dev)
        if self.train_stratified and self.training:
            rand = torch.rand((*shape, 1 if self.single_jitter else num_samples + 1), dtype=bins.dtype, device=dev)
            centers = (bins[1:] + bins[:-1]) / 2.0
            upper = torch.cat([centers, bins[-1:]], -1)
            lower = torch.cat([bins[:1], centers], -1)
            return lower + (upper - lower) * rand
        return bins if shared_ok else bins.repeat(*shape, 1)

    def generate_ray_samples(self, ray_bundle: RayBundle, num_samples: Optional[int] = None) -> RaySamples:
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        return ray_bundle.samples_from_bins(self.spacing_bins(ray_bundle, num_samples))


_U_BASE: dict = {}


def _train_u_base(nb: int, device) -> torch.Tensor:
    """linspace(0, 1 - 1/nb, nb) of ray_samplers.py:391-394 on ``device``, computed once on the host and uploaded once per
    (nb, device): same values as before, no host-to-device copy per training step (which a step recorded into a HIP graph
    could not contain)."""
    return hip.cached_device_constant(_U_BASE, (nb, hip.device_key(device)), device,
                                      lambda: torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb))


class PDFSampler(Sampler):
    """ray_samplers.py:326-451."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = True, histogram_padding: float = 0.01):
        super().__init__(num_samples)
        if include_original:
            raise NotImplementedError("include_original=True is never used by the reference model (ray_samplers.py:483)")
        if histogram_padding != 0.01:
            raise ValueError("the HIP kernel implements the reference's histogram_padding=0.01")
        self.train_stratified, self.single_jitter = train_stratified, single_jitter

    def u_values(self, batch_shape, num_samples: int, device, shared_ok: bool = False) -> torch.Tensor:
        """ray_samplers.py:388-409."""
        nb = num_samples + 1
        if self.train_stratified and self.training:
            u = _train_u_base(nb, device).expand((*batch_shape, nb))
            rand = torch.rand((*batch_shape, 1 if self.single_jitter else nb), device=device) / nb
            return (u + rand).contiguous()
        u = pdf_u_eval(num_samples, device)
        return u if shared_ok else u.expand((*batch_shape, nb)).contiguous()

    @torch.no_grad()
    def generate_ray_samples(self, ray_bundle: RayBundle = None, ray_samples: RaySamples = None,
                             weights: torch.Tensor = None, num_samples: Optional[int] = None, eps: float = 1e-5):
        if ray_samples is None or ray_bundle is None:
            raise ValueError("ray_samples must be provided")
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        assert ray_samples.spacing_starts is not None and ray_samples.spacing_ends is not None, \
            "ray_sample spacing_starts and spacing_ends must be provided"
        w = weights[..., 0].contiguous()
        shape = w.shape[:-1]
        bins_in = ray_samples.spacing_bins().contiguous()
        u = self.u_values(shape, num_samples, w.device, shared_ok=True)
        bins = torch.empty((*shape, num_samples + 1), dtype=torch.float32, device=w.device)
        hip.pdf_resample(w, bins_in, u, num_samples, 1.0, bins)
        return ray_bundle.samples_from_bins(bins)


class ProposalNetworkSampler(Sampler):
    """ray_samplers.py:454-552."""

    def __init__(self, num_proposal_samples_per_ray: Tuple[int, ...] = (64,), num_nerf_samples_per_ray: int = 32,
                 num_proposal_network_iterations: int = 2, single_jitter: bool = False,
                 update_sched: Callable = lambda x: 1, initial_sampler: Optional[Sampler] = None):
        super().__init__()
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        if self.num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        if initial_sampler is None:
            raise NotImplementedError("the default UniformLinDispPiecewiseSampler is dead code in the reference "
                                      "(Model always passes a UniformSampler, model.py:189)")
        self.initial_sampler = initial_sampler
        self.pdf_sampler = PDFSampler(include_original=False, single_jitter=single_jitter)
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def _level_counts(self):
        n = self.num_proposal_network_iterations
        return [self.num_proposal_samples_per_ray[i] if i < n else self.num_nerf_samples_per_ray for i in range(n + 1)]

    def _mark_updated(self):
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        if updated:
            self._steps_since_update = 0
        return updated

    @torch.no_grad()
    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None,
                             density_fns: Optional[List[Callable]] = None):
        """Generic route: arbitrary density callbacks, level loop of ray_samplers.py:515-548."""
        assert ray_bundle is not None
        assert density_fns is not None
        assert len(density_fns) == self.num_proposal_network_iterations
        weights_list, ray_samples_list = [], []
        counts = self._level_counts()
        n = self.num_proposal_network_iterations
        weights = ray_samples = None
        self._mark_updated()
        for lvl in range(n + 1):
            if lvl == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=counts[0])
            else:
                annealed = torch.pow(weights, self._anneal)
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, annealed, num_samples=counts[lvl])
            if lvl < n:
                density = density_fns[lvl](ray_samples.get_positions())
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        return ray_samples, weights_list, ray_samples_list

    @torch.no_grad()
    def generate_ray_samples_fused(self, ray_bundle: RayBundle, networks: Sequence, pixel_encoding, z_near, z_far,
                                   want_lists: bool, dump_out: Optional[list] = None,
                                   feature_maps: Optional[Sequence] = None):
        """Fused route: one ``njf_proposal_forward`` per level.  ``networks`` are DensityDecoderMlp modules.
        ``dump_out`` (perception-mode training): receives one dict per level with the proposal net's backward-pass
        inputs (``act``, ``pe``, ``foot_idx``, ``foot_w``), its ``density`` [B,R,S] and ``updated`` -- whether this
        step trains the proposal nets (the grad / no-grad schedule of ray_samplers.py:512-549).  ``feature_maps``: per
        level (hip.FeatureMap, first channel) when the caller has hoisted every network of the frame into one map
        (Model._joint_hoist); default: each network's own map."""
        assert len(networks) == self.num_proposal_network_iterations
        from .decoder import _cameras

        o, d = ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous()
        b, r = o.shape[:2]
        dev = o.device
        counts = self._level_counts()
        self.initial_sampler.train(self.training)
        self.pdf_sampler.train(self.training)
        updated = self._mark_updated()
        cams = _cameras(pixel_encoding, False, z_near, z_far)
        bins = self.initial_sampler.spacing_bins(ray_bundle, counts[0], shared_ok=True)
        weights_list, bins_list = [], []
        for lvl, net in enumerate(networks):
            s_in, s_out = counts[lvl], counts[lvl + 1]
            w, bias = net.packed()
            if feature_maps is None:
                fmap, goff = hip.make_feature_map(net.hoisted_map(pixel_encoding.features)), 0
            else:
                fmap, goff = feature_maps[lvl]
            u = self.pdf_sampler.u_values((b, r), s_out, dev, shared_ok=True)
            bins_out = torch.empty(b, r, s_out + 1, dtype=torch.float32, device=dev)
            w_out = torch.empty(b, r, s_in, dtype=torch.float32, device=dev) if want_lists else None
            dump = sigma = None
            if dump_out is not None:
                pts = b * r * s_in
                sigma = torch.empty(b, r, s_in, dtype=torch.float32, device=dev)
                from . import training as _tr   # (fp16 under the opt-in 16-bit training storage)
                dump = {"act": torch.empty(11, pts, 128, dtype=_tr.activation_dump_dtype(net.precision), device=dev),
                        "pe": torch.empty(pts, 64, dtype=torch.float32, device=dev),
                        "foot_idx": torch.empty(pts, 4, dtype=torch.int32, device=dev),
                        "foot_w": torch.empty(pts, 4, dtype=torch.float32, device=dev),
                        "mask": torch.empty(11, pts, 4, dtype=torch.int32, device=dev)}   # ReLU masks for the backward chain (ABI v17)
                dump_out.append({**dump, "density": sigma, "updated": updated, "forward_precision": net.precision})
            hip.proposal_forward(o, d, cams, fmap, goff, w, bias, bins.contiguous(), s_in, u, s_out, self._anneal, bins_out,
                                 w_out, sigma, precision=net.precision, dump=dump)
            if want_lists:
                weights_list.append(w_out[..., None])
                bins_list.append(bins if bins.dim() > 1 else bins.expand(b, r, -1))
            bins = bins_out
        return bins, weights_list, bins_list

    def __str__(self):
        return (f"ProposalNetworkSampler(num_proposal_samples_per_ray={self.num_proposal_samples_per_ray}, "
                f"num_nerf_samples_per_ray={self.num_nerf_samples_per_ray}, "
                f"num_proposal_network_iterations={self.num_proposal_network_iterations})")
