"""Anchor statistics and densification of the Scaffold model on the GPU.

Host mirror of GaussianModel::training_statis / adjust_anchor / anchor_growing / prune_anchor
(src/gaussian_model.cpp:1459-1503, 1701-1762, 1559-1699, 1505-1558) over the C ABI of include/segs_densify.h, on the
candidate-domain layout of neural_gaussians.py.  The per-iteration statistics and each growing level run as fused HIP
kernels; the tensor bookkeeping of adjust_anchor (appending rows, resetting counters, pruning by boolean mask with the
Adam moments) is done with device tensor ops inside the model's capacity-sized buckets.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _capi
from .neural_gaussians import NeuralGaussians, ScaffoldModel, _p


@dataclass
class DensifyParams:
    """Model.* / Optimization.* keys of cfg/gaussian_mapper/RGB-D/Replica/office0.yaml:13-18,130-137."""
    voxel_size: float = 0.001
    update_depth: int = 3
    update_init_factor: int = 16
    update_hierachy_factor: int = 4
    start_stat: int = 500
    update_from: int = 1500
    update_interval: int = 100
    update_until: int = 25500
    min_opacity: float = 0.005
    success_threshold: float = 0.8
    densify_grad_threshold: float = 0.0002


class AnchorDensifier:
    def __init__(self, model: ScaffoldModel, params: Optional[DensifyParams] = None):
        self.model, self.p = model, params or DensifyParams()
        self._lib = model._lib
        self._alloc_stats(model.capacity)

    STAT_NAMES = ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom")

    def _alloc_stats(self, capacity: int):
        """The four accumulators are segments of ONE flat tensor (`_stats_flat`), and so is their keyframe-parallel shadow
        `_delta_flat`: with N ranks every rank accumulates its own keyframe's increments there, and reduce_statistics() folds
        the sum over ranks into the replicated accumulators with one all-reduce right before adjust_anchor."""
        f = dict(dtype=torch.float32, device=self.model.device)
        no = self.model.dims.n_offsets
        old, old_delta = getattr(self, "_stats", None), getattr(self, "_delta", None)
        self._stats_capacity = capacity
        sizes = (capacity, capacity, capacity * no, capacity * no)
        self._stats_flat, self._delta_flat = torch.zeros(sum(sizes), **f), torch.zeros(sum(sizes), **f)
        self._stats, self._delta, off = {}, {}, 0
        for name, n in zip(self.STAT_NAMES, sizes):
            self._stats[name], self._delta[name] = self._stats_flat[off:off + n], self._delta_flat[off:off + n]
            off += n
        for new, prev in ((self._stats, old), (self._delta, old_delta)):
            if prev is not None:
                for k, t in prev.items():
                    new[k][:t.numel()] = t

    def reduce_statistics(self, process_group=None):
        """Fold every rank's increments since the last call into the replicated accumulators: one all-reduce(sum) of the
        flat shadow, the same words on every rank, so adjust_anchor then runs bit-identically everywhere."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            dist.all_reduce(self._delta_flat, group=process_group)
        self._stats_flat += self._delta_flat
        self._delta_flat.zero_()

    # views over the live rows, shaped like the reference's tensors
    def stat(self, name):
        A, no = self.model.A, self.model.dims.n_offsets
        n = A if name in ("opacity_accum", "anchor_demon") else A * no
        return self._stats[name][:n].view(-1, 1)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.model.device).cuda_stream)

    def training_statis(self, neural: NeuralGaussians, visible_radii: torch.Tensor, radii: torch.Tensor, dL_dmean2D: torch.Tensor,
                        skip_flag: Optional[C.c_void_p] = None, into_delta: bool = False):
        """src/gaussian_model.cpp:1459-1503 in the candidate domain (called between start_stat and update_until).
        `skip_flag`: device address of the resident rasterizer's overflow word; the pass is then dropped on the device.
        `into_delta`: keyframe-parallel ranks accumulate into the shadow that reduce_statistics() sums over ranks."""
        m = self.model
        s = self._delta if into_delta else self._stats
        st = self._lib.segs_training_statis_guarded(m.A, m.dims.n_offsets, _p(neural.neural_opacity), _p(visible_radii), _p(radii),
                                                    _p(dL_dmean2D), _p(s["opacity_accum"]), _p(s["anchor_demon"]),
                                                    _p(s["offset_gradient_accum"]), _p(s["offset_denom"]), skip_flag, self._stream())
        _capi.check(st, "segs_training_statis_guarded")

    # ---- adjust_anchor ---------------------------------------------------------------------------------------------
    def _append(self, new_anchor: torch.Tensor, new_feat: torch.Tensor, cur_size: float, new_scaling: Optional[torch.Tensor] = None):
        """:1623-1696: concatenate the new rows to the six tensors, zero-extend the Adam moments and the counters."""
        m = self.model
        n_new = new_anchor.shape[0]
        A0, A1 = m.A, m.A + n_new
        if A1 > m.capacity:
            m.reserve(int(A1 * 1.5) + 1024)
        if A1 > self._stats_capacity:
            self._alloc_stats(m.capacity)
        m.A = A1
        for bucket in (m.grads, m.exp_avg, m.exp_avg_sq):
            for name in m.widths:
                m._view(bucket, name)[A0:A1] = 0
        m.param("anchor")[A0:A1] = new_anchor
        m.param("offset")[A0:A1] = 0
        m.param("anchor_feat")[A0:A1] = new_feat
        m.param("scaling")[A0:A1] = (torch.log(torch.ones((n_new, 6), dtype=torch.float32, device=m.device) * cur_size)
                                     if new_scaling is None else new_scaling)
        m.rotation[A0:A1] = 0
        m.rotation[A0:A1, 0] = 1.0
        x = 0.1 * torch.ones((n_new, 1), dtype=torch.float32, device=m.device)
        m.opacity[A0:A1] = torch.log(x / (1 - x))
        self._stats["anchor_demon"][A0:A1] = 0
        self._stats["opacity_accum"][A0:A1] = 0

    def increase_pcd(self, points: torch.Tensor) -> int:
        """GaussianModel::increasePcd (src/gaussian_model.cpp:443-520): new anchors at the voxel centres of `points` (N,3)
        -- unique among themselves, NOT checked against the existing anchors, like the reference -- with zero offsets and
        features, scales from the mean squared distance to the 3 nearest new voxels, zero counters and zero Adam moments
        (densificationPostfix).  Returns the number of anchors added."""
        from .neural_gaussians import anchors_from_points
        if points.numel() == 0:
            return 0
        m = self.model
        anchor, scaling = anchors_from_points(points.to(m.device, torch.float32), self.p.voxel_size)
        n_new = anchor.shape[0]
        A0, no = m.A, m.dims.n_offsets
        self._append(anchor, torch.zeros((n_new, m.dims.feat_dim), dtype=torch.float32, device=m.device), 0.0, scaling)
        self._stats["offset_denom"][A0 * no:m.A * no] = 0
        self._stats["offset_gradient_accum"][A0 * no:m.A * no] = 0
        if getattr(m, "coarse", None) is not None:        # :517-518: increasePcd -> increasePcdCoarse on the same points
            m.coarse.increase_pcd(points)
        return n_new

    def anchor_growing(self, grads: torch.Tensor, threshold: float, offset_mask: torch.Tensor, rands: List[torch.Tensor]):
        """:1559-1699.  grads (A_init*no,), offset_mask (A_init*no,) bool, rands[i] (A_init*no,) in [0,1)."""
        m, p = self.model, self.p
        no = m.dims.n_offsets
        A_init = m.A
        mask_u8 = offset_mask.to(torch.uint8).contiguous()
        grads = grads.contiguous()
        n_new_dev = torch.zeros(1, dtype=torch.int32, device=m.device)
        for i in range(p.update_depth):
            cur_threshold = threshold * (math.floor(p.update_hierachy_factor / 2) ** i)
            size_factor = math.floor(p.update_init_factor / (p.update_hierachy_factor ** i))
            cur_size = float(torch.tensor(p.voxel_size * size_factor, dtype=torch.float32))
            if m.A == A_init and i > 0:
                continue                                    # :1573-1577
            max_new = A_init * no
            temp = torch.empty(self._lib.segs_anchor_growing_temp_bytes(m.A, A_init * no), dtype=torch.uint8, device=m.device)
            new_anchor = torch.empty((max_new, 3), dtype=torch.float32, device=m.device)
            new_feat = torch.empty((max_new, m.dims.feat_dim), dtype=torch.float32, device=m.device)
            st = self._lib.segs_anchor_growing_level(
                m.A, A_init, no, m.dims.feat_dim, _p(m.param("anchor")), _p(m.param("offset")), _p(m.param("scaling")),
                _p(m.param("anchor_feat")), _p(grads), _p(mask_u8), _p(rands[i].contiguous()), float(cur_threshold),
                float(0.5 ** (i + 1)), cur_size, max_new, _p(new_anchor), _p(new_feat), _p(n_new_dev), _p(temp), self._stream())
            _capi.check(st, "segs_anchor_growing_level")
            n_new = int(n_new_dev.item())
            if n_new > 0:
                self._append(new_anchor[:n_new], new_feat[:n_new], cur_size)

    def adjust_anchor(self, check_interval: Optional[int] = None, success_threshold: Optional[float] = None,
                      grad_threshold: Optional[float] = None, min_opacity: Optional[float] = None,
                      rands: Optional[List[torch.Tensor]] = None, generator: Optional[torch.Generator] = None,
                      views_per_iteration: int = 1) -> torch.Tensor:
        """:1701-1762.  `rands` (one tensor of A*no uniforms per level) stands for torch::rand_like (:1568); drawn from
        `generator` (shared seed on every rank) when absent.  Returns the prune mask.
        `views_per_iteration`: keyframe-parallel training accumulates N keyframes per iteration, so the "seen in more than
        this fraction of the window" thresholds (check_interval * success_threshold) count views, not iterations."""
        m, p = self.model, self.p
        no = m.dims.n_offsets
        check_interval = (p.update_interval if check_interval is None else check_interval) * int(views_per_iteration)
        success_threshold = p.success_threshold if success_threshold is None else success_threshold
        grad_threshold = p.densify_grad_threshold if grad_threshold is None else grad_threshold
        min_opacity = p.min_opacity if min_opacity is None else min_opacity
        A_init = m.A
        if rands is None:
            rands = [torch.rand(A_init * no, generator=generator, device=m.device if generator is None else generator.device)
                     .to(m.device) for _ in range(p.update_depth)]
        accum, denom = self.stat("offset_gradient_accum"), self.stat("offset_denom")
        grads = accum / denom
        grads[grads.isnan()] = 0.0
        grads_norm = torch.linalg.norm(grads, dim=-1)
        offset_mask = (denom > check_interval * success_threshold * 0.5).squeeze(1)
        self.anchor_growing(grads_norm, grad_threshold, offset_mask, rands)
        # counters of the offsets that were eligible restart; rows of the new anchors start at zero (:1714-1724)
        s = self._stats
        s["offset_denom"][:A_init * no][offset_mask] = 0
        s["offset_gradient_accum"][:A_init * no][offset_mask] = 0
        s["offset_denom"][A_init * no:m.A * no] = 0
        s["offset_gradient_accum"][A_init * no:m.A * no] = 0
        A = m.A
        prune_mask = (s["opacity_accum"][:A] < min_opacity * s["anchor_demon"][:A])
        anchors_mask = s["anchor_demon"][:A] > check_interval * success_threshold
        prune_mask = prune_mask & anchors_mask
        s["opacity_accum"][:A][anchors_mask] = 0.0          # :1738-1748
        s["anchor_demon"][:A][anchors_mask] = 0.0
        if A > 0:
            self.prune_anchor(prune_mask)
        return prune_mask

    def prune_anchor(self, mask: torch.Tensor):
        """:1505-1558 plus the row filtering of the counters (:1730-1754): stable compaction of every per-anchor row."""
        m = self.model
        no = m.dims.n_offsets
        A = m.A
        keep = torch.nonzero(~mask).squeeze(1)
        A1 = int(keep.numel())
        for bucket in (m.params, m.exp_avg, m.exp_avg_sq, m.grads):
            for name in m.widths:
                v = m._view(bucket, name, rows=A)
                v[:A1] = v[keep]
        m.rotation[:A1] = m.rotation[:A][keep]
        m.opacity[:A1] = m.opacity[:A][keep]
        s = self._stats
        for k in ("opacity_accum", "anchor_demon"):
            s[k][:A1] = s[k][:A][keep]
        for k in ("offset_gradient_accum", "offset_denom"):
            s[k][:A1 * no] = s[k][:A * no].view(A, no)[keep].reshape(-1)
        m.A = A1
        sc = m.param("scaling")
        sc[:, 3:] = torch.clamp(sc[:, 3:], max=0.05)        # :1525-1532
