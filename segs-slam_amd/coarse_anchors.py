"""The second, "coarse" anchor set of Model.use_coarse_anchor = 1 (the two offline configurations cfg/colmap/*.yaml set it).

Reference: GaussianModel's constructor (src/gaussian_model.cpp:102-152: five more Sequential MLPs with the `_coarse`
dimensions), createCoarseAnchorFromPcd (:288-325), increasePcdCoarse + densificationPostfixCoarse (:383-441, 1855-1899), the
eleven extra optimizer groups of trainingSetup (:686-787) and their learning-rate schedule (:917-982).

What the reference DOES with this set: it creates it next to the fine one, appends to it whenever the fine set takes new points
(increasePcd -> increasePcdCoarse), registers it with the optimizer -- and nothing else.  The live renderer
(src/gaussian_renderer.cpp) reads only the fine tensors, so no coarse tensor ever receives a gradient; torch::optim::Adam skips a
parameter whose gradient is undefined (no state entry is ever made for it), and savePly / save_mlp_checkpoints do not write it.
The set is an inert payload of the training state, and that is what this module keeps: the tensors with the reference's shapes,
initial values and growth, the names and learning rates of their optimizer groups, no kernel.  (Only the stale
`gaussian_renderer copy.cpp` evaluates the coarse MLPs; it is not compiled by the reference's build.)

Two quirks of the reference are kept on purpose:
  * createCoarseAnchorFromPcd rounds at `coarse_voxel_size` but places the anchors at `unique * voxel_size` (the FINE size, :290);
  * increasePcdCoarse rounds AND places at the fine `voxel_size` (:385-386), and both functions size offsets and features with the
    FINE n_offsets / feat_dim (:292-297), whatever Model.n_offsets_coarse / feat_dim_coarse say (those size the MLPs only)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch


@dataclass
class CoarseParams:
    """Model.*_coarse and Optimization.*_coarse of the configuration file (src/gaussian_mapper.cpp:433-491)."""
    feat_dim_coarse: int = 32
    n_offsets_coarse: int = 10
    coarse_voxel_size: float = 0.2
    appearance_dim_coarse: int = 32
    anchor_lr_init_coarse: float = 0.0
    anchor_lr_final_coarse: float = 0.0
    anchor_lr_delay_mult_coarse: float = 0.01
    anchor_lr_max_steps_coarse: int = 30000
    feature_lr_coarse: float = 0.0075
    opacity_lr_coarse: float = 0.02
    scaling_lr_coarse: float = 0.007
    rotation_lr_coarse: float = 0.002
    offset_lr_init_coarse: float = 0.01
    offset_lr_final_coarse: float = 0.0001
    offset_lr_delay_mult_coarse: float = 0.01
    offset_lr_max_steps_coarse: int = 30000
    mlp_opacity_lr_init_coarse: float = 0.002
    mlp_opacity_lr_final_coarse: float = 0.00002
    mlp_opacity_lr_delay_mult_coarse: float = 0.01
    mlp_opacity_lr_max_steps_coarse: int = 30000
    mlp_cov_lr_init_coarse: float = 0.004
    mlp_cov_lr_final_coarse: float = 0.004
    mlp_cov_lr_delay_mult_coarse: float = 0.01
    mlp_cov_lr_max_steps_coarse: int = 30000
    mlp_color_lr_init_coarse: float = 0.008
    mlp_color_lr_final_coarse: float = 0.00005
    mlp_color_lr_delay_mult_coarse: float = 0.01
    mlp_color_lr_max_steps_coarse: int = 30000
    mlp_featurebank_lr_init_coarse: float = 0.01
    mlp_featurebank_lr_final_coarse: float = 0.00001
    mlp_featurebank_lr_delay_mult_coarse: float = 0.01
    mlp_featurebank_lr_max_steps_coarse: int = 30000
    appearance_lr_init_coarse: float = 0.05
    appearance_lr_final_coarse: float = 0.0005
    appearance_lr_delay_mult_coarse: float = 0.01
    appearance_lr_max_steps_coarse: int = 30000


def coarse_mlp_shapes(dims, cp: CoarseParams) -> Dict[str, Tuple[int, ...]]:
    """nn::Linear weights / biases of the five coarse Sequentials (src/gaussian_model.cpp:112-148), in construction order."""
    f = cp.feat_dim_coarse
    od, cd, kd = int(dims.add_opacity_dist), int(dims.add_cov_dist), int(dims.add_color_dist)
    shapes = {
        "mlp_opacity_c.0": (f, f + 3 + od), "mlp_opacity_c.2": (cp.n_offsets_coarse, f),
        "mlp_cov_c.0": (f, f + 3 + cd), "mlp_cov_c.2": (7 * cp.n_offsets_coarse, f),
        "mlp_color_c.0": (f, f + 3 + kd + cp.appearance_dim_coarse), "mlp_color_c.2": (3 * cp.n_offsets_coarse, f),
        "mlp_apperance_c.0": (cp.appearance_dim_coarse, 7),
    }
    if dims.use_feat_bank:
        shapes.update({"mlp_feature_bank_c.0": (f, 4), "mlp_feature_bank_c.2": (3, f)})
    out = {}
    for k, (o, i) in shapes.items():
        out[k + ".weight"] = (o, i)
        out[k + ".bias"] = (o,)
    return out


def expon_lr(step: int, lr_init: float, lr_final: float, lr_delay_mult: float, max_steps: int) -> float:
    """GaussianModel::getExponLrFunc (src/gaussian_model.cpp:1393-1409) as updateLearningRate calls it for the coarse groups
    (:920-950): lr_delay_steps = 0, so the delay rate is 1 and lr_delay_mult plays no part -- the fine groups' function."""
    from .gaussian_trainer import expon_lr as fine
    return fine(step, lr_init, lr_final, max_steps)


@dataclass
class CoarseAnchors:
    params: CoarseParams
    voxel_size: float
    dims: object                                   # the FINE ModelDims (n_offsets, feat_dim size the rows)
    device: torch.device
    anchor: torch.Tensor = None                    # (Nc, 3)
    offset: torch.Tensor = None                    # (Nc, n_offsets, 3)
    anchor_feat: torch.Tensor = None               # (Nc, feat_dim)
    scaling: torch.Tensor = None                   # (Nc, 6) log-scales
    rotation: torch.Tensor = None                  # (Nc, 4), requires no gradient in the reference
    opacity: torch.Tensor = None                   # (Nc, 1), likewise
    max_radii2D: torch.Tensor = None
    mlp: Dict[str, torch.Tensor] = field(default_factory=dict)
    spatial_lr_scale: float = 1.0

    @property
    def n(self) -> int:
        return 0 if self.anchor is None else int(self.anchor.shape[0])

    # ---- createCoarseAnchorFromPcd (:288-325)
    @staticmethod
    def create_from_pcd(points: torch.Tensor, params: CoarseParams, voxel_size: float, dims, device, mlp_seed: int = 0) -> "CoarseAnchors":
        from .neural_gaussians import anchors_from_points
        device = torch.device(device)
        c = CoarseAnchors(params, float(voxel_size), dims, device)
        anchor, scaling = anchors_from_points(points.to(device, torch.float32), params.coarse_voxel_size, place_size=voxel_size)
        c._set_rows(anchor, scaling)
        g = torch.Generator().manual_seed(0xC0A5 + mlp_seed)
        for name, shape in coarse_mlp_shapes(dims, params).items():        # torch::nn::Linear's default range, like init_mlps
            fan_in = shape[1] if len(shape) == 2 else coarse_mlp_shapes(dims, params)[name.replace(".bias", ".weight")][1]
            c.mlp[name] = ((torch.rand(*shape, generator=g) * 2 - 1) / math.sqrt(fan_in)).to(device)
        return c

    def _rows_for(self, anchor: torch.Tensor, scaling: torch.Tensor):
        n, z = anchor.shape[0], lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)  # noqa: E731
        rot = z(n, 4)
        rot[:, 0] = 1.0
        x = 0.1 * torch.ones((n, 1), dtype=torch.float32, device=self.device)
        return anchor, z(n, self.dims.n_offsets, 3), z(n, self.dims.feat_dim), scaling, rot, torch.log(x / (1 - x))

    def _set_rows(self, anchor, scaling):
        self.anchor, self.offset, self.anchor_feat, self.scaling, self.rotation, self.opacity = self._rows_for(anchor, scaling)
        self.max_radii2D = torch.zeros(self.n, dtype=torch.float32, device=self.device)

    # ---- increasePcdCoarse + densificationPostfixCoarse (:383-441, 1855-1899)
    def increase_pcd(self, points: torch.Tensor) -> int:
        """Appends the voxel centres of `points` -- rounded and placed at the FINE voxel size, unique among themselves only.  No
        Adam moments to extend: these parameters never had a gradient, so the reference's `state.find(key)` (:1868) finds nothing
        and takes the branch that only concatenates."""
        from .neural_gaussians import anchors_from_points
        if points.numel() == 0:
            return 0
        anchor, scaling = anchors_from_points(points.to(self.device, torch.float32), self.voxel_size)
        new = self._rows_for(anchor, scaling)
        old = (self.anchor, self.offset, self.anchor_feat, self.scaling, self.rotation, self.opacity)
        self.anchor, self.offset, self.anchor_feat, self.scaling, self.rotation, self.opacity = (torch.cat([a, b], 0) for a, b in zip(old, new))
        self.max_radii2D = torch.zeros(self.n, dtype=torch.float32, device=self.device)      # :1899
        return int(anchor.shape[0])

    # ---- trainingSetup's coarse groups and updateLearningRate (:686-787, 917-982)
    def optimizer_groups(self, iteration: int) -> List[Tuple[str, int, float]]:
        """(name, number of parameters, learning rate at `iteration`) of the coarse parameter groups in the reference's order:
        anchor, offset, anchor_feat, opacity, scaling, rotation, then the MLPs (feature bank only with Model.use_feat_bank,
        appearance only with appearance_dim > 0, as the three branches of :686-787 do)."""
        p, s = self.params, self.spatial_lr_scale
        e = lambda k: expon_lr(iteration, getattr(p, k + "_lr_init_coarse"), getattr(p, k + "_lr_final_coarse"),  # noqa: E731
                               getattr(p, k + "_lr_delay_mult_coarse"), getattr(p, k + "_lr_max_steps_coarse"))
        anchor_lr = expon_lr(iteration, p.anchor_lr_init_coarse * s, p.anchor_lr_final_coarse * s, p.anchor_lr_delay_mult_coarse,
                             p.anchor_lr_max_steps_coarse)
        offset_lr = expon_lr(iteration, p.offset_lr_init_coarse * s, p.offset_lr_final_coarse * s, p.offset_lr_delay_mult_coarse,
                             p.offset_lr_max_steps_coarse)
        nmlp = lambda prefix: sum(t.numel() for k, t in self.mlp.items() if k.startswith(prefix + "."))  # noqa: E731
        groups = [("anchor_c", self.anchor.numel(), anchor_lr), ("offset_c", self.offset.numel(), offset_lr),
                  ("anchor_feat_c", self.anchor_feat.numel(), p.feature_lr_coarse), ("opacity_c", self.opacity.numel(), p.opacity_lr_coarse),
                  ("scaling_c", self.scaling.numel(), p.scaling_lr_coarse), ("rotation_c", self.rotation.numel(), p.rotation_lr_coarse),
                  ("mlp_opacity_c", nmlp("mlp_opacity_c"), e("mlp_opacity")), ("mlp_cov_c", nmlp("mlp_cov_c"), e("mlp_cov")),
                  ("mlp_color_c", nmlp("mlp_color_c"), e("mlp_color"))]
        if self.dims.appearance_dim > 0:
            groups.append(("mlp_apperance_c", nmlp("mlp_apperance_c"), e("appearance")))
        if self.dims.use_feat_bank:
            groups.append(("mlp_feature_bank_c", nmlp("mlp_feature_bank_c"), e("mlp_featurebank")))
        return groups
