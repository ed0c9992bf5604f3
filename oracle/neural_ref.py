"""CPU restatement of the reference's neural-Gaussian generation (Scaffold-GS anchors -> Gaussians).  TEST ONLY.

Follows GaussianRenderer::generate_neural_gaussians, /root/reference src/gaussian_renderer.cpp:214-334, op by op with
the same tensor operations LibTorch performs there (boolean-mask index, cat, Linear, repeat, split); the MLP stacks are
those of the GaussianModel constructor, src/gaussian_model.cpp:61-98.  Autograd through this function is the oracle of
the fused backward kernel.  Parity is UNPINNED against the reference itself (it holds no fixture for this function,
SURVEY.md 8c); what pins this restatement is that it is the same sequence of library ops.

Only tests/ , __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class NeuralDims:
    feat_dim: int = 32
    n_offsets: int = 10
    appearance_dim: int = 32
    use_feat_bank: bool = True
    add_opacity_dist: bool = False
    add_cov_dist: bool = False
    add_color_dist: bool = False

    def tensor_shapes(self):
        """MLP parameter tensors in the order of the reference's Adam groups (src/gaussian_model.cpp:654-690):
        mlp_opacity, mlp_cov, mlp_color, then mlp_apperance / mlp_feature_bank when enabled.  Names follow
        torch::nn::Sequential's state_dict keys."""
        fd, no = self.feat_dim, self.n_offsets
        out = [
            ("mlp_opacity.0.weight", (fd, fd + 3 + int(self.add_opacity_dist))), ("mlp_opacity.0.bias", (fd,)),
            ("mlp_opacity.2.weight", (no, fd)), ("mlp_opacity.2.bias", (no,)),
            ("mlp_cov.0.weight", (fd, fd + 3 + int(self.add_cov_dist))), ("mlp_cov.0.bias", (fd,)),
            ("mlp_cov.2.weight", (7 * no, fd)), ("mlp_cov.2.bias", (7 * no,)),
            ("mlp_color.0.weight", (fd, fd + 3 + int(self.add_color_dist) + self.appearance_dim)), ("mlp_color.0.bias", (fd,)),
            ("mlp_color.2.weight", (3 * no, fd)), ("mlp_color.2.bias", (3 * no,)),
        ]
        if self.appearance_dim > 0:
            out += [("mlp_apperance.0.weight", (self.appearance_dim, 7)), ("mlp_apperance.0.bias", (self.appearance_dim,))]
        if self.use_feat_bank:
            out += [("mlp_feature_bank.0.weight", (fd, 4)), ("mlp_feature_bank.0.bias", (fd,)),
                    ("mlp_feature_bank.2.weight", (3, fd)), ("mlp_feature_bank.2.bias", (3,))]
        return out


def _mlp(p, name, x):
    h = F.relu(F.linear(x, p[name + ".0.weight"], p[name + ".0.bias"]))
    return F.linear(h, p[name + ".2.weight"], p[name + ".2.bias"])


def generate_neural_gaussians(dims: NeuralDims, anchor, offset, anchor_feat, scaling_log, mlp, camera_center, pose7,
                              visible_mask):
    """anchor (A,3), offset (A,n_offsets,3), anchor_feat (A,feat_dim), scaling_log (A,6) [get_scaling = exp,
    src/gaussian_model.cpp get_scaling], mlp: dict name -> tensor, camera_center (3,), pose7 = (t_xyz, q_wxyz) of the
    keyframe (gaussian_renderer.cpp:258-261), visible_mask (A,) bool.
    Returns (xyz, color, opacity, scaling, rot, neural_opacity, mask) exactly as gaussian_renderer.cpp:333."""
    feat = anchor_feat[visible_mask]                                  # :228
    anc = anchor[visible_mask]                                        # :229
    grid_offsets = offset[visible_mask]                               # :230
    grid_scaling = torch.exp(scaling_log)[visible_mask]               # :231
    ob_view = anc - camera_center                                     # :232
    ob_dist = torch.linalg.norm(ob_view, dim=1, keepdim=True)         # :233
    ob_view = ob_view / ob_dist                                       # :234
    if dims.use_feat_bank:                                            # :236-249
        cat_view = torch.cat([ob_view, ob_dist], dim=1)
        bank_weight = torch.softmax(_mlp(mlp, "mlp_feature_bank", cat_view), dim=1).unsqueeze(1)
        f = feat.unsqueeze(-1)
        f = (f[:, ::4, :1].repeat(1, 4, 1) * bank_weight[:, :, :1]
             + f[:, ::2, :1].repeat(1, 2, 1) * bank_weight[:, :, 1:2]
             + f[:, ::1, :1] * bank_weight[:, :, 2:])
        feat = f.squeeze(-1)
    cat_local_view = torch.cat([feat, ob_view, ob_dist], dim=1)       # :251
    cat_local_view_wodist = torch.cat([feat, ob_view], dim=1)         # :252
    if dims.appearance_dim > 0:                                       # :256-270
        ob_pose = pose7.reshape(1, 7).expand(cat_local_view.shape[0], -1)
        appearance_feat = F.linear(ob_pose, mlp["mlp_apperance.0.weight"], mlp["mlp_apperance.0.bias"])
    x = cat_local_view if dims.add_opacity_dist else cat_local_view_wodist
    neural_opacity = torch.tanh(_mlp(mlp, "mlp_opacity", x)).reshape(-1, 1)   # :273-278
    mask = (neural_opacity > 0.0).view(-1)                            # :279-280
    opacity = neural_opacity[mask]                                    # :282
    x = cat_local_view if dims.add_color_dist else cat_local_view_wodist
    if dims.appearance_dim > 0:
        x = torch.cat([x, appearance_feat], dim=1)
    color = torch.sigmoid(_mlp(mlp, "mlp_color", x)).reshape(anc.shape[0] * dims.n_offsets, 3)   # :285-299
    x = cat_local_view if dims.add_cov_dist else cat_local_view_wodist
    scale_rot = _mlp(mlp, "mlp_cov", x).reshape(anc.shape[0] * dims.n_offsets, 7)                # :301-306
    offsets = grid_offsets.reshape(-1, 3)                             # :308
    concatenated = torch.cat([grid_scaling, anc], dim=-1)             # :309
    concatenated_repeated = concatenated.repeat(1, dims.n_offsets).view(anc.shape[0] * dims.n_offsets, -1)  # :315-316
    concatenated_all = torch.cat([concatenated_repeated, color, scale_rot, offsets], dim=-1)     # :319
    masked = concatenated_all[mask]                                   # :320
    scaling_repeat, repeat_anchor, color, scale_rot, offsets = masked.split([6, 3, 3, 7, 3], dim=-1)  # :321-326
    scaling = scaling_repeat[:, 3:] * torch.sigmoid(scale_rot[:, :3])  # :327-328
    rot = F.normalize(scale_rot[:, 3:7])                              # :329-330
    offsets = offsets * scaling_repeat[:, :3]                         # :331
    xyz = repeat_anchor + offsets                                     # :332
    return xyz, color, opacity, scaling, rot, neural_opacity, mask


def random_model(dims: NeuralDims, A: int, seed: int = 0, dtype=torch.float32):
    """Seeded anchors + MLPs for tests: features ~ N(0, 0.5), torch Linear default-like uniform init."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)  # noqa: E731
    n = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    anchor = torch.stack([(r(A) - 0.5) * 4.0, (r(A) - 0.5) * 3.0, 1.0 + 4.0 * r(A)], dim=1)
    offset = 0.5 * n(A, dims.n_offsets, 3)
    feat = 0.5 * n(A, dims.feat_dim)
    scaling_log = torch.log(0.01 + 0.04 * r(A, 6))
    mlp = {}
    for name, shape in dims.tensor_shapes():
        fan_in = shape[1] if len(shape) == 2 else dims.feat_dim
        bound = 1.0 / (fan_in ** 0.5)
        mlp[name] = ((r(*shape) * 2 - 1) * bound)
    cast = lambda t: t.to(dtype)  # noqa: E731
    return cast(anchor), cast(offset), cast(feat), cast(scaling_log), {k: cast(v) for k, v in mlp.items()}
