"""CPU restatement of the reference's anchor statistics and densification.  TEST ONLY.

Follows GaussianModel::training_statis / anchor_growing / adjust_anchor / prune_anchor,
/root/reference src/gaussian_model.cpp:1459-1503, 1559-1699, 1701-1762, 1505-1558, op by op with the same torch tensor
operations (the reference calls them through LibTorch; `scatter_max` of torch_scatter 2.1.2 is restated with
Tensor.scatter_reduce(amax)).  The random keep mask (`torch::rand_like`, :1568) is an INPUT here: the reference's CUDA
generator stream cannot be reproduced, parity is defined for equal random numbers.  Parity is UNPINNED against the
reference itself (no fixture exists).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch


@dataclass
class DensifyState:
    """The tensors adjust_anchor touches.  Adam state is carried as exp_avg / exp_avg_sq dicts keyed like `params`."""
    params: Dict[str, torch.Tensor]          # anchor (A,3), offset (A,no,3), anchor_feat (A,32), opacity (A,1), scaling (A,6), rotation (A,4)
    exp_avg: Dict[str, torch.Tensor]
    exp_avg_sq: Dict[str, torch.Tensor]
    opacity_accum: torch.Tensor              # (A,1)
    anchor_demon: torch.Tensor               # (A,1)
    offset_gradient_accum: torch.Tensor      # (A*no,1)
    offset_denom: torch.Tensor               # (A*no,1)
    n_offsets: int = 10
    feat_dim: int = 32
    voxel_size: float = 0.001
    update_depth: int = 3
    update_init_factor: int = 16
    update_hierachy_factor: int = 4


GROUPS = ("anchor", "offset", "anchor_feat", "opacity", "scaling", "rotation")


def training_statis(st: DensifyState, viewspace_grad, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    """:1459-1503.  viewspace_grad (P,3) = grad of screenspace points of the P compacted Gaussians, opacity = neural_opacity
    (Av*no,1), update_filter (P,) = radii > 0, offset_selection_mask (Av*no,) = mask, anchor_visible_mask (A,)."""
    temp_opacity = opacity.clone().view(-1).detach()
    temp_opacity = torch.where(temp_opacity < 0, torch.zeros_like(temp_opacity), temp_opacity)
    temp_opacity = temp_opacity.view(-1, st.n_offsets)
    st.opacity_accum[anchor_visible_mask] = st.opacity_accum[anchor_visible_mask] + temp_opacity.sum(1, keepdim=True)
    st.anchor_demon[anchor_visible_mask] = st.anchor_demon[anchor_visible_mask] + 1
    avm = anchor_visible_mask.unsqueeze(1).repeat(1, st.n_offsets).view(-1)
    combined_mask = torch.zeros_like(st.offset_gradient_accum, dtype=torch.bool).squeeze(1)
    combined_mask[avm] = offset_selection_mask
    temp_mask = combined_mask.clone()
    combined_mask[temp_mask] = update_filter
    grad_norm = torch.linalg.norm(viewspace_grad[update_filter][:, :2], dim=-1, keepdim=True)
    st.offset_gradient_accum[combined_mask] = st.offset_gradient_accum[combined_mask] + grad_norm
    st.offset_denom[combined_mask] = st.offset_denom[combined_mask] + 1


def _extend(st: DensifyState, new: Dict[str, torch.Tensor]):
    for g in GROUPS:
        st.params[g] = torch.cat([st.params[g], new[g]], dim=0)
        if g in st.exp_avg:
            st.exp_avg[g] = torch.cat([st.exp_avg[g], torch.zeros_like(new[g])], dim=0)
            st.exp_avg_sq[g] = torch.cat([st.exp_avg_sq[g], torch.zeros_like(new[g])], dim=0)


def anchor_growing(st: DensifyState, grads, threshold: float, offset_mask, rands: List[torch.Tensor]):
    """:1559-1699.  rands[i] stands for torch::rand_like at level i (shape of `grads`)."""
    no = st.n_offsets
    init_length = st.params["anchor"].shape[0] * no
    for i in range(st.update_depth):
        cur_threshold = threshold * (math.floor(st.update_hierachy_factor / 2) ** i)
        candidate_mask = (grads >= cur_threshold) & offset_mask
        rand_mask = rands[i] > (0.5 ** (i + 1))
        candidate_mask = candidate_mask & rand_mask
        length_inc = st.params["anchor"].shape[0] * no - init_length
        if length_inc == 0:
            if i > 0:
                continue
        else:
            candidate_mask = torch.cat([candidate_mask, torch.zeros(length_inc, dtype=torch.bool)], dim=0)
        anchor = st.params["anchor"]
        all_xyz = anchor.unsqueeze(1) + st.params["offset"] * torch.exp(st.params["scaling"])[:, :3].unsqueeze(1)
        size_factor = math.floor(st.update_init_factor / (st.update_hierachy_factor ** i))
        cur_size = st.voxel_size * size_factor
        cur_size = float(torch.tensor(cur_size, dtype=torch.float32))      # `float cur_size` (:1587)
        grid_coords = torch.round(anchor / cur_size).to(torch.int32)
        selected_xyz = all_xyz.view(-1, 3)[candidate_mask]
        selected_grid_coords = torch.round(selected_xyz / cur_size).to(torch.int32)
        uniq, inverse = torch.unique(selected_grid_coords, dim=0, sorted=True, return_inverse=True)
        if uniq.shape[0] > 0:
            # (unique.unsqueeze(1) == grid_coords).all(-1).any(-1), chunked in the reference (:1601-1615)
            ukey = uniq.to(torch.int64)
            gkey = grid_coords.to(torch.int64)
            pack = lambda t: ((t[:, 0] + (1 << 20)) << 42) | ((t[:, 1] + (1 << 20)) << 21) | (t[:, 2] + (1 << 20))  # noqa: E731
            remove_duplicates = torch.isin(pack(ukey), pack(gkey))
        else:
            remove_duplicates = torch.zeros(0, dtype=torch.bool)
        remove_duplicates = ~remove_duplicates
        candidate_anchor = uniq[remove_duplicates] * cur_size
        if candidate_anchor.shape[0] > 0:
            n_new = candidate_anchor.shape[0]
            new_scaling = torch.log(torch.ones_like(candidate_anchor).repeat(1, 2).float() * cur_size)
            new_rotation = torch.zeros(n_new, 4)
            new_rotation[:, 0] = 1.0
            x = 0.1 * torch.ones(n_new, 1)
            new_opacities = torch.log(x / (1 - x))                             # general_utils::inverse_sigmoid
            new_feat = st.params["anchor_feat"].unsqueeze(1).repeat(1, no, 1).view(-1, st.feat_dim)[candidate_mask]
            idx = inverse.unsqueeze(1).expand(-1, new_feat.shape[1])
            smax = torch.full((uniq.shape[0], new_feat.shape[1]), -float("inf")).scatter_reduce(0, idx, new_feat, "amax", include_self=True)
            new_feat = smax[remove_duplicates]
            new_offsets = torch.zeros_like(candidate_anchor).unsqueeze(1).repeat(1, no, 1).float()
            st.anchor_demon = torch.cat([st.anchor_demon, torch.zeros(n_new, 1)], dim=0)
            st.opacity_accum = torch.cat([st.opacity_accum, torch.zeros(n_new, 1)], dim=0)
            _extend(st, {"anchor": candidate_anchor.float(), "offset": new_offsets, "anchor_feat": new_feat,
                         "opacity": new_opacities, "scaling": new_scaling, "rotation": new_rotation})


def prune_anchor(st: DensifyState, mask):
    """:1505-1558."""
    valid = ~mask
    for g in GROUPS:
        st.params[g] = st.params[g][valid]
        if g in st.exp_avg:
            st.exp_avg[g] = st.exp_avg[g][valid].clone()
            st.exp_avg_sq[g] = st.exp_avg_sq[g][valid].clone()
        if g == "scaling":
            st.params[g][:, 3:] = torch.clamp(st.params[g][:, 3:], max=0.05)


def adjust_anchor(st: DensifyState, check_interval: int, success_threshold: float, grad_threshold: float, min_opacity: float,
                  rands: List[torch.Tensor]):
    """:1701-1762."""
    no = st.n_offsets
    grads = st.offset_gradient_accum / st.offset_denom
    grads[grads.isnan()] = 0.0
    grads_norm = torch.linalg.norm(grads, dim=-1)
    offset_mask = (st.offset_denom > check_interval * success_threshold * 0.5).squeeze(1)
    anchor_growing(st, grads_norm, grad_threshold, offset_mask, rands)
    st.offset_denom[offset_mask] = 0
    A = st.params["anchor"].shape[0]
    st.offset_denom = torch.cat([st.offset_denom, torch.zeros(A * no - st.offset_denom.shape[0], 1)], dim=0)
    st.offset_gradient_accum[offset_mask] = 0
    st.offset_gradient_accum = torch.cat([st.offset_gradient_accum, torch.zeros(A * no - st.offset_gradient_accum.shape[0], 1)], dim=0)
    prune_mask = (st.opacity_accum < min_opacity * st.anchor_demon).squeeze(1)
    anchors_mask = (st.anchor_demon > check_interval * success_threshold).squeeze(1)
    prune_mask = prune_mask & anchors_mask
    st.offset_denom = st.offset_denom.view(-1, no)[~prune_mask].view(-1, 1)
    st.offset_gradient_accum = st.offset_gradient_accum.view(-1, no)[~prune_mask].view(-1, 1)
    if int(anchors_mask.sum()) > 0:
        st.opacity_accum[anchors_mask] = 0.0
        st.anchor_demon[anchors_mask] = 0.0
    st.opacity_accum = st.opacity_accum[~prune_mask]
    st.anchor_demon = st.anchor_demon[~prune_mask]
    if prune_mask.shape[0] > 0:
        prune_anchor(st, prune_mask)
    return prune_mask
