"""Host-side mirror of the point-set helpers (include/segs_points.h) with the reference's names and semantics:

  distCUDA2                                   third_party/simple-knn/spatial.cu:15-26
  transformPoints, scaleAndTransformThenMarkVisiblePoints        src/operate_points.cu:73-143
  reprojectDepthPinhole, monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints  src/stereo_vision.cu:136-213

The reference mutates its tensor arguments through C++ references; here the functions return the new tensors (and
mutate in place where the reference uses index_put_).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi
from .rasterize_points import _f32c, _ptr, _require_gpu, _stream, markVisible


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    _require_gpu(points, "points")
    P = int(points.size(0))
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)
    if P:
        pts = _f32c(points)
        lib = _capi.lib()
        temp = torch.empty(lib.segs_knn_temp_bytes(P), dtype=torch.uint8, device=points.device)
        with torch.cuda.device(points.device):
            _capi.check(lib.segs_knn_mean_dist2(P, _ptr(pts), _ptr(means), _ptr(temp), _stream(points.device)), "segs_knn_mean_dist2")
    return means


def transformPoints(points: torch.Tensor, transformmatrix: torch.Tensor) -> torch.Tensor:
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    _require_gpu(points, "points")
    P = int(points.size(0))
    out = torch.zeros_like(points)
    if P == 0:
        return points
    pts, m = _f32c(points), _f32c(transformmatrix)
    with torch.cuda.device(points.device):
        _capi.check(_capi.lib().segs_transform_points(P, _ptr(pts), _ptr(m), _ptr(out), _stream(points.device)), "segs_transform_points")
    return out


def scaleAndTransformThenMarkVisiblePoints(points, rots, point_not_transformed_mask, point_unstable_mask, transformmatrix,
                                           viewmatrix, projmatrix, num_transformed: int, scale: float = 1.0) -> int:
    """Mutates points / rots / point_not_transformed_mask in place like the reference; returns the updated count."""
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    present = markVisible(points, viewmatrix, projmatrix)
    n = present.size(0)
    if point_not_transformed_mask.size(0) != n or point_unstable_mask.size(0) != n:
        raise RuntimeError("points_mask must have dimensions (num_points)")
    final_mask = torch.logical_and(torch.logical_and(point_not_transformed_mask, point_unstable_mask), present)
    num_transformed += int(final_mask.sum().item())
    P = int(points.size(0))
    if P:
        tp, tr = torch.zeros_like(points), torch.zeros_like(rots)
        pts, rr, m = _f32c(points), _f32c(rots), _f32c(transformmatrix)
        mask_u8 = final_mask.to(torch.uint8)
        with torch.cuda.device(points.device):
            _capi.check(_capi.lib().segs_scale_and_transform_points(P, float(scale), _ptr(pts), _ptr(rr), _ptr(m), _ptr(mask_u8),
                                                                    _ptr(tp), _ptr(tr), _stream(points.device)),
                        "segs_scale_and_transform_points")
        points[final_mask] = tp[final_mask]
        rots[final_mask] = tr[final_mask]
        point_not_transformed_mask[final_mask] = False
    return num_transformed


def reprojectDepthPinhole(depth: torch.Tensor, mask: torch.Tensor, intr, width: int) -> torch.Tensor:
    if depth.dim() != 1:
        raise RuntimeError("points must have dimensions (num_points)")
    _require_gpu(depth, "depth")
    P = int(depth.size(0))
    points = torch.zeros((P, 3), dtype=depth.dtype, device=depth.device)
    if P:
        d, k = _f32c(depth), mask.to(torch.uint8).contiguous()
        with torch.cuda.device(depth.device):
            _capi.check(_capi.lib().segs_reproject_depths_pinhole(P, int(width), *[float(x) for x in intr[:4]], _ptr(d), _ptr(k),
                                                                  _ptr(points), _stream(depth.device)), "segs_reproject_depths_pinhole")
    return points


def monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints(kps_pixel, kps_has3D, kps_point_local, colors,
                                                                       max_pixel_dist: float, intr, width: int):
    if kps_pixel.dim() != 2 or kps_pixel.size(1) != 2:
        raise RuntimeError("kps_pixel must have dimensions (num_points, 2)")
    if kps_has3D.dim() != 1:
        raise RuntimeError("kps_has3D must have dimensions (num_points)")
    if kps_point_local.dim() != 2 or kps_point_local.size(1) != 3:
        raise RuntimeError("kps_point_local must have dimensions (num_points, 3)")
    _require_gpu(kps_pixel, "kps_pixel")
    N = int(kps_pixel.size(0))
    result_pt, result_color = torch.zeros_like(kps_point_local), torch.zeros_like(kps_point_local)
    if N:
        px, h, p3, col = _f32c(kps_pixel), kps_has3D.to(torch.uint8).contiguous(), _f32c(kps_point_local), _f32c(colors)
        with torch.cuda.device(kps_pixel.device):
            _capi.check(_capi.lib().segs_search_neighborhood_depth(N, int(width), *[float(x) for x in intr[:4]], float(max_pixel_dist),
                                                                   _ptr(px), _ptr(h), _ptr(p3), _ptr(col), _ptr(result_pt),
                                                                   _ptr(result_color), _stream(kps_pixel.device)),
                        "segs_search_neighborhood_depth")
        valid = result_pt[:, 2] > 0.0
        result_pt, result_color = result_pt[valid], result_color[valid]
    return result_pt, result_color
