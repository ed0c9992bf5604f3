"""Host-side mirror of the reference's tensor-typed entry points (include/rasterize_points.h:18-102,
src/rasterize_points.cu) on top of the C ABI (include/segs_raster.h).

Same names, argument order, return tuples, "absent = 0-element tensor" convention and error behaviour
as the reference, so the parity tests read like calls into the reference:

  RasterizeGaussiansCUDA          src/rasterize_points.cu:36-114
  RasterizeGaussiansBackwardCUDA  src/rasterize_points.cu:116-193
  markVisible                     src/rasterize_points.cu:195-214
  RasterizeGaussiansfilterCUDA    src/rasterize_points.cu:216-280
  RasterizeGaussiansprojectCUDA   src/rasterize_points.cu:282-363

torch is plumbing only here (device memory + current stream); every kernel is in csrc/.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi

NUM_CHANNELS = 3  # cuda_rasterizer/config.h:15


def _ptr(t: torch.Tensor):
    """data_ptr of a contiguous tensor; a 0-element tensor means 'absent' -> NULL (rasterize_points.cu:95-105)."""
    if t is None or t.numel() == 0:
        return None
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: this build has no CPU raster path "
                           "(neither has the reference, src/rasterize_points.cu:71-75)")


class _ResizableBuffer:
    """resizeFunctional (src/rasterize_points.cu:28-34): a byte tensor grown by the allocator callback.

    The callback object is made per call (`callback()`) and NOT kept on the buffer: a ctypes callback stored on the object whose
    bound method it wraps is a reference cycle, and the scratch tensor -- 0.8 GB per call at 3 M Gaussians -- then lives until the
    cyclic garbage collector runs instead of until the caller drops it: the tensor library's caching allocator kept answering
    the next calls with fresh device allocations (5.5 GB reserved after seven calls, a device allocation right after the host
    synchronisation of most forwards; round 4 found it through a 5x slower benchmark loop)."""

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)

    def callback(self):
        def _alloc(_ctx, nbytes, box=self):
            box.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=box.device)
            return box.tensor.data_ptr()
        return _capi.ALLOC_FN(_alloc)


def RasterizeGaussiansCUDA(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                           viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                           prefiltered):
    """-> (num_rendered, out_color(3,H,W), radii(P) int32, geomBuffer, binningBuffer, imgBuffer)."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # AT_ERROR, rasterize_points.cu:57-59
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    # the reference zero-fills both (:68-69); that only shows for P == 0 (:81) -- otherwise every pixel and radius is written
    mk = torch.empty if P != 0 else torch.zeros
    out_color = mk((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    radii = mk((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _ResizableBuffer(dev), _ResizableBuffer(dev), _ResizableBuffer(dev)
    rendered = 0
    if P != 0:  # rasterize_points.cu:81 (P == 0 leaves the zero image, not the background)
        M = int(sh.size(1)) if sh.numel() != 0 else 0
        keep = [_f32c(t) for t in (background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp,
                                   viewmatrix, projmatrix, campos)]
        bg, m3, shc, col, opa, sca, rot, cov, view, proj, cam = keep
        n = C.c_int(0)
        gcb, bcb, icb = geom.callback(), binning.callback(), img.callback()
        with torch.cuda.device(dev):
            st = _capi.lib().segs_rasterize_forward(
                gcb, None, bcb, None, icb, None, P, int(degree), M, _ptr(bg), W, H, _ptr(m3), _ptr(shc),
                _ptr(col), _ptr(opa), _ptr(sca), float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(view), _ptr(proj),
                _ptr(cam), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), _ptr(out_color), _ptr(radii),
                _stream(dev), C.byref(n))
        _capi.check(st, "segs_rasterize_forward")
        rendered = int(n.value)
    return rendered, out_color, radii, geom.tensor, binning.tensor, img.tensor


def RasterizeGaussiansBackwardCUDA(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                   viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                   geomBuffer, R, binningBuffer, imageBuffer):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    opts = dict(dtype=torch.float32, device=dev)
    # The kernels write every row, so empty() replaces the reference's nine torch::zeros fills (:149-157).
    dL_dmeans3D = torch.empty((P, 3), **opts)
    dL_dmeans2D = torch.empty((P, 3), **opts)
    dL_dcolors = torch.empty((P, NUM_CHANNELS), **opts)
    # dL_dconic (P,2,2) is an internal product of the reference's backward (:153), never returned: not materialised here
    dL_dopacity = torch.empty((P, 1), **opts)
    dL_dcov3D = torch.empty((P, 6), **opts)
    dL_dsh = torch.zeros((P, M, 3), **opts)
    has_sr = scales.numel() != 0
    dL_dscales = torch.empty((P, 3), **opts) if has_sr else torch.zeros((P, 3), **opts)
    dL_drotations = torch.empty((P, 4), **opts) if has_sr else torch.zeros((P, 4), **opts)
    if P != 0:  # rasterize_points.cu:159
        keep = [_f32c(t) for t in (background, means3D, sh, colors, scales, rotations, cov3D_precomp, viewmatrix,
                                   projmatrix, campos, dL_dout_color)]
        bg, m3, shc, col, sca, rot, cov, view, proj, cam, dL = keep
        rad = radii.contiguous()
        with torch.cuda.device(dev):
            st = _capi.lib().segs_rasterize_backward(
                P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(m3), _ptr(shc), _ptr(col), _ptr(sca),
                float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(view), _ptr(proj), _ptr(cam), float(tan_fovx),
                float(tan_fovy), _ptr(rad), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(dL),
                _ptr(dL_dmeans2D), None, _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dmeans3D),
                _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales) if has_sr else None,
                _ptr(dL_drotations) if has_sr else None, _stream(dev))
        _capi.check(st, "segs_rasterize_backward")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def markVisible(means3D, viewmatrix, projmatrix):
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m3, view, proj = _f32c(means3D), _f32c(viewmatrix), _f32c(projmatrix)
        with torch.cuda.device(dev):
            st = _capi.lib().segs_mark_visible(P, _ptr(m3), _ptr(view), _ptr(proj), C.c_void_p(present.data_ptr()), _stream(dev))
        _capi.check(st, "segs_mark_visible")
    return present


def RasterizeGaussiansfilterCUDA(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                                 tan_fovx, tan_fovy, image_height, image_width, prefiltered, debug=False):
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    if P != 0:
        m3, sca, rot, cov, view, proj = (_f32c(t) for t in (means3D, scales, rotations, cov3D_precomp, viewmatrix, projmatrix))
        with torch.cuda.device(dev):
            st = _capi.lib().segs_visible_filter(P, 0, int(image_width), int(image_height), _ptr(m3), _ptr(sca),
                                                 float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(view), _ptr(proj),
                                                 float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), _ptr(radii),
                                                 _stream(dev))
        _capi.check(st, "segs_visible_filter")
    return radii


def RasterizeGaussiansprojectCUDA(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                  viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                                  campos, prefiltered):
    """-> (points_image (P,2), radii (P), out_color (P,3))  (src/rasterize_points.cu:361)."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    out_color = torch.zeros((P, NUM_CHANNELS), dtype=torch.float32, device=dev)
    points_image = torch.zeros((P, 2), dtype=torch.float32, device=dev)
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    if P != 0:
        M = int(sh.size(1)) if sh.numel() != 0 else 0
        keep = [_f32c(t) for t in (means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos)]
        m3, shc, col, opa, sca, rot, cov, view, proj, cam = keep
        with torch.cuda.device(dev):
            st = _capi.lib().segs_project2_image(P, int(degree), M, int(image_width), int(image_height), _ptr(m3), _ptr(shc),
                                                 _ptr(col), _ptr(opa), _ptr(sca), float(scale_modifier), _ptr(rot), _ptr(cov),
                                                 _ptr(view), _ptr(proj), _ptr(cam), float(tan_fovx), float(tan_fovy),
                                                 int(bool(prefiltered)), _ptr(out_color), _ptr(points_image), _ptr(radii),
                                                 _stream(dev))
        _capi.check(st, "segs_project2_image")
    return points_image, radii, out_color


# ---- parity-test support (not in the reference): expose the private scratch as the reference's state arrays
def debug_state(P, W, H, R, radii, geomBuffer, binningBuffer, imageBuffer):
    dev = geomBuffer.device
    l = _capi.lib()
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(
        means2D=torch.zeros((P, 2), dtype=torch.float32, device=dev),
        conic_opacity=torch.zeros((P, 4), dtype=torch.float32, device=dev),
        depths=torch.zeros((P,), dtype=torch.float32, device=dev),
        tiles_touched=torch.zeros((P,), dtype=torch.int32, device=dev),
        point_offsets=torch.zeros((P,), dtype=torch.int32, device=dev),
        rgb=torch.zeros((P, 3), dtype=torch.float32, device=dev),
        keys=torch.zeros((R,), dtype=torch.int64, device=dev),
        point_list=torch.zeros((R,), dtype=torch.int32, device=dev),
        ranges=torch.zeros((tiles, 2), dtype=torch.int32, device=dev),
        final_T=torch.zeros((H, W), dtype=torch.float32, device=dev),
        n_contrib=torch.zeros((H, W), dtype=torch.int32, device=dev),
    )
    p = lambda k: C.c_void_p(out[k].data_ptr()) if out[k].numel() else None  # noqa: E731
    st = _stream(dev)
    with torch.cuda.device(dev):
        if P:
            _capi.check(l.segs_debug_unpack_geometry(_ptr(geomBuffer), P, _ptr(radii), p("means2D"), p("conic_opacity"),
                                                     p("depths"), p("tiles_touched"), p("point_offsets"), p("rgb"), st),
                        "segs_debug_unpack_geometry")
        if R:
            _capi.check(l.segs_debug_unpack_binning(_ptr(binningBuffer), _ptr(geomBuffer), P, R, W, H, p("keys"), p("point_list"), st),
                        "segs_debug_unpack_binning")
        _capi.check(l.segs_debug_unpack_image(_ptr(imageBuffer), W, H, p("ranges"), p("final_T"), p("n_contrib"), st),
                    "segs_debug_unpack_image")
    return out
