"""Host-side mirror of include/gaussian_rasterizer.h:25-151 / src/gaussian_rasterizer.cpp.

GaussianRasterizationSettings, GaussianRasterizerFunction (autograd), rasterizeGaussians and the
GaussianRasterizer module keep the reference's names, argument order and error behaviour.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import rasterize_points as rp


@dataclass
class GaussianRasterizationSettings:
    """include/gaussian_rasterizer.h:25-57 (fields keep the reference's trailing underscore)."""
    image_height_: int
    image_width_: int
    tanfovx_: float
    tanfovy_: float
    bg_: torch.Tensor
    scale_modifier_: float
    viewmatrix_: torch.Tensor
    projmatrix_: torch.Tensor
    sh_degree_: int
    campos_: torch.Tensor
    prefiltered_: bool = False


class GaussianRasterizerFunction(torch.autograd.Function):
    """src/gaussian_rasterizer.cpp:27-154."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = rp.RasterizeGaussiansCUDA(
            rs.bg_, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier_, cov3Ds_precomp,
            rs.viewmatrix_, rs.projmatrix_, rs.tanfovx_, rs.tanfovy_, rs.image_height_, rs.image_width_, sh,
            rs.sh_degree_, rs.campos_, rs.prefiltered_)
        ctx.num_rendered = num_rendered
        ctx.scale_modifier = rs.scale_modifier_
        ctx.tanfovx, ctx.tanfovy = rs.tanfovx_, rs.tanfovy_
        ctx.sh_degree = rs.sh_degree_
        ctx.save_for_backward(rs.bg_, rs.viewmatrix_, rs.projmatrix_, rs.campos_, colors_precomp, means3D, scales,
                              rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii=None):
        (bg, viewmatrix, projmatrix, campos, colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
         geomBuffer, binningBuffer, imgBuffer) = ctx.saved_tensors
        (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
         dL_drotations) = rp.RasterizeGaussiansBackwardCUDA(
            bg, means3D, radii, colors_precomp, scales, rotations, ctx.scale_modifier, cov3Ds_precomp, viewmatrix,
            projmatrix, ctx.tanfovx, ctx.tanfovy, grad_out_color, sh, ctx.sh_degree, campos, geomBuffer,
            ctx.num_rendered, binningBuffer, imgBuffer)
        # gradient order of src/gaussian_rasterizer.cpp:143-153; "absent" inputs (0-element tensors) get None
        g = lambda t, ref: t if ref.numel() != 0 else None  # noqa: E731
        return (dL_dmeans3D, dL_dmeans2D, g(dL_dsh, sh), g(dL_dcolors, colors_precomp), dL_dopacity,
                g(dL_dscales, scales), g(dL_drotations, rotations), g(dL_dcov3D, cov3Ds_precomp), None)


def rasterizeGaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    """include/gaussian_rasterizer.h:79-101."""
    return GaussianRasterizerFunction.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                            cov3Ds_precomp, raster_settings)


class GaussianRasterizer(torch.nn.Module):
    """include/gaussian_rasterizer.h:103-151, src/gaussian_rasterizer.cpp:19-25,156-307."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings_ = raster_settings

    def _absent(self, like: torch.Tensor) -> torch.Tensor:
        return torch.empty(0, dtype=torch.float32, device=like.device)

    def markVisibleGaussians(self, positions):
        with torch.no_grad():
            rs = self.raster_settings_
            return rp.markVisible(positions, rs.viewmatrix_, rs.projmatrix_)

    def _check(self, has_shs, has_colors_precomp, has_scales, has_rotations, has_cov3D_precomp):
        if (not has_shs and not has_colors_precomp) or (has_shs and has_colors_precomp):
            raise RuntimeError("Please provide excatly one of either SHs or precomputed colors!")
        if ((not has_scales or not has_rotations) and not has_cov3D_precomp) or \
                ((has_scales or has_rotations) and has_cov3D_precomp):
            raise RuntimeError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")

    def forward(self, means3D, means2D, opacities, has_shs, has_colors_precomp, has_scales, has_rotations,
                has_cov3D_precomp, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        self._check(has_shs, has_colors_precomp, has_scales, has_rotations, has_cov3D_precomp)
        e = self._absent(means3D)
        shs = shs if has_shs else e
        colors_precomp = colors_precomp if has_colors_precomp else e
        scales = scales if has_scales else e
        rotations = rotations if has_rotations else e
        cov3D_precomp = cov3D_precomp if has_cov3D_precomp else e
        color, radii = rasterizeGaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                          cov3D_precomp, self.raster_settings_)
        return color, radii

    def visible_filter(self, means3D, has_scales, has_rotations, has_cov3D_precomp, scales=None, rotations=None,
                       cov3D_precomp=None):
        rs = self.raster_settings_
        e = self._absent(means3D)
        scales = scales if has_scales else e
        rotations = rotations if has_rotations else e
        cov3D_precomp = cov3D_precomp if has_cov3D_precomp else e
        with torch.no_grad():
            return rp.RasterizeGaussiansfilterCUDA(means3D, scales, rotations, rs.scale_modifier_, cov3D_precomp,
                                                   rs.viewmatrix_, rs.projmatrix_, rs.tanfovx_, rs.tanfovy_,
                                                   rs.image_height_, rs.image_width_, rs.prefiltered_, False)

    def project2_image(self, means3D, means2D, opacities, has_shs, has_colors_precomp, has_scales, has_rotations,
                       has_cov3D_precomp, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        self._check(has_shs, has_colors_precomp, has_scales, has_rotations, has_cov3D_precomp)
        rs = self.raster_settings_
        e = self._absent(means3D)
        shs = shs if has_shs else e
        colors_precomp = colors_precomp if has_colors_precomp else e
        scales = scales if has_scales else e
        rotations = rotations if has_rotations else e
        cov3D_precomp = cov3D_precomp if has_cov3D_precomp else e
        points_image_2d, radii, color = rp.RasterizeGaussiansprojectCUDA(
            rs.bg_, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier_, cov3D_precomp,
            rs.viewmatrix_, rs.projmatrix_, rs.tanfovx_, rs.tanfovy_, rs.image_height_, rs.image_width_, shs,
            rs.sh_degree_, rs.campos_, rs.prefiltered_)
        return points_image_2d, radii, color
