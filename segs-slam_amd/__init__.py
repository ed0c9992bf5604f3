"""MI355X-native 3D-Gaussian-splatting rasterizer hot path for SEGS-SLAM (host-side mirror).

Sub-modules:
  scenes              seeded synthetic scenes + camera tensors (gaussian_keyframe.cpp math)
  _capi               ctypes binding of the C-ABI library (include/segs_raster.h)
  rasterize_points    mirror of include/rasterize_points.h (tensor-typed entry points)
  gaussian_rasterizer mirror of include/gaussian_rasterizer.h (settings, autograd Function, module)
"""
__all__ = ["scenes"]
