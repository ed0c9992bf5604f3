"""ctypes front-end of the CPU oracle (oracle/gs_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (segs-slam_amd/) never does.  Parity status: "parity unpinned" by the reference (it
ships no tests/golden vectors for this path) -- see the header of gs_oracle.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgs_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("gs_oracle.cpp", "aux_oracle.cpp", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gso_create.restype = C.c_void_p
        _lib.gso_destroy.argtypes = [C.c_void_p]
        _lib.gso_forward.restype = C.c_int
        _lib.gso_num_rendered.restype = C.c_int
        _lib.gso_num_rendered.argtypes = [C.c_void_p]
        _lib.gso_sort_bits.restype = C.c_int
        _lib.gso_sort_bits.argtypes = [C.c_void_p]
        _lib.gso_threads.restype = C.c_int
    return _lib


def _p(a, dtype=np.float32):
    """C pointer of a contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """One forward (+ optional backward) of the reference algorithm on the CPU."""

    def __init__(self):
        self._l = lib()
        self._h = C.c_void_p(self._l.gso_create())
        self.P = self.W = self.H = 0
        self.M = 0

    def __del__(self):
        try:
            self._l.gso_destroy(self._h)
        except Exception:
            pass

    # -- CudaRasterizer::Rasterizer::forward (cuda_rasterizer/rasterizer_impl.cu:198-336)
    def forward(self, bg, means3D, colors, opacity, scales, scale_modifier, rotations, viewmatrix, projmatrix,
                tanfovx, tanfovy, H, W, cov3D_precomp=None, sh=None, degree=0, campos=None) -> int:
        self._in = dict(bg=_f32(bg), means3D=_f32(means3D), colors=_f32(colors), opacity=_f32(opacity),
                        scales=_f32(scales), rotations=_f32(rotations), view=_f32(viewmatrix), proj=_f32(projmatrix),
                        cov=_f32(cov3D_precomp), mod=float(scale_modifier), tx=float(tanfovx), ty=float(tanfovy),
                        sh=_f32(sh), campos=_f32(campos), D=int(degree), M=0 if sh is None else int(np.asarray(sh).shape[1]))
        self.M = self._in["M"]
        i = self._in
        self.P, self.W, self.H = int(i["means3D"].shape[0]), int(W), int(H)
        R = self._l.gso_forward(self._h, C.c_int(self.P), _p(i["bg"]), C.c_int(W), C.c_int(H), _p(i["means3D"]),
                                _p(i["colors"]), _p(i["opacity"]), _p(i["scales"]), C.c_float(i["mod"]),
                                _p(i["rotations"]), _p(i["cov"]), _p(i["view"]), _p(i["proj"]),
                                C.c_float(i["tx"]), C.c_float(i["ty"]), _p(i["sh"]), C.c_int(i["D"]), C.c_int(i["M"]), _p(i["campos"]))
        self.R = int(R)
        return self.R

    # -- CudaRasterizer::Rasterizer::backward (cuda_rasterizer/rasterizer_impl.cu:397-490)
    def backward(self, dL_dout_color, dL_dmean2D_in=None, dL_dconic_in=None):
        i = self._in
        dL = _f32(dL_dout_color)
        a, b = _f32(dL_dmean2D_in), _f32(dL_dconic_in)
        self._l.gso_backward(self._h, _p(i["bg"]), _p(i["means3D"]), _p(i["scales"]), C.c_float(i["mod"]),
                             _p(i["rotations"]), _p(i["cov"]), _p(i["view"]), _p(i["proj"]),
                             C.c_float(i["tx"]), C.c_float(i["ty"]), _p(dL), _p(a), _p(b), _p(i["sh"]), _p(i["campos"]))
        return {k: self.get(k) for k in ("dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_dmean3D",
                                         "dL_dcov3D", "dL_dscale", "dL_drot", "dL_dsh")}

    _SHAPES = {
        "radii": (np.int32, lambda s: (s.P,)), "depths": (np.float32, lambda s: (s.P,)),
        "means2D": (np.float32, lambda s: (s.P, 2)), "conic_opacity": (np.float32, lambda s: (s.P, 4)),
        "cov3D": (np.float32, lambda s: (s.P, 6)),
        "tiles_touched": (np.uint32, lambda s: (s.P,)), "point_offsets": (np.uint32, lambda s: (s.P,)),
        "keys_unsorted": (np.uint64, lambda s: (s.R,)), "vals_unsorted": (np.uint32, lambda s: (s.R,)),
        "keys": (np.uint64, lambda s: (s.R,)), "point_list": (np.uint32, lambda s: (s.R,)),
        "ranges": (np.uint32, lambda s: (s.ntiles, 2)),
        "n_contrib": (np.uint32, lambda s: (s.H, s.W)), "final_T": (np.float32, lambda s: (s.H, s.W)),
        "out_color": (np.float32, lambda s: (3, s.H, s.W)),
        "dL_dmean2D": (np.float32, lambda s: (s.P, 3)), "dL_dconic": (np.float32, lambda s: (s.P, 2, 2)),
        "dL_dopacity": (np.float32, lambda s: (s.P, 1)), "dL_dcolor": (np.float32, lambda s: (s.P, 3)),
        "dL_dmean3D": (np.float32, lambda s: (s.P, 3)), "dL_dcov3D": (np.float32, lambda s: (s.P, 6)),
        "dL_dscale": (np.float32, lambda s: (s.P, 3)), "dL_drot": (np.float32, lambda s: (s.P, 4)),
        "dL_dsh": (np.float32, lambda s: (s.P, s.M, 3)), "rgb": (np.float32, lambda s: (s.P, 3)),
    }

    @property
    def ntiles(self) -> int:
        return ((self.W + 15) // 16) * ((self.H + 15) // 16)

    @property
    def sort_bits(self) -> int:
        return int(self._l.gso_sort_bits(self._h))

    def get(self, name: str) -> np.ndarray:
        dt, shp = self._SHAPES[name]
        out = np.zeros(shp(self), dtype=dt)
        getattr(self._l, "gso_get_" + name)(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    def unstable_pixels(self, rel: float = 1e-4) -> np.ndarray:
        out = np.zeros((self.H, self.W), dtype=np.uint8)
        self._l.gso_unstable_pixels(self._h, C.c_float(rel), out.ctypes.data_as(C.c_void_p))
        return out.astype(bool)


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    """src/rasterize_points.cu:195-214 -> cuda_rasterizer/rasterizer_impl.cu:54-66."""
    m, v, p = _f32(means3D), _f32(viewmatrix), _f32(projmatrix)
    out = np.zeros(m.shape[0], dtype=np.uint8)
    lib().gso_mark_visible(C.c_int(m.shape[0]), _p(m), _p(v), _p(p), out.ctypes.data_as(C.c_void_p))
    return out.astype(bool)


def visible_filter(means3D, scales, rotations, scale_modifier, viewmatrix, projmatrix, tanfovx, tanfovy, H, W,
                   cov3D_precomp=None) -> np.ndarray:
    """src/rasterize_points.cu:216-280 -> rasterizer_impl.cu:339-393 -> forward.cu:259-334 (radii only)."""
    l = lib()
    h = C.c_void_p(l.gso_create())
    m, s, r = _f32(means3D), _f32(scales), _f32(rotations)
    out = np.zeros(m.shape[0], dtype=np.int32)
    l.gso_visible_filter(h, C.c_int(m.shape[0]), C.c_int(W), C.c_int(H), _p(m), _p(s), C.c_float(scale_modifier),
                         _p(r), _p(_f32(cov3D_precomp)), _p(_f32(viewmatrix)), _p(_f32(projmatrix)),
                         C.c_float(tanfovx), C.c_float(tanfovy), out.ctypes.data_as(C.c_void_p))
    l.gso_destroy(h)
    return out


def run_scene(scene, backward: bool = True):
    """Convenience: oracle forward(+backward) of a segs_slam_amd.scenes.Scene."""
    cam = scene.camera
    o = Oracle()
    o.forward(scene.bg, scene.means3D, scene.colors, scene.opacity, scene.scales, scene.scale_modifier,
              scene.rotations, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
              cam.height, cam.width)
    grads = o.backward(scene.dL_dout_color) if backward else None
    return o, grads


# ---- point-set helpers (oracle/aux_oracle.cpp) ------------------------------------------------------------
def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def knn_mean_dist2(points) -> np.ndarray:
    """distCUDA2 (third_party/simple-knn/spatial.cu:15-26): exact mean of the 3 smallest squared distances (O(P^2))."""
    p = _f32(points)
    out = np.zeros(p.shape[0], dtype=np.float32)
    lib().gso_knn_mean_dist2(C.c_int(p.shape[0]), _p(p), _p(out))
    return out


def transform_points(points, M) -> np.ndarray:
    p, m = _f32(points), _f32(M)
    out = np.zeros_like(p)
    lib().gso_transform_points(C.c_int(p.shape[0]), _p(p), _p(m), _p(out))
    return out


def scale_and_transform_points(points, rots, M, mask, scale):
    p, r, m, k = _f32(points), _f32(rots), _f32(M), _u8(mask)
    out_p, out_r = np.zeros_like(p), np.zeros_like(r)
    lib().gso_scale_and_transform_points(C.c_int(p.shape[0]), C.c_float(scale), _p(p), _p(r), _p(m), _p(k, np.uint8),
                                         _p(out_p), _p(out_r))
    return out_p, out_r


def reproject_depths_pinhole(depths, mask, intr, width) -> np.ndarray:
    d, k = _f32(depths), _u8(mask)
    out = np.zeros((d.shape[0], 3), dtype=np.float32)
    lib().gso_reproject_depths_pinhole(C.c_int(d.shape[0]), C.c_int(width), *[C.c_float(x) for x in intr], _p(d), _p(k, np.uint8), _p(out))
    return out


def search_neighborhood_depth(pixels, has3D, p3d, colors, max_pixel_dist, intr, width):
    px, h, p, c = _f32(pixels), _u8(has3D), _f32(p3d), _f32(colors)
    out_p, out_c = np.zeros_like(p), np.zeros_like(p)
    lib().gso_search_neighborhood_depth(C.c_int(px.shape[0]), C.c_int(width), *[C.c_float(x) for x in intr], C.c_float(max_pixel_dist),
                                        _p(px), _p(h, np.uint8), _p(p), _p(c), _p(out_p), _p(out_c))
    return out_p, out_c
