"""ctypes binding of libsegs_raster.so (include/segs_raster.h).

The HIP library is the product: there is NO CPU fallback.  If the shared object is missing or a symbol
is absent, importing/using this module raises immediately (RuntimeError), so a test can never pass on a
silent fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# SEGS_RASTER_LIB: load another build of the library (A/B measurements); the default is the in-tree build
LIB_PATH = os.environ.get("SEGS_RASTER_LIB") or os.path.join(CSRC, "libsegs_raster.so")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class NeuralDims(C.Structure):
    """segs_neural_dims (include/segs_neural.h)."""
    _fields_ = [("feat_dim", C.c_int), ("n_offsets", C.c_int), ("appearance_dim", C.c_int), ("use_feat_bank", C.c_int),
                ("add_opacity_dist", C.c_int), ("add_cov_dist", C.c_int), ("add_color_dist", C.c_int)]


class ProjectionTargets(C.Structure):
    """segs_projection_targets (include/segs_raster.h)."""
    _fields_ = [("records", C.c_void_p), ("radii", C.c_void_p), ("tiles_touched", C.c_void_p), ("depth_keys", C.c_void_p),
                ("tile_ranges", C.c_void_p), ("depth_overflow", C.c_void_p), ("num_tiles", C.c_int), ("flags", C.c_uint32)]


class AdamSegment(C.Structure):
    """segs_adam_segment (include/segs_train.h)."""
    _fields_ = [("offset", C.c_int64), ("count", C.c_int64), ("lr", C.c_double)]

# name -> (restype, argtypes); every symbol include/segs_raster.h declares
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SYMBOLS = {
    "segs_last_error": (C.c_char_p, []),
    "segs_raster_set_flags": (C.c_uint, [C.c_uint]),
    "segs_raster_set_status_mirror": (_vp, [_vp]),
    "segs_training_statis": (_i, [_i, _i] + [_vp] * 9),
    "segs_training_statis_guarded": (_i, [_i, _i] + [_vp] * 10),
    "segs_anchor_growing_temp_bytes": (_sz, [_i, _i]),
    "segs_anchor_growing_level": (_i, [_i, _i, _i, _i] + [_vp] * 7 + [_f, _f, _f, _i] + [_vp] * 5),
    "segs_neural_set_flags": (C.c_uint, [C.c_uint]),
    "segs_neural_param_layout": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "segs_neural_temp_bytes": (_sz, [_vp, _i]),
    "segs_neural_forward": (_i, [_vp, _i] + [_vp] * 16),
    "segs_neural_forward_projected": (_i, [_vp, _i] + [_vp] * 16 + [_i, _i, _f, _f, _f, _vp, _vp]),
    "segs_neural_backward": (_i, [_vp, _i] + [_vp] * 17 + [_f, _vp, _vp, _vp]),
    "segs_geometry_bytes": (_sz, [_i]),
    "segs_image_bytes": (_sz, [_i, _i]),
    "segs_binning_bytes": (_sz, [_i]),
    "segs_rasterize_forward": (_i, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp,
                                     _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, C.POINTER(_i)]),
    "segs_rasterize_backward": (_i, [_i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "segs_resident_binning_bytes": (_sz, [_i, _i]),
    "segs_rasterize_forward_resident": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp,
                                              _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "segs_resident_projection_targets": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "segs_rasterize_forward_resident_projected": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "segs_rasterize_backward_resident": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp,
                                               _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "segs_visible_filter": (_i, [_i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp]),
    "segs_visible_filter_log_scales": (_i, [_i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _f, _f, _vp, _vp]),
    "segs_mark_visible": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "segs_project2_image": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i,
                                  _vp, _vp, _vp, _vp]),
    "segs_debug_unpack_geometry": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "segs_debug_instance_values": (_i, [_vp, _i, _vp, _vp]),
    "segs_debug_unpack_binning": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "segs_debug_unpack_image": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "segs_debug_preprocess_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _vp]),
    "segs_sort_pairs": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "segs_knn_temp_bytes": (_sz, [_i]),
    "segs_knn_mean_dist2": (_i, [_i, _vp, _vp, _vp, _vp]),
    "segs_transform_points": (_i, [_i, _vp, _vp, _vp, _vp]),
    "segs_scale_and_transform_points": (_i, [_i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "segs_reproject_depths_pinhole": (_i, [_i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "segs_search_neighborhood_depth": (_i, [_i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "segs_l1_ssim_temp_bytes": (_sz, [_i, _i]),
    "segs_l1_ssim_loss": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "segs_freq_pyramid": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "segs_spectrum_magnitude": (_i, [_vp, _sz, _vp, _vp]),
    "segs_freq_temp_bytes": (_sz, [_i, _i, _vp, _vp]),
    "segs_freq_spectrum_loss": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "segs_freq_pyramid_backward_add": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "segs_freq_plan_create": (_i, [_i, _i, _i, _vp, _f, _vp]),
    "segs_freq_plan_destroy": (None, [_vp]),
    "segs_freq_plan_levels": (_i, [_vp, _vp, _vp, _vp]),
    "segs_freq_target_floats": (_sz, [_vp]),
    "segs_freq_target": (_i, [_vp, _vp, _vp, _vp]),
    "segs_freq_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "segs_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_double, C.c_double, C.c_double, C.c_int64, _f, _i, _vp]),
    "segs_adam_step_guarded": (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_double, C.c_double, C.c_double, C.c_int64, _f, _i, _vp, _vp]),
    "segs_adam_step_device": (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_double, C.c_double, C.c_double, _vp, _i, _f, _i, _vp, _vp]),
    "segs_adam_step_graph": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, C.c_double, C.c_double, C.c_double, _vp, _f, _i, _vp, _vp]),
    "segs_set_doubles": (_i, [_vp, _vp, _i, _vp]),
    "segs_profile_begin": (_i, [C.c_uint]),
    "segs_profile_end": (_i, []),
    "segs_profile_kernel_count": (_i, []),
    "segs_profile_kernel_name": (C.c_char_p, [_i]),
    "segs_profile_query": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-s", "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension must be built (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C segs-slam_amd/csrc). There is no CPU fallback.")
        # PyTorch first: its wheel bundles its own libamdhip64; loaded afterwards, torch would bring a SECOND HIP runtime into
        # the process next to the system one this library resolved, and launches from here then fail with "no ROCm-capable
        # device is detected".  With torch loaded first the loader binds this library to the runtime torch already uses.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:  # pragma: no cover
                raise RuntimeError(f"libsegs_raster.so does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class SegsError(RuntimeError):
    pass


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().segs_last_error()
        raise SegsError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
