#!/usr/bin/env python3
"""Generates tests/golden/raster_scenes.npz: the CPU oracle's results on the seeded scenes SURVEY.md 8c(iv) lists
(P in {0, 1, 17, 1000} in full; P = 50 000 at 640x480 as digests and per-tile sums).

This fixture pins OUR oracle against drift (compiler, flags, refactors) and gives the HIP path a target that does not
need the oracle at run time.  It does NOT pin the oracle against the reference: the reference's rasterizer is CUDA-only
and holds no golden vectors for it -- for the rasterizer, parity with the reference stays "unpinned" and
rests on the oracle citing forward.cu / backward.cu / rasterizer_impl.cu line by line plus the independent autograd
check (tests/test_oracle.py).  The fixture holds inputs and outputs only."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gs_oracle  # noqa: E402
from segs_slam_amd import scenes  # noqa: E402

INT_KEYS = ("radii", "tiles_touched", "point_offsets", "keys", "point_list", "ranges", "n_contrib")
FLOAT_KEYS = ("means2D", "depths", "conic_opacity", "final_T", "out_color")
GRAD_KEYS = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot")
FULL_CASES = [(0, 32, 32, 30.0, (0.5, 0.25, 0.125)), (1, 16, 16, 14.4, (0, 0, 0)), (17, 33, 17, 29.7, (0, 0, 0)),
              (1000, 64, 64, 57.6, (0.1, 0.2, 0.3))]


def small_scene(P, W, H, f, bg):
    sc = scenes.make_scene(P, W, H, f, f, seed=4000 + P, bg=bg)
    sc.scales *= 3.0
    sc.dL_dout_color[:] = scenes.uniform01(sc.dL_dout_color.size, 55, P).reshape(sc.dL_dout_color.shape) * 2 - 1
    return sc


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tile_sums(img):      # (C,H,W) -> (C, ceil(H/16), ceil(W/16)) float64 sums
    Cn, H, W = img.shape
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = np.zeros((Cn, gy * 16, gx * 16), np.float64)
    pad[:, :H, :W] = img
    return pad.reshape(Cn, gy, 16, gx, 16).sum(axis=(2, 4))


def main():
    out = {}
    for P, W, H, f, bg in FULL_CASES:
        sc = small_scene(P, W, H, f, bg)
        o, _ = gs_oracle.run_scene(sc, backward=False)
        unstable = o.unstable_pixels(1e-5)
        dL = sc.dL_dout_color.copy()
        dL[:, unstable] = 0.0
        grads = o.backward(dL)
        tag = f"p{P}_"
        out[tag + "R"] = np.int64(o.R)
        out[tag + "sort_bits"] = np.int64(o.sort_bits)
        out[tag + "unstable"] = unstable
        for k in INT_KEYS + FLOAT_KEYS:
            out[tag + k] = o.get(k)
        for k in GRAD_KEYS:
            out[tag + k] = grads[k]
    # config 1 (BASELINE.json configs[0]): digests of the integer products, per-tile sums of the floats
    sc = scenes.make_config_scene("c1")
    o, _ = gs_oracle.run_scene(sc, backward=False)
    unstable = o.unstable_pixels(1e-5)
    dL = sc.dL_dout_color.copy()
    dL[:, unstable] = 0.0
    grads = o.backward(dL)
    out["c1_R"] = np.int64(o.R)
    out["c1_unstable_packed"] = np.packbits(unstable)
    for k in INT_KEYS[:-1]:
        out["c1_sha256_" + k] = np.array(digest(o.get(k)))
    stable = ~unstable
    out["c1_sha256_n_contrib_stable"] = np.array(digest(o.get("n_contrib")[stable]))
    out["c1_out_color_tile_sums"] = tile_sums(o.get("out_color") * stable[None])
    out["c1_final_T_tile_sums"] = tile_sums((o.get("final_T") * stable)[None])
    for k in GRAD_KEYS:
        g = grads[k].reshape(sc.P, -1).astype(np.float64)
        out["c1_colsum_" + k] = g.sum(0)
        out["c1_colabs_" + k] = np.abs(g).sum(0)
        out["c1_sample_" + k] = grads[k].reshape(sc.P, -1)[::97].copy()     # every 97th row in full
    path = os.path.join(ROOT, "tests", "golden", "raster_scenes.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
