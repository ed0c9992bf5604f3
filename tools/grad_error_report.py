#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP rasterizer against the CPU oracle at BASELINE.json's sizes.

For each workload and each of the seven gradient tensors: the fraction of entries inside the PURE 1e-4 relative bound, the
fraction inside the tests' bound (1e-4 |ref| + 1e-6 max|ref|), the largest error over max|ref|, and the same three numbers for
the resident path (what bench.py times).  usage (GPU box): python tools/grad_error_report.py [workload ...] > profiles/rNN_grad_error.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import gs_oracle  # noqa: E402
from segs_slam_amd import scenes  # noqa: E402
from segs_slam_amd.raster_engine import RasterEngine  # noqa: E402

DEV = "cuda:0"


def stats(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    err, mx = np.abs(a - b), np.abs(b).max() + 1e-300
    nz = np.abs(b) > 0
    pure = (err[nz] <= 1e-4 * np.abs(b[nz])).mean() if nz.any() else 1.0
    test = (err <= 1e-4 * np.abs(b) + 1e-6 * mx).mean()
    return pure, test, err.max() / mx


def main():
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)  # noqa: E731
    for wl in (sys.argv[1:] or ["c2_1080p", "1080p_3m"]):
        sc = scenes.make_config_scene(wl)
        cam = sc.camera
        o, _ = gs_oracle.run_scene(sc, backward=False)
        unstable = o.unstable_pixels(1e-5)
        dL = sc.dL_dout_color.copy()
        dL[:, unstable] = 0
        ref = o.backward(dL)
        a = [t(x) for x in (sc.bg, sc.means3D, sc.colors, sc.opacity, sc.scales, sc.rotations, cam.world_view_transform,
                            cam.full_proj_transform, cam.camera_center)]
        print(f"== {wl}: P={sc.P} R={o.R} unstable pixels={int(unstable.sum())} ({unstable.mean():.2e} of the image)")
        print(f"{'tensor':12s} {'path':9s} {'within 1e-4 rel':>16s} {'within test bound':>18s} {'max err / max|ref|':>19s}")
        for resident in (False, True):
            eng = RasterEngine(sc.P, cam.width, cam.height, DEV, resident=resident, want_cov3D_grad=True)
            for _ in range(2):
                eng.forward(*a, cam.tanfovx, cam.tanfovy)
                eng.backward(t(dL))
            eng.check()
            torch.cuda.synchronize()
            got = {"dL_dmean3D": eng.grads["means3D"], "dL_dscale": eng.grads["scales"], "dL_drot": eng.grads["rotations"],
                   "dL_dopacity": eng.grads["opacity"], "dL_dcolor": eng.grads["colors"], "dL_dmean2D": eng.dL_dmean2D,
                   "dL_dcov3D": eng.dL_dcov3D}
            for k, v in got.items():
                p, q, m = stats(v.cpu().numpy().reshape(ref[k].shape), ref[k])
                print(f"{k:12s} {'resident' if resident else 'sync':9s} {p:16.6f} {q:18.6f} {m:19.3e}")


if __name__ == "__main__":
    main()
