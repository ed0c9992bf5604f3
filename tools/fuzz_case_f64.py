#!/usr/bin/env python3
"""Who is right when a fuzz case misses the gradient bar?  Rebuilds the scene of tools/fuzz_raster.py for one seed and
compares the per-Gaussian backward stage (K12 + K13 + cov3D backward) of the device and of the CPU oracle, both fed the
oracle's dL_dmean2D / dL_dconic, with the float64 evaluation of the defining equations (oracle/preprocess_backward_f64.py):
fraction of the nonzero entries within 1e-4 relative of the float64 value, and the largest error over max|truth|.

usage (GPU box): python tools/fuzz_case_f64.py SEED"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from segs_slam_amd import _capi, scenes  # noqa: E402
import test_raster_gpu as t  # noqa: E402
import torch  # noqa: E402
from oracle.preprocess_backward_f64 import stage_f64  # noqa: E402

seed = int(sys.argv[1])
sc = t.fuzz_scene(seed)
P, W, H, f = sc.P, sc.camera.width, sc.camera.height, float("nan")
print(f"seed {seed}: P={P} {W}x{H} f={f:.1f} mean scale {float(sc.scales.mean()):.4f}")
o, ref = t.gs_oracle.run_scene(sc)
cam = sc.camera
m3, sca, rot = t._t(sc.means3D), t._t(sc.scales), t._t(sc.rotations)
view, proj = t._t(cam.world_view_transform), t._t(cam.full_proj_transform)
radii = t._t(o.get("radii"), torch.int32)
d2, dc = t._t(ref["dL_dmean2D"]), t._t(ref["dL_dconic"])
outs = [torch.empty((P, n), device=t.DEV) for n in (3, 6, 3, 4)]
p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
_capi.check(_capi.lib().segs_debug_preprocess_backward(P, cam.width, cam.height, p(m3), p(radii), p(sca), 1.0, p(rot), None, p(view),
                                                       p(proj), cam.tanfovx, cam.tanfovy, p(d2), p(dc), *[p(x) for x in outs],
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "preprocess_backward")
torch.cuda.synchronize()
ref2 = o.backward(sc.dL_dout_color, ref["dL_dmean2D"], ref["dL_dconic"])
truth = stage_f64(sc.means3D, sc.scales, sc.rotations, cam.world_view_transform, cam.full_proj_transform, cam.width, cam.height,
                  cam.tanfovx, cam.tanfovy, ref["dL_dmean2D"], ref["dL_dconic"], o.get("radii"))
for x, k in zip(outs, ("dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot")):
    want = truth[k]
    nz = want != 0
    line = f"  {k:11s}"
    for who, got in (("device", x.cpu().numpy().astype(np.float64)), ("oracle", ref2[k].astype(np.float64))):
        err = np.abs(got - want)
        pure = float((err[nz] <= 1e-4 * np.abs(want[nz])).mean()) if nz.any() else 1.0
        line += f"  {who}: {pure * 100:7.3f} % within 1e-4, max err {err.max() / max(np.abs(want).max(), 1e-300):.2e} of max"
    both = np.abs(x.cpu().numpy().astype(np.float64) - ref2[k].astype(np.float64))
    nz2 = ref2[k] != 0
    line += f"  device vs oracle: {float((both[nz2] <= 1e-4 * np.abs(ref2[k][nz2])).mean()) * 100:7.3f} %"
    print(line)
