#!/usr/bin/env python3
"""fwd+bwd iterations per second through the Python mirror of the reference's tensor-typed entry points
(segs_slam_amd.rasterize_points), on a scene file written by bench.py (boundary_test's in.bin format):
    tools/dropin_python.py scene.bin steps warmup flags        -> one JSON object on stdout
flags: segs_raster_set_flags bits (0 = the reference's lists, 32 = SEGS_RASTER_TIGHT_BINNING)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from segs_slam_amd import _capi, rasterize_points as rp  # noqa: E402

path, steps, warmup, flags = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4], 0)
dev = torch.device("cuda:0")
with open(path, "rb") as f:
    P, W, H = np.fromfile(f, np.int32, 3)
    tanx, tany = np.fromfile(f, np.float32, 2)
    rd = lambda *shape: torch.from_numpy(np.fromfile(f, np.float32, int(np.prod(shape))).reshape(shape)).to(dev)  # noqa: E731
    bg, m3, col, op, sca, rot = rd(3), rd(P, 3), rd(P, 3), rd(P, 1), rd(P, 3), rd(P, 4)
    view, proj, campos, dL = rd(4, 4), rd(4, 4), rd(3), rd(3, H, W)
e = torch.empty(0, device=dev)
_capi.lib().segs_raster_set_flags(flags)


def one():
    R, color, radii, geom, binning, img = rp.RasterizeGaussiansCUDA(bg, m3, col, op, sca, rot, 1.0, e, view, proj, float(tanx), float(tany),
                                                                    int(H), int(W), e, 0, campos, False)
    rp.RasterizeGaussiansBackwardCUDA(bg, m3, radii, col, sca, rot, 1.0, e, view, proj, float(tanx), float(tany), dL, e, 0, campos, geom,
                                      R, binning, img)
    return R


for _ in range(warmup):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    R = one()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(json.dumps({"iters_per_s": steps / wall, "ms_per_step": wall / steps * 1e3, "steps": steps, "warmup": warmup, "flags": flags,
                  "num_rendered_returned": int(R), "P": int(P), "width": int(W), "height": int(H)}))
