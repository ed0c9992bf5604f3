import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import gs_oracle
from segs_slam_amd import scenes
import test_raster_gpu as t
for (P, W, H) in [(2000, 64, 64), (20000, 64, 64), (20000, 128, 128)]:
    sc = scenes.make_scene(P, W, H, 60.0, 60.0, seed=5)
    o, _ = gs_oracle.run_scene(sc, backward=False)
    args, fwd = t.gpu_forward(sc)
    unstable = o.unstable_pixels(1e-5)
    dL = sc.dL_dout_color.copy(); dL[:, unstable] = 0
    ref = o.backward(dL); got = t.gpu_backward(sc, args, fwd, dL)
    a, b = got["dL_dcolor"].reshape(ref["dL_dcolor"].shape), ref["dL_dcolor"]
    bad = np.abs(a - b) > 1e-4 * np.abs(b) + 1e-5 * np.abs(b).max()
    ranges = o.get("ranges"); print(P, W, H, "R", fwd[0], "max list", int((ranges[:,1]-ranges[:,0]).max()), "bad colour rows", int(bad.any(axis=1).sum()), "of", int((np.abs(b).sum(axis=1) > 0).sum()))
