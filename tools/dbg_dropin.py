import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segs_slam_amd import scenes, _capi, rasterize_points as rp
from segs_slam_amd.raster_engine import KernelProfile
dev = torch.device("cuda:0")
sc = scenes.make_config_scene("1080p_3m"); cam = sc.camera
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bg, m3, col, op, sca, rot = t(sc.bg), t(sc.means3D), t(sc.colors), t(sc.opacity), t(sc.scales), t(sc.rotations)
view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
dL = t(sc.dL_dout_color); e = torch.empty(0, device=dev); lib = _capi.lib()
def one():
    R, color, radii, geom, binning, img = rp.RasterizeGaussiansCUDA(bg, m3, col, op, sca, rot, 1.0, e, view, proj, cam.tanfovx, cam.tanfovy, cam.height, cam.width, e, 0, campos, False)
    rp.RasterizeGaussiansBackwardCUDA(bg, m3, radii, col, sca, rot, 1.0, e, view, proj, cam.tanfovx, cam.tanfovy, dL, e, 0, campos, geom, R, binning, img)
    return R
for flags in (32, 0, 32, 0):
    lib.segs_raster_set_flags(flags)
    for _ in range(10): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): R = one()
    torch.cuda.synchronize(); w = time.perf_counter() - t0
    with KernelProfile() as prof:
        for _ in range(10): one()
        torch.cuda.synchronize()
    ks = {k: round(v["total_ms"] / 10, 4) for k, v in prof.result.items() if v["total_ms"] > 0}
    print("flags", flags, "R", R, "ms/step", round(w / 30 * 1e3, 4), "kernel sum", round(sum(ks.values()), 4), ks, flush=True)
    print(torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_reserved() >> 20)
