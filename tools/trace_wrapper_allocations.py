"""Per call of the Python mirror of RasterizeGaussiansCUDA / BackwardCUDA: wall time, scratch sizes, device allocations the tensor
library had to make and reserved memory (round 4: how the callback reference cycle of _ResizableBuffer was found).
    python tools/trace_wrapper_allocations.py <segs_raster_set_flags bits>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segs_slam_amd import scenes, _capi, rasterize_points as rp
dev = torch.device("cuda:0")
sc = scenes.make_config_scene("1080p_3m"); cam = sc.camera
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bg, m3, col, op, sca, rot = t(sc.bg), t(sc.means3D), t(sc.colors), t(sc.opacity), t(sc.scales), t(sc.rotations)
view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
dL = t(sc.dL_dout_color); e = torch.empty(0, device=dev)
_capi.lib().segs_raster_set_flags(int(sys.argv[1]))
orig_empty = torch.empty
log = []
def empty(*a, **k):
    s0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
    r = orig_empty(*a, **k)
    s1 = torch.cuda.memory_stats(dev)["num_device_alloc"]
    if s1 != s0: log.append((r.numel() * r.element_size()) >> 20)
    return r
rp.torch.empty = empty
def one():
    R, color, radii, geom, binning, img = rp.RasterizeGaussiansCUDA(bg, m3, col, op, sca, rot, 1.0, e, view, proj, cam.tanfovx, cam.tanfovy, cam.height, cam.width, e, 0, campos, False)
    rp.RasterizeGaussiansBackwardCUDA(bg, m3, radii, col, sca, rot, 1.0, e, view, proj, cam.tanfovx, cam.tanfovy, dL, e, 0, campos, geom, R, binning, img)
    return geom.numel() >> 20, binning.numel() >> 20, img.numel() >> 20
for i in range(14):
    log.clear()
    st0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter(); sizes = one(); torch.cuda.synchronize(); w = time.perf_counter() - t0
    st = torch.cuda.memory_stats(dev)
    print(i, "ms", round(w * 1e3, 3), "sizes MB", sizes, "new device allocs (MB)", log, "frees", st["num_device_free"] - st0["num_device_free"], "reserved MB", st["reserved_bytes.all.current"] >> 20, flush=True)
