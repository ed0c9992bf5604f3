"""Time segs_knn_mean_dist2 (distCUDA2) on seeded point clouds (GPU box): python tools/time_knn.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segs_slam_amd.points import distCUDA2
for n in [int(a) for a in sys.argv[1:]] or [100_000, 1_000_000]:
    g = torch.Generator().manual_seed(n)
    pts = (torch.rand(n, 3, generator=g) * torch.tensor([8.0, 6.0, 3.0])).cuda()
    distCUDA2(pts); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d = distCUDA2(pts)
    torch.cuda.synchronize()
    print(f"N={n}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call, mean dist2 {float(d.mean()):.3e}")
