#!/usr/bin/env python3
"""What the tile kernels iterate over, per workload (GPU box): tile list lengths, (Gaussian, quadrant) pairs per wave,
entries of a wave's quadrant per 64-entry chunk, batches of 16 when cut per chunk or from the stream, and the spread of
work over workgroups (a tile's four waves end together only if their quadrants hold the same number of pairs).

usage: python tools/tile_stats.py [workload ...]      (default: 1080p_3m c2_1080p)
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segs_slam_amd import _capi, scenes  # noqa: E402
from segs_slam_amd.raster_engine import RasterEngine  # noqa: E402


def stats(workload: str) -> None:
    dev = torch.device("cuda", 0)
    sc = scenes.make_config_scene(workload, keyframe=0)
    cam = sc.camera
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    eng = RasterEngine(sc.P, cam.width, cam.height, dev, resident=True)
    args = (t(sc.bg), t(sc.means3D), t(sc.colors), t(sc.opacity), t(sc.scales), t(sc.rotations), t(cam.world_view_transform),
            t(cam.full_proj_transform), t(cam.camera_center), cam.tanfovx, cam.tanfovy)
    for _ in range(2):          # the second forward is the resident one (tight rectangles, dead instances dropped)
        eng.forward(*args)
        eng.backward(t(sc.dL_dout_color))
    eng.check()
    torch.cuda.synchronize()
    lib = _capi.lib()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    tiles = ((cam.width + 15) // 16) * ((cam.height + 15) // 16)
    ranges = torch.zeros((tiles, 2), dtype=torch.int32, device=dev)
    ncontrib = torch.zeros((cam.height, cam.width), dtype=torch.int32, device=dev)
    _capi.check(lib.segs_debug_unpack_image(C.c_void_p(eng._img_r.data_ptr()), cam.width, cam.height, C.c_void_p(ranges.data_ptr()),
                                            None, C.c_void_p(ncontrib.data_ptr()), st), "unpack_image")
    vals = torch.zeros(eng.capacity, dtype=torch.int32, device=dev)
    _capi.check(lib.segs_debug_instance_values(C.c_void_p(eng._bin_r.data_ptr()), eng.capacity, C.c_void_p(vals.data_ptr()), st),
                "instance_values")
    torch.cuda.synchronize()
    ranges = ranges.cpu().numpy().astype(np.int64)
    vals = vals.cpu().numpy().view(np.uint32)
    nc = ncontrib.cpu().numpy()
    H, W = nc.shape
    tx = (W + 15) // 16
    lens = ranges[:, 1] - ranges[:, 0]
    print(f"== {workload}: tiles {tiles}, live instances {int(lens.sum())}, tile list length mean {lens.mean():.0f} "
          f"median {np.median(lens):.0f} p99 {np.percentile(lens, 99):.0f} max {lens.max()}")

    pairs = np.zeros((tiles, 4), dtype=np.int64)       # entries of each quadrant up to its deepest contributor
    batches_chunk = np.zeros((tiles, 4), dtype=np.int64)
    batches_stream = np.zeros((tiles, 4), dtype=np.int64)
    chunks = np.zeros((tiles, 4), dtype=np.int64)
    n_hist = np.zeros(65, dtype=np.int64)
    for tile in range(tiles):
        s, e = ranges[tile]
        if e <= s:
            continue
        v = vals[s:e]
        ty, txi = divmod(tile, tx)
        for q in range(4):
            y0, x0 = ty * 16 + (q >> 1) * 8, txi * 16 + (q & 1) * 8
            blk = nc[y0:y0 + 8, x0:x0 + 8]
            last = int(blk.max()) if blk.size else 0
            if last == 0:
                continue
            rel = ((v[:last] >> np.uint32(28 + q)) & np.uint32(1)).astype(np.int64)
            nchunk = (last + 63) // 64
            pad = np.zeros(nchunk * 64, dtype=np.int64)
            pad[:last] = rel
            per_chunk = pad.reshape(nchunk, 64).sum(axis=1)
            n_hist += np.bincount(per_chunk, minlength=65)
            pairs[tile, q] = rel.sum()
            chunks[tile, q] = nchunk
            batches_chunk[tile, q] = ((per_chunk + 15) // 16).sum()
            batches_stream[tile, q] = (rel.sum() + 15) // 16
    tot = pairs.sum()
    waves = (pairs > 0).sum()
    print(f"   (Gaussian, quadrant) pairs {tot} over {waves} waves: mean {tot / max(waves, 1):.0f} per wave, max {pairs.max()}; "
          f"chunks {chunks.sum()}, entries of the wave's quadrant per chunk: mean {tot / max(chunks.sum(), 1):.1f}")
    print(f"   batches of 16: cut per chunk {batches_chunk.sum()} ({tot / max(batches_chunk.sum(), 1) / 16:.2f} full), "
          f"cut from the stream {batches_stream.sum()} ({tot / max(batches_stream.sum(), 1) / 16:.2f} full)")
    wg = pairs.max(axis=1)           # a workgroup's slot is held until its longest wave ends
    print(f"   per workgroup: sum over tiles of the longest quadrant {wg.sum()} vs mean quadrant {pairs.mean(axis=1).sum():.0f} "
          f"(x{wg.sum() / max(pairs.mean(axis=1).sum(), 1):.2f}); longest workgroup {wg.max()} pairs "
          f"= {wg.max() / max(wg.sum() / 1024, 1):.2f} of one slot's share at 1024 resident workgroups")
    pix_useful = int(nc.astype(np.int64).sum())
    print(f"   pixel evaluations {tot * 64} (pairs x 64); sum n_contrib {pix_useful}")


if __name__ == "__main__":
    for wl in (sys.argv[1:] or ["1080p_3m", "c2_1080p"]):
        stats(wl)
