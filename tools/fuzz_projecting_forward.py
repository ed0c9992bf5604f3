#!/usr/bin/env python3
"""Randomised A/B of the projecting neural forward (segs_neural_forward_projected + segs_rasterize_forward_resident_projected:
K1 and prefilter_voxel inside the neural kernels) against segs_visible_filter_log_scales + segs_neural_forward +
segs_rasterize_forward_resident on the same model: image, radii, visible radii, candidate geometry, R and R_live must be equal
BIT FOR BIT; raster gradients within the atomics' summation noise.

usage: tools/fuzz_projecting_forward.py CASES SEED0      (on the MI355X box)
Cases vary: anchors (1 .. 60 000, not multiples of 32), model dimensions (feature bank, appearance width, the three *_dist
switches), image size (odd sizes included), camera rotation / position (anchors behind the camera, outside the frustum, very
near), anchor spread and scale, the rasterizer's KEEP_DEAD_INSTANCES flag, spare capacity rows behind the live anchors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from segs_slam_amd import neural_gaussians as ng, scenes

DEV = torch.device("cuda:0")


def rot(ax, ang):
    c, s = np.cos(ang), np.sin(ang)
    m = {0: [[1, 0, 0], [0, c, -s], [0, s, c]], 1: [[c, 0, s], [0, 1, 0], [-s, 0, c]], 2: [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[ax]
    return np.array(m, dtype=np.float32)


def one_case(seed):
    rng = np.random.default_rng(seed)
    A = int(rng.choice([1, 5, 31, 33, 700, 4001, 20_000, 60_000], p=[.05, .05, .05, .05, .2, .3, .2, .1]))
    W, H = (int(rng.integers(48, 700)), int(rng.integers(32, 500))) if rng.random() < 0.7 else ((640, 480) if rng.random() < 0.5 else (1200, 680))
    dims = ng.ModelDims(appearance_dim=int(rng.choice([0, 8, 16, 32])), use_feat_bank=bool(rng.random() < 0.5),
                        add_opacity_dist=bool(rng.random() < 0.3), add_cov_dist=bool(rng.random() < 0.3), add_color_dist=bool(rng.random() < 0.3))
    f = float(rng.uniform(0.5, 1.5)) * W
    R = rot(1, rng.normal() * 0.6) @ rot(0, rng.normal() * 0.3) @ rot(2, rng.normal() * 0.5)
    T = (rng.normal(size=3) * np.array([0.5, 0.5, 1.5])).astype(np.float32)
    cam = scenes.make_camera(W, H, f, f, R.astype(np.float32), T)
    keep_dead = rng.random() < 0.25
    spare = int(rng.choice([0, 0, 7, 1000]))
    mul_anchor, add_scale, mul_offset, mul_mlp = float(rng.uniform(0.3, 3.0)), float(rng.normal() * 0.7), float(rng.uniform(0.2, 3.0)), float(rng.uniform(0.5, 2.0))
    steps = []
    for fuse in (False, True):
        model = ng.synthetic_model(A, dims, scenes.make_camera(W, H, f, f, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32)), DEV,
                                   seed=seed)
        with torch.no_grad():   # spread the anchors and their scales beyond the tidy synthetic scene
            model.param("anchor").mul_(mul_anchor)
            model.param("scaling").add_(add_scale)
            model.param("offset").mul_(mul_offset)
            model.mlp_params.mul_(mul_mlp)
        if spare:
            model.reserve(A + spare)
        step = ng.ScaffoldTrainerStep(model, W, H)
        step.fuse_projection = fuse
        if keep_dead:
            step.engine.flags |= 2
        steps.append(step)
    a, b = steps
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)  # noqa: E731
    pose7 = torch.tensor(np.concatenate([T, [1.0, 0.0, 0.0, 0.0]]).astype(np.float32), device=DEV)
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center), pose7, cam.tanfovx, cam.tanfovy)
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(seed)).to(DEV) / (3 * H * W)
    info = ""
    for it in range(3):
        ia, ib = a.render(kf), b.render(kf)
        torch.cuda.synchronize()
        oka, okb = a.engine.check(raise_on_overflow=False), b.engine.check(raise_on_overflow=False)
        if not (oka and okb):
            assert oka == okb, "overflow flagged on one side only"
            info = " (capacity overflow on both: re-calibrating)"
            continue
        P = a.neural.P
        assert torch.equal(ia, ib), f"image differs at render {it}"
        assert torch.equal(a.engine.radii[:P], b.engine.radii[:P]), f"radii differ at render {it}"
        assert torch.equal(a.visible_radii[:A], b.visible_radii[:A]), f"visible radii differ at render {it}"
        assert (a.engine.R, a.engine.R_live) == (b.engine.R, b.engine.R_live), f"instance counts differ at render {it}"
        rows = (a.visible_radii[:A] > 0).repeat_interleave(10)
        for name in ("means3D", "scales", "rotations", "neural_opacity"):
            assert torch.equal(getattr(a.neural, name)[:P][rows], getattr(b.neural, name)[:P][rows]), f"{name} differs at render {it}"
        if a.engine.R > 0:
            # the tile backward sums with float atomics: two backwards of the SAME engine differ by that noise (and the per-Gaussian
            # stage amplifies it for ill-conditioned covariances), so the bar is ten times that noise as measured on side a (one pair of runs), plus 1e-5 of the largest entry
            ga = {k: v.clone() for k, v in a.engine.backward(dL).items()}
            ga2 = {k: v.clone() for k, v in a.engine.backward(dL).items()}
            gb = b.engine.backward(dL)
            for k in ga:
                scale = max(float(ga[k].abs().max()), 1e-30)
                noise = float((ga[k][:P] - ga2[k][:P]).abs().max())
                dev = float((ga[k][:P] - gb[k][:P]).abs().max())
                assert dev <= 10.0 * noise + 1e-5 * scale, f"gradient {k} differs at render {it}: {dev:.3e} against the engine's own run-to-run {noise:.3e} (largest entry {scale:.3e})"
    nvis = int((a.visible_radii[:A] > 0).sum())
    return (f"A={A:6d} {W}x{H} bank={int(dims.use_feat_bank)} app={dims.appearance_dim:2d} dist={int(dims.add_opacity_dist)}{int(dims.add_cov_dist)}{int(dims.add_color_dist)} "
            f"keep_dead={int(keep_dead)} spare={spare} visible={nvis} live={int(a.neural.mask().sum())} binned={int((a.engine.radii[:a.neural.P] > 0).sum())} R={a.engine.R}{info}")


def main():
    n, seed0 = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for i in range(n):
        try:
            print(f"case {i:3d} seed {seed0 + i} {one_case(seed0 + i)}: ok", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"case {i:3d} seed {seed0 + i}: FAILED {e}", flush=True)
    print(f"{n} cases, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
