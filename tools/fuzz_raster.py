#!/usr/bin/env python3
"""Randomised parity sweep of the rasterizer against the CPU oracle (run on the GPU box; not part of the test suite):
tools/fuzz_raster.py [n_cases] [seed0].  Scenes vary in P, image size (also non-multiples of 16), focal length, background,
scale multiplier; every case goes through tests/test_raster_gpu.run_parity (bit-exact integers, tolerances on floats)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from segs_slam_amd import scenes  # noqa: E402
import test_raster_gpu as t  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fails = 0
for i in range(n_cases):
    rng = np.random.default_rng(seed0 + i)
    P = int(rng.choice([1, 3, 50, 700, 4000, 20000, 60000]))
    W, H = int(rng.integers(17, 400)), int(rng.integers(17, 300))
    f = float(rng.uniform(0.4, 1.5)) * max(W, H)
    bg = tuple(float(x) for x in rng.choice([0.0, 0.5, 1.0], size=3))
    sc = scenes.make_scene(P, W, H, f, f, seed=seed0 + i, bg=bg)
    sc.scales *= float(rng.choice([0.3, 1.0, 3.0, 10.0]))
    try:
        t.run_parity(sc, backward=True)
        status = "ok"
    except AssertionError as e:
        status = "FAIL " + str(e)[:200]
        fails += 1
    print(f"case {i:3d} seed {seed0 + i} P={P:6d} {W}x{H} f={f:7.1f} bg={bg} scale_mul={float(sc.scales.mean()):.4f}: {status}", flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
