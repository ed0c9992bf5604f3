#!/usr/bin/env python3
"""One case of tools/fuzz_raster.py with the full traceback: python tools/fuzz_one.py SEED"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from segs_slam_amd import scenes  # noqa: E402
import test_raster_gpu as t  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
P = int(rng.choice([1, 3, 50, 700, 4000, 20000, 60000]))
W, H = int(rng.integers(17, 400)), int(rng.integers(17, 300))
f = float(rng.uniform(0.4, 1.5)) * max(W, H)
bg = tuple(float(x) for x in rng.choice([0.0, 0.5, 1.0], size=3))
sc = scenes.make_scene(P, W, H, f, f, seed=seed, bg=bg)
sc.scales *= float(rng.choice([0.3, 1.0, 3.0, 10.0]))
if rng.random() < 0.3:
    sc.opacity[:] = (sc.opacity * float(rng.choice([0.02, 0.2]))).astype(np.float32)
o, _ = t.gs_oracle.run_scene(sc, backward=False)
print("unstable fraction", o.unstable_pixels(1e-5).mean())
try:
    t.run_parity(sc, backward=True)
    print("parity ok")
except AssertionError:
    traceback.print_exc()
