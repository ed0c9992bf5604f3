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
sc = t.fuzz_scene(seed)      # the scene tools/fuzz_raster.py builds for this seed
o, _ = t.gs_oracle.run_scene(sc, backward=False)
print("unstable fraction", o.unstable_pixels(1e-5).mean())
try:
    t.run_parity(sc, backward=True)
    print("parity ok")
except AssertionError:
    traceback.print_exc()
