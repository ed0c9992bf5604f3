#!/usr/bin/env python3
"""The Replica mapper step with its frequency regulariser (bench.py's `replica_step` block) on its own, for rocprofv3:
    rocprofv3 --kernel-trace --stats ... -- python3 tools/replica_step.py --variant fused|autograd_mirror [--anchors N]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="fused", choices=["fused", "autograd_mirror"])
ap.add_argument("--anchors", type=int, default=50_000)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
out = bench.replica_step_block(torch.device("cuda:0"), anchors=a.anchors, steps=a.steps,
                               variants=((a.variant, a.variant == "fused"),))
print(json.dumps(out))
