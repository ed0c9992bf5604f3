#!/usr/bin/env python3
"""tools/config3_growth.py [max_iters] [target_anchors] [grad_threshold] [teacher]: the config-3 growth run (segs_slam_amd.config3) with
its anchors-over-time log (run on the GPU box)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from segs_slam_amd import config3, densify  # noqa: E402

max_iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
target = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
params = densify.DensifyParams(start_stat=100, update_from=300, update_interval=100, update_until=10 ** 9)
if len(sys.argv) > 3:
    params.densify_grad_threshold = float(sys.argv[3])
teacher = sys.argv[4] if len(sys.argv) > 4 else "c2"
run = config3.Config3Run(torch.device("cuda:0"), params=params, teacher=teacher)
print(f"student: {run.model.A} anchors, teacher frames {len(run.targets)}", flush=True)
res = run.run(max_iters, target, log=lambda s: print(s, flush=True))
assert torch.isfinite(run.model.params).all()
print(json.dumps(res))
