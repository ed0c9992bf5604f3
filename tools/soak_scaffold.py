#!/usr/bin/env python3
"""Soak run of the anchor-level mapper step with densification (run on the GPU box), SURVEY.md 8d config 3 in spirit:
   tools/soak_scaffold.py [iters] [anchors] [densify_grad_threshold] [stop_at_anchors]
A low threshold makes the map grow quickly (capacity growth, Adam-state extension, pruning all exercised)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segs_slam_amd import densify, neural_gaussians as ng, scenes  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
A = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0002
stop_at = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
sc = scenes.make_config_scene("c2")
cam = sc.camera
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
kfs, gts = [], []
from segs_slam_amd.keyframe_window import SlidingWindowKeyframes  # noqa: E402
for k in range(8):                      # 8 keyframes on a small orbit, targets = a smooth pattern per keyframe
    camk = scenes.make_config_scene("c2", keyframe=k).camera
    kfs.append(ng.Keyframe(t(camk.world_view_transform), t(camk.full_proj_transform), t(camk.camera_center),
                           torch.tensor([0.01 * k, 0.0, 0.0, 1.0, 0.0, 0.01 * k, 0.0], device=dev), camk.tanfovx, camk.tanfovy))
    yy, xx = torch.meshgrid(torch.linspace(0, 1, cam.height, device=dev), torch.linspace(0, 1, cam.width, device=dev), indexing="ij")
    gts.append(torch.stack([0.5 + 0.4 * torch.sin(6 * xx + k), 0.5 + 0.4 * torch.cos(5 * yy), 0.3 + 0.2 * xx * yy]).contiguous())
model = ng.synthetic_model(A, ng.ModelDims(), cam, dev, seed=1)
step = ng.ScaffoldTrainerStep(model, cam.width, cam.height, scaling_reg_weight=0.01)
dens = densify.AnchorDensifier(model, densify.DensifyParams(voxel_size=0.01, start_stat=100, update_from=300, update_interval=100,
                                                            update_until=iters, densify_grad_threshold=thr))
step.enable_densification(dens, seed=0)
if os.environ.get("SOAK_FREQUENCY_REGULARIZER"):      # the Replica mapper loss: multi_scale_loss on from the first iteration, row mask on
    step.enable_frequency_regularization(start=0)
    step.row_mask = True
step.keyframe_selector = SlidingWindowKeyframes(seed=0)      # the mapper's keyframe walk (src/gaussian_mapper.cpp:1459-1495)
for _ in kfs:
    step.keyframe_selector.add_keyframe(8)                    # Mapper.new_keyframe_times_of_use
t0 = t_last = time.perf_counter()
for it in range(1, iters + 1):
    loss = step.training_once(kfs, gts)
    if it % 100 == 0:
        torch.cuda.synchronize()
        l = float(loss)
        assert np.isfinite(l), (it, l)
        print(f"it {it:5d} loss {l:.5f} anchors {model.A:7d} capacity {model.capacity:7d} R {step.engine.R:8d} "
              f"mem {torch.cuda.memory_allocated() / 2**20:7.0f} MiB  {1e3 * (time.perf_counter() - t0) / it:.3f} ms/it since start, "
              f"{10 * (time.perf_counter() - t_last):.3f} ms/it over the last 100 (one adjust_anchor included)", flush=True)
        t_last = time.perf_counter()
        if stop_at and model.A >= stop_at:
            break
step.finish()      # (a drop of the last iteration is only seen by the next call)
assert torch.isfinite(model.params).all()
print(f"soak ok (dropped {step.dropped_steps()} redone {step.redone_steps} lost {step.lost_steps()})")
