"""Keyframe-parallel training of the anchor-level step rehearsed on ONE GPU: two processes (gloo, both on
cuda:0) hold identical replicas, each renders its own keyframe, the flat gradient bucket is all-reduced and the fused
Adam applies the mean.  Checks: replicas stay bit-identical; the all-reduced gradient of the first step equals the sum of
the two keyframes' gradients accumulated by a single process (parity at gradient level; float atomics in the
raster backward sum in arbitrary order -> 1e-4 relative, not bits); the parameters after two steps agree except where
Adam (eps 1e-15) turns a gradient that is itself rounding noise into a full-size step.  The product path uses backend
"nccl" (RCCL) with one GPU per rank."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _keyframes(dev):
    from segs_slam_amd import neural_gaussians as ng, scenes
    out = []
    for k in range(2):
        ang = 0.04 * (k + 1)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=np.float32)
        cam = scenes.make_camera(320, 240, 300.0, 300.0, R, np.array([0.05 * k, 0.0, 0.0], dtype=np.float32))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        out.append((cam, ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                                     torch.tensor([0.05 * k, 0.0, 0.0, 1.0, 0.0, ang, 0.0], device=dev), cam.tanfovx, cam.tanfovy)))
    return out


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    from segs_slam_amd import neural_gaussians as ng
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    kfs = _keyframes(dev)
    model = ng.synthetic_model(3000, ng.ModelDims(), kfs[0][0], dev, seed=7)
    step = ng.ScaffoldTrainerStep(model, 320, 240)
    gts = [torch.full((3, 240, 320), 0.3 + 0.2 * k, device=dev) for k in range(2)]
    # the first step's exchanged gradient, as training_once forms it (the optimizer zeroes the bucket afterwards)
    step._forward_backward(kfs[rank][1], gts[rank])
    dist.all_reduce(model.grads)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(outdir, "grads_step1.npy"), model.grads.cpu().numpy())
    model.grads.zero_()
    for _ in range(2):
        step.training_once([k for _, k in kfs], gts)       # rank r takes keyframe (it * world + r) % 2 = r
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"params_{rank}.npy"), model.params.cpu().numpy())
    dist.destroy_process_group()


def test_two_rank_scaffold_step_matches_single_process_mean_gradient():
    import torch.multiprocessing as mp
    from segs_slam_amd import neural_gaussians as ng
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, 29533, d), nprocs=2, join=True)
        p0, p1 = np.load(os.path.join(d, "params_0.npy")), np.load(os.path.join(d, "params_1.npy"))
        g_dp = np.load(os.path.join(d, "grads_step1.npy"))
    assert np.array_equal(p0, p1), "replicas diverged"
    # single process: both keyframes' gradients accumulated, one Adam step with the mean
    dev = torch.device("cuda:0")
    kfs = _keyframes(dev)
    model = ng.synthetic_model(3000, ng.ModelDims(), kfs[0][0], dev, seed=7)
    step = ng.ScaffoldTrainerStep(model, 320, 240)
    gts = [torch.full((3, 240, 320), 0.3 + 0.2 * k, device=dev) for k in range(2)]
    for it in range(2):
        for k in range(2):
            step._forward_backward(kfs[k][1], gts[k])       # gradients accumulate in model.grads
        if it == 0:
            g_one = model.grads.cpu().numpy()
            tol = 1e-4 * np.abs(g_one) + 1e-5 * np.abs(g_one).max()
            assert np.all(np.abs(g_dp - g_one) <= tol), float(np.abs(g_dp - g_one).max())
        step.world = 2                                       # grad_scale = 1/2
        step.iteration += 1
        groups = model.adam_groups(step.learning_rates(step.iteration))
        step.mlp_steps += 1
        step.anchor_steps += 1
        step._adam(groups, step.mlp_steps)
    torch.cuda.synchronize()
    ref = model.params.cpu().numpy()
    scale = np.abs(ref).max()
    off = np.abs(p0 - ref) > 2e-5 * scale
    assert off.mean() < 1e-3, (float(off.mean()), float(np.abs(p0 - ref).max()))
    assert np.abs(p0 - ref).max() <= 4 * 0.08, float(np.abs(p0 - ref).max())     # two steps of at most lr each, both ways
