"""The C++/LibTorch-ROCm trainer twin (segs-slam_amd/csrc/torch_boundary/gaussian_trainer.{h,cpp}: trainingOnce over the C ABI,
src/gaussian_trainer.cpp:47-117 / src/gaussian_mapper.cpp:861-1030) against the Python step
(segs-slam_amd/neural_gaussians.py::ScaffoldTrainerStep) on the same model, keyframe and target.  Both drive the same HIP
kernels, so they differ only by the summation order of the float atomics in the raster backward: losses agree to 1e-5
relative in every iteration, and the parameters after the run to 2e-3 of each entry's own update on all but a sliver of
entries (Adam with eps 1e-15 turns a gradient that is itself rounding noise into a full-size step of either sign).
Step-level parity of the Python step against the float64 / oracle chain: tests/test_step_parity_gpu.py."""
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TB = os.path.join(ROOT, "segs-slam_amd", "csrc", "torch_boundary")


def test_trainer_twin_is_built_and_links_the_c_abi():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "segs-slam_amd", "csrc"), "-s", "-j4"])
    subprocess.check_call(["make", "-C", TB, "-s", "-j4"])
    out = subprocess.check_output(["nm", "-DC", os.path.join(TB, "libgaussian_trainer.so")], text=True)
    assert any("segs_host::GaussianTrainerStep::trainingOnce(" in l and " T " in l for l in out.splitlines())
    for sym in ("segs_visible_filter", "segs_neural_forward", "segs_rasterize_forward_resident", "segs_l1_ssim_loss",
                "segs_rasterize_backward_resident", "segs_neural_backward", "segs_adam_step_device"):
        assert any(l.strip().endswith("U " + sym) for l in out.splitlines()), sym      # resolved from libsegs_raster.so
    assert os.path.exists(os.path.join(TB, "trainer_test"))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["replica", "scannet"])
def test_cpp_trainer_matches_python_step(tmp_path, cfg):
    from segs_slam_amd import neural_gaussians as ng, scenes
    dims_kw = dict(replica=dict(appearance_dim=32, use_feat_bank=True), scannet=dict(appearance_dim=16, use_feat_bank=False))[cfg]
    dev = torch.device("cuda:0")
    W, H, A, n_steps, reg = 320, 240, 3000, 6, 0.01
    cam = scenes.make_camera(W, H, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    dims = ng.ModelDims(**dims_kw)
    model = ng.synthetic_model(A, dims, cam, dev, seed=31)
    pose7 = np.array([0.1, -0.05, 0.02, 0.98, 0.05, -0.1, 0.15], dtype=np.float32)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(12))
    p_init = model.params.cpu().numpy().copy()

    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([A, W, H, dims.appearance_dim, int(dims.use_feat_bank), n_steps], np.int32).tofile(f)
        np.array([cam.tanfovx, cam.tanfovy, reg], np.float32).tofile(f)
        for name in ("anchor", "offset", "anchor_feat", "scaling"):
            model.param(name).cpu().numpy().astype(np.float32).tofile(f)
        model.mlp_params.cpu().numpy().astype(np.float32).tofile(f)
        for a in (cam.world_view_transform, cam.full_proj_transform, cam.camera_center, pose7, gt.numpy()):
            np.ascontiguousarray(a, np.float32).tofile(f)
    exe = os.path.join(TB, "trainer_test")
    assert os.path.exists(exe), "build the drop-in layer first (make -C segs-slam_amd/csrc/torch_boundary)"
    subprocess.check_call([exe, str(fin), str(fout)])
    raw = np.fromfile(fout, np.float32)
    losses_cpp, regs_cpp = raw[:n_steps], raw[n_steps:2 * n_steps]
    steps_taken, resident = int(raw[2 * n_steps]), int(raw[2 * n_steps + 1])
    n = model.params.numel()
    p_cpp = raw[2 * n_steps + 2:2 * n_steps + 2 + n]
    img_cpp = raw[2 * n_steps + 2 + n:].reshape(3, H, W)
    assert steps_taken == n_steps and resident == n_steps - 1      # one calibrating pass, the rest without a host sync

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center), t(pose7), cam.tanfovx, cam.tanfovy)
    step = ng.ScaffoldTrainerStep(model, W, H, scaling_reg_weight=reg)
    gtd = gt.to(dev)
    losses_py, regs_py = [], []
    for _ in range(n_steps):
        losses_py.append(float(step.training_once([kf], [gtd])))
        regs_py.append(float(step.neural.scaling_reg))
    torch.cuda.synchronize()
    assert step._mlp_count.value() == n_steps
    np.testing.assert_allclose(losses_cpp, losses_py, rtol=1e-5)
    np.testing.assert_allclose(regs_cpp, regs_py, rtol=1e-5)
    assert losses_py[-1] < losses_py[0]
    img_py = step.engine.out_color.cpu().numpy()
    # (the image is rendered from parameters after five noisy Adam steps on both sides: a statistical bound)
    assert np.abs(img_cpp - img_py).max() <= 2e-2 and np.mean(np.abs(img_cpp - img_py) > 1e-4) < 1e-2
    p_py = model.params.cpu().numpy()
    upd = p_py - p_init
    moved = np.abs(upd) > 0
    err = np.abs(p_cpp - p_py)
    bad = err > 2e-3 * np.abs(upd) + 1e-6
    assert moved.mean() > 0.05 and bad.mean() < 5e-3, (float(moved.mean()), float(bad.mean()), float(err.max()))
    assert err.max() <= 2 * n_steps * 0.08, float(err.max())
