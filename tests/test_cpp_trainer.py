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
    for cls in ("segs_host::AnchorDensifier::adjust_anchor(", "segs_host::AnchorDensifier::training_statis(",
                "segs_host::KeyframeExchange::reduce_gradients(", "segs_host::KeyframeExchange::gather("):
        assert any(cls in l and " T " in l for l in out.splitlines()), cls
    for sym in ("segs_visible_filter_log_scales", "segs_neural_forward", "segs_rasterize_forward_resident", "segs_l1_ssim_loss",
                "segs_rasterize_backward_resident", "segs_neural_backward", "segs_adam_step_device", "segs_training_statis_guarded",
                "segs_anchor_growing_level"):
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
    # the same run with the per-Gaussian projection and the prefilter as kernels of their own (SURVEY 8f n3 off): the forward
    # is bit-identical either way, so the first losses agree to rounding of the loss reduction and the rest to the atomics' noise
    subprocess.check_call([exe, str(fin), str(tmp_path / "out_unfused.bin")], env=dict(os.environ, SEGS_TRAINER_TEST_UNFUSED="1"))
    raw_u = np.fromfile(tmp_path / "out_unfused.bin", np.float32)
    assert raw[0] == raw_u[0]                                 # (iteration 1 calibrates on both sides; from iteration 2 on the forward is fused)
    np.testing.assert_allclose(raw[:n_steps], raw_u[:n_steps], rtol=1e-4)
    # an iteration dropped on the device (forced: the resident capacity is shrunk in front of iteration 3) is run again by the host
    # before the next one; with the redo switched off the optimizer has taken one step fewer
    for env, want_steps, want_redone in ((dict(SEGS_TRAINER_TEST_OVERFLOW_AT="3"), n_steps, 1),
                                         (dict(SEGS_TRAINER_TEST_OVERFLOW_AT="3", SEGS_TRAINER_TEST_NO_REDO="1"), n_steps - 1, 0)):
        out = subprocess.run([exe, str(fin), str(tmp_path / "out_drop.bin")], env=dict(os.environ, **env), capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        raw_d = np.fromfile(tmp_path / "out_drop.bin", np.float32)
        assert int(raw_d[2 * n_steps]) == want_steps and f"redone {want_redone}" in out.stdout, (env, out.stdout)
        if want_redone:
            keep = [i for i in range(n_steps) if i != 3]
            np.testing.assert_allclose(raw_d[:n_steps][keep], raw[:n_steps][keep], rtol=2e-4)
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


# ---- the mapper loop in C++: densification (anchor_densifier.{h,cpp}) and the keyframe-parallel exchange (keyframe_exchange.{h,cpp})
def _mapper_case(tmp_path, n_steps=12, A=3000, W=320, H=240, seed=3):
    from segs_slam_amd import neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    cams, kfs = [], []
    for k in range(2):
        ang = 0.04 * (k + 1)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=np.float32)
        cam = scenes.make_camera(W, H, 300.0, 300.0, R, np.array([0.05 * k, 0.0, 0.0], dtype=np.float32))
        pose7 = np.array([0.05 * k, 0.0, 0.0, 1.0, 0.0, ang, 0.0], dtype=np.float32)
        cams.append((cam, pose7))
    dims = ng.ModelDims()
    model = ng.synthetic_model(A, dims, cams[0][0], dev, seed=7)
    gts = [np.full((3, H, W), 0.3 + 0.2 * k, dtype=np.float32) for k in range(2)]
    dp = dict(voxel_size=0.01, start_stat=2, update_from=5, update_interval=10, densify_grad_threshold=1e-7)
    fin = tmp_path / "mapper_in.bin"
    with open(fin, "wb") as f:
        np.array([A, W, H, dims.appearance_dim, int(dims.use_feat_bank), n_steps, 2, dp["start_stat"], dp["update_from"],
                  dp["update_interval"], seed], np.int32).tofile(f)
        np.array([cams[0][0].tanfovx, cams[0][0].tanfovy, 0.0, dp["voxel_size"], dp["densify_grad_threshold"]], np.float32).tofile(f)
        for name in ("anchor", "offset", "anchor_feat", "scaling"):
            model.param(name).cpu().numpy().astype(np.float32).tofile(f)
        model.mlp_params.cpu().numpy().astype(np.float32).tofile(f)
        for (cam, pose7), gt in zip(cams, gts):
            for a in (cam.world_view_transform, cam.full_proj_transform, cam.camera_center, pose7, gt):
                np.ascontiguousarray(a, np.float32).tofile(f)
    return dict(fin=fin, model=model, dims=dims, cams=cams, gts=gts, dp=dp, n_steps=n_steps, A=A, W=W, H=H, seed=seed, dev=dev)


def _read_mapper_out(path, n_steps, dims):
    raw = np.fromfile(path, np.uint8)
    head = raw[:16].view(np.int32)
    A, capacity, mlp_steps, anchor_steps = (int(v) for v in head)
    off = 16
    sizes = raw[off:off + 4 * n_steps].view(np.int32).copy(); off += 4 * n_steps
    fl = raw[off:].view(np.float32)
    out = dict(A=A, capacity=capacity, steps=(mlp_steps, anchor_steps), sizes=sizes, losses=fl[:n_steps].copy())
    p = n_steps
    no = dims.n_offsets
    for name, w in (("anchor", 3), ("offset", 3 * no), ("anchor_feat", dims.feat_dim), ("scaling", 6)):
        out[name] = fl[p:p + A * w].copy(); p += A * w
    n_stats = 2 * A + 2 * A * no
    out["mlp"] = fl[p:fl.size - n_stats].copy()
    p = fl.size - n_stats
    for name, n in (("opacity_accum", A), ("anchor_demon", A), ("offset_gradient_accum", A * no), ("offset_denom", A * no)):
        out[name] = fl[p:p + n].copy(); p += n
    return out


@pytest.mark.gpu
def test_cpp_mapper_loop_grows_and_prunes_like_the_python_loop(tmp_path):
    """trainingOnce with densification in C++ (statistics every iteration of the window, adjust_anchor at iteration 10, the Adam
    state migrated with the rows, the anchor groups skipping that iteration's step) against the Python loop on the same model,
    keyframes and random keep masks (same LibTorch CPU generator stream on both sides).  The statistics are float sums of
    atomically accumulated gradients, so the two runs agree the way two runs of either twin do: the map size to 0.5 %, the
    integer counters exactly, the losses to 1e-4 before the map changes."""
    from segs_slam_amd import densify, neural_gaussians as ng
    c = _mapper_case(tmp_path)
    exe = os.path.join(TB, "trainer_test")
    fout = tmp_path / "mapper_out.bin"
    subprocess.check_call([exe, "--mapper", str(c["fin"]), str(fout)])
    cpp = _read_mapper_out(fout, c["n_steps"], c["dims"])

    dev = c["dev"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kfs = [ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center), t(p7), cam.tanfovx, cam.tanfovy)
           for cam, p7 in c["cams"]]
    gts = [t(g) for g in c["gts"]]
    step = ng.ScaffoldTrainerStep(c["model"], c["W"], c["H"])
    dp = c["dp"]
    dens = densify.AnchorDensifier(c["model"], densify.DensifyParams(voxel_size=dp["voxel_size"], start_stat=dp["start_stat"],
                                                                      update_from=dp["update_from"], update_interval=dp["update_interval"],
                                                                      update_until=10 ** 9, densify_grad_threshold=dp["densify_grad_threshold"]))
    step.enable_densification(dens, seed=c["seed"])
    losses, sizes = [], []
    for _ in range(c["n_steps"]):
        losses.append(float(step.training_once(kfs, gts)))
        sizes.append(c["model"].A)
    torch.cuda.synchronize()
    assert cpp["steps"] == (c["n_steps"], c["n_steps"] - 1) == (step._mlp_count.value(), step._anchor_count.value())
    assert list(cpp["sizes"][:9]) == sizes[:9] == [c["A"]] * 9 and cpp["sizes"][-1] != c["A"]
    assert abs(int(cpp["sizes"][-1]) - sizes[-1]) <= 0.005 * sizes[-1], (cpp["sizes"][-1], sizes[-1])
    np.testing.assert_allclose(cpp["losses"][:10], losses[:10], rtol=1e-4)
    np.testing.assert_allclose(cpp["losses"][10:], losses[10:], rtol=2e-2)
    assert cpp["capacity"] >= cpp["A"] and np.isfinite(cpp["mlp"]).all() and np.isfinite(cpp["anchor"]).all()
    # integer-valued view counters (reset only for the anchors seen in more than 80 % of the window): same set of values
    m = c["model"]
    assert np.array_equal(np.unique(cpp["anchor_demon"]), np.unique(dens.stat("anchor_demon").cpu().numpy()))
    assert abs(float(cpp["anchor_demon"].sum()) - float(dens.stat("anchor_demon").sum())) <= 0.01 * float(cpp["anchor_demon"].sum())
    # the MLP block took the same 12 steps from the same start: close wherever the update is more than noise
    mlp_py = m.mlp_params.cpu().numpy()
    assert np.mean(np.abs(cpp["mlp"] - mlp_py) > 2e-2 * np.abs(mlp_py) + 1e-3) < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["dense", "sharded"])
def test_two_cpp_ranks_keep_bit_identical_replicas_through_adjust_anchor(tmp_path, mode):
    """Two processes of trainer_test (both on cuda:0) as keyframe-parallel ranks: each renders its own keyframe, the overflow
    word and the gradient bucket are exchanged through KeyframeExchange (over the test's store-backed c10d::Backend: host
    staging, rank-ordered sums), the densification statistics are summed over ranks before adjust_anchor.  Everything the
    ranks hold must be bit-identical afterwards, in both exchange modes."""
    c = _mapper_case(tmp_path)
    exe = os.path.join(TB, "trainer_test")
    store = tmp_path / f"store_{mode}"
    procs = []
    for r in range(2):
        cmd = [exe, "--mapper", str(c["fin"]), str(tmp_path / f"out_{mode}_{r}.bin"), "--world", "2", "--rank", str(r), "--store", str(store)]
        if mode == "dense":
            cmd.append("--dense")
        procs.append(subprocess.Popen(cmd))
    assert [p.wait(timeout=300) for p in procs] == [0, 0]
    r0, r1 = (_read_mapper_out(tmp_path / f"out_{mode}_{r}.bin", c["n_steps"], c["dims"]) for r in range(2))
    assert r0["steps"] == r1["steps"] == (c["n_steps"], c["n_steps"] - 1)
    assert r0["A"] == r1["A"] != c["A"] and list(r0["sizes"]) == list(r1["sizes"]) and r0["sizes"][8] == c["A"]
    for k in ("anchor", "offset", "anchor_feat", "scaling", "mlp", "opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        assert np.array_equal(r0[k], r1[k]), k
    assert not np.array_equal(r0["losses"], r1["losses"])          # (each rank reports its own keyframe's loss)
    # two views per iteration of the window (iterations 3..12) are counted, and the counters are whole numbers
    assert 10.0 < r0["anchor_demon"].max() <= 2.0 * (c["n_steps"] - 2) and np.array_equal(r0["anchor_demon"], np.round(r0["anchor_demon"]))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["dense", "sharded"])
def test_two_cpp_ranks_run_an_iteration_again_when_one_of_them_overflowed(tmp_path, mode):
    """Rank 1's forward overflows its resident capacity at loop index 5 (rank 0's does not): both devices drop that pass (the
    summed overflow word guards statistics and optimizer), both hosts read the summed word from their pinned mirror at the top
    of the next trainingOnce and run iteration 6 again together.  No optimizer step is lost (the reference never skips an
    iteration, src/gaussian_mapper.cpp:1027-1030) and the replicas stay bit-identical, through the adjust_anchor that follows."""
    c = _mapper_case(tmp_path)
    exe = os.path.join(TB, "trainer_test")
    store = tmp_path / f"store_redo_{mode}"
    env = dict(os.environ, SEGS_TRAINER_TEST_OVERFLOW_AT="5", SEGS_TRAINER_TEST_OVERFLOW_RANK="1")
    procs = []
    for r in range(2):
        cmd = [exe, "--mapper", str(c["fin"]), str(tmp_path / f"redo_{mode}_{r}.bin"), "--world", "2", "--rank", str(r), "--store", str(store)]
        if mode == "dense":
            cmd.append("--dense")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert all(o.strip().endswith("redone 1") for o in outs), outs            # BOTH ranks ran it again, once
    r0, r1 = (_read_mapper_out(tmp_path / f"redo_{mode}_{r}.bin", c["n_steps"], c["dims"]) for r in range(2))
    assert r0["steps"] == r1["steps"] == (c["n_steps"], c["n_steps"] - 1)     # every optimizer step of the sequence was taken
    assert r0["A"] == r1["A"] != c["A"] and list(r0["sizes"]) == list(r1["sizes"])
    for k in ("anchor", "offset", "anchor_feat", "scaling", "mlp", "opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        assert np.array_equal(r0[k], r1[k]), k
    assert 10.0 < r0["anchor_demon"].max() <= 2.0 * (c["n_steps"] - 2)         # the dropped pass left no statistics behind


@pytest.mark.gpu
def test_cpp_exchange_through_a_one_rank_rccl_group(tmp_path):
    """The same loop with a ONE-rank c10d::ProcessGroupNCCL constructed in C++ and every collective forced: the overflow word's
    all-reduce, reduce_scatter -> Adam on the shard -> all_gather (and the dense all-reduce of the adjust_anchor iteration)
    run through RCCL from the C++ host.  With one rank they are identities: the run must behave like the plain one."""
    c = _mapper_case(tmp_path)
    exe = os.path.join(TB, "trainer_test")
    plain, nccl = tmp_path / "plain.bin", tmp_path / "nccl.bin"
    subprocess.check_call([exe, "--mapper", str(c["fin"]), str(plain)])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.check_call([exe, "--mapper", str(c["fin"]), str(nccl), "--nccl1", str(tmp_path / "nccl_store")], env=env)
    a, b = (_read_mapper_out(p, c["n_steps"], c["dims"]) for p in (plain, nccl))
    assert a["steps"] == b["steps"] == (c["n_steps"], c["n_steps"] - 1)
    assert list(a["sizes"][:9]) == list(b["sizes"][:9]) and abs(a["A"] - b["A"]) <= 0.005 * a["A"]
    np.testing.assert_allclose(a["losses"][:10], b["losses"][:10], rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,freq", [("replica", True), ("scannet", False)])
def test_cpp_trainer_matches_the_oracle_chain_directly(tmp_path, cfg, freq):
    """The C++ host against the float64 / oracle chain itself (tests/test_step_parity_gpu.py::ReferenceScaffoldStep), not
    through the Python step -- an error common to both hosts' use of the C ABI would pass a host-against-host comparison.
    `trainer_test --chain` runs two iterations and dumps, per iteration, the gradient bucket as its optimizer received it and
    the parameters afterwards.  First iteration: anchor visibility exact, loss 1e-5, image and every bucket gradient at the
    raster bar against the chain.  Both iterations: torch.optim.Adam fed the C++ host's own bucket gradient must land on its
    parameters (every live entry, 1e-3 of the update: parameters are stored in float32, a 5e-3 step on a 0.5 value carries 1e-5 of
    rounding by itself).  With `freq`, the Replica configuration's frequency regulariser is on
    in C++ (segs_freq_* plan API) and is added to the chain with the torch.fft mirror + float64 autograd."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_step_parity_gpu import DIMS, ReferenceScaffoldStep, _grad_check
    from segs_slam_amd import loss_utils, neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    W, H, A, n_steps, reg = 320, 240, 3000, 2, 0.01
    cam = scenes.make_camera(W, H, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    dims = ng.ModelDims(**DIMS[cfg])
    model = ng.synthetic_model(A, dims, cam, dev, seed=21)
    pose7 = (0.1, -0.05, 0.02, 0.98, 0.05, -0.1, 0.15)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(11))
    lam_f = 0.01 if freq else 0.0

    opt = ng.ScaffoldOptimizationParams()
    ref = ReferenceScaffoldStep(model, DIMS[cfg], cam, opt.lambda_dssim, reg)
    ref.sync_params(model)
    loss_ref, image_ref, unstable = ref.forward(gt, pose7)

    fin, fout = tmp_path / "chain_in.bin", tmp_path / "chain_out.bin"
    with open(fin, "wb") as f:
        np.array([A, W, H, dims.appearance_dim, int(dims.use_feat_bank), n_steps], np.int32).tofile(f)
        np.array([cam.tanfovx, cam.tanfovy, reg], np.float32).tofile(f)
        for name in ("anchor", "offset", "anchor_feat", "scaling"):
            model.param(name).cpu().numpy().astype(np.float32).tofile(f)
        model.mlp_params.cpu().numpy().astype(np.float32).tofile(f)
        for a in (cam.world_view_transform, cam.full_proj_transform, cam.camera_center, np.array(pose7, np.float32), gt.numpy(),
                  (~unstable).float().numpy()):
            np.ascontiguousarray(a, np.float32).tofile(f)
        np.array([0, 1000], np.int32).tofile(f)
        np.array([lam_f], np.float32).tofile(f)
    subprocess.check_call([os.path.join(TB, "trainer_test"), "--chain", str(fin), str(fout)])
    raw = np.fromfile(fout, np.float32)
    n = model.params.numel()
    scal = raw[:3 * n_steps].reshape(n_steps, 3)
    pos = 3 * n_steps
    g_cpp, p_cpp = [], []
    for _ in range(n_steps):
        g_cpp.append(raw[pos:pos + n]); p_cpp.append(raw[pos + n:pos + 2 * n]); pos += 2 * n
    radii_cpp = raw[pos:pos + A].astype(np.int32); pos += A
    img_cpp = raw[pos:pos + 3 * H * W].reshape(3, H, W)

    # ---- first iteration against the chain
    if freq:
        # The regulariser of src/gaussian_mapper.cpp:938-945 in float64 (torch.fft mirror + autograd), evaluated AT THE C++
        # HOST'S IMAGE: its gradient is discontinuous wherever a spectrum magnitude crosses the target's (frequency_loss.py),
        # and the oracle's image differs from the device's by up to 1e-4 -- enough to flip the sign of a handful of the
        # 230 k frequencies, each of which moves dL/dimage by 0.6 % everywhere.  The rest of the chain stays at the oracle's image.
        img64 = torch.from_numpy(img_cpp.copy()).double().requires_grad_(True)
        fl = lam_f * loss_utils.multi_scale_loss(img64, gt.double(), (1.0, 0.5, 0.25))
        (gfl,) = torch.autograd.grad(fl, img64)
        ref.dL = ref.dL + gfl
        loss_ref += float(fl)
        assert abs(float(scal[0, 2]) - float(fl)) <= 1e-5 * float(fl)
    grads_ref = ref.backward()
    assert np.array_equal(radii_cpp, ref.radii), "prefilter_voxel radii differ"
    total = float(scal[0, 0]) + float(scal[0, 1])      # trainingOnce's word holds L1/SSIM (+ the frequency term); the scaling regulariser is separate
    assert abs(total - loss_ref) <= 1e-5 * abs(loss_ref), (total, loss_ref)
    ok = ~unstable.numpy()
    assert np.all(np.abs(img_cpp - image_ref.numpy())[:, ok] <= 1e-4 * np.abs(image_ref.numpy())[:, ok] + 2e-5)
    g0 = torch.from_numpy(g_cpp[0].copy())
    for name in list(ref.p) + list(ref.mlp):
        got = model._view(g0, name).numpy()
        _grad_check(f"{name} (C++ host)", got, grads_ref[name].reshape(got.shape))

    # ---- both iterations: the optimizer, fed the C++ host's own gradient
    adam = ReferenceScaffoldStep(model, DIMS[cfg], cam, opt.lambda_dssim, reg)
    step_lr = ng.ScaffoldTrainerStep(model, W, H, opt)        # (only for learning_rates(): the schedule of updateLearningRate)
    p_before = model.params.cpu().numpy().copy()
    live = np.zeros(n, dtype=bool)
    for name in ("anchor", "offset", "anchor_feat", "scaling"):
        o, cnt = model.segments[name]
        live[o:o + cnt] = True
    live[model.mlp_offset:] = True
    for it in range(n_steps):
        adam.sync_params_from_flat(model, p_before)
        gt_it = torch.from_numpy(g_cpp[it].copy())
        adam.adam({nm: model._view(gt_it, nm).numpy() for nm in list(adam.p) + list(adam.mlp)}, step_lr.learning_rates(it + 1))
        p_ref = np.zeros(n, np.float32)
        v = torch.from_numpy(p_ref)
        for nm in adam.p:
            model._view(v, nm).copy_(adam.p[nm].reshape(model._view(v, nm).shape))
        for nm in adam.mlp:
            model._view(v, nm).copy_(adam.mlp[nm])
        upd_cpp, upd_ref = (p_cpp[it] - p_before)[live], (p_ref - p_before)[live]
        off = np.abs(upd_cpp - upd_ref) > 1e-3 * np.abs(upd_ref) + 1.2e-7 * np.abs(p_before[live]) + 1e-9    # (+ one float32 ulp of the parameter itself)
        if off.any():
            idx = np.flatnonzero(live)[off]
            seg = {nm: int(((idx >= o) & (idx < o + c)).sum()) for nm, (o, c) in list(model.segments.items()) + [("mlp", (model.mlp_offset, model.mlp_total))]}
            worst = np.argmax(np.abs(upd_cpp - upd_ref))
            raise AssertionError((it, int(off.sum()), seg, float(upd_cpp[worst]), float(upd_ref[worst])))
        assert float((upd_ref != 0).mean()) > 0.7
        p_before = p_cpp[it].copy()


@pytest.mark.gpu
def test_cpp_frequency_target_cache_survives_recycled_addresses_and_in_place_refreshes(tmp_path):
    """The C++ host's |FFT(target)| cache under the reference mapper's host pattern (src/gaussian_mapper.cpp:845,921: a fresh
    target tensor per iteration): a freed target whose address the allocator hands to the next image, and a staging tensor
    refreshed in place, must both get their OWN tables.  trainer_test --freq-cache re-evaluates the regulariser after every
    iteration with a table made from the target it handed over; the step's value must be that value bit for bit."""
    from segs_slam_amd import neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    W, H, A = 320, 240, 2000
    cam = scenes.make_camera(W, H, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    dims = ng.ModelDims(appearance_dim=32, use_feat_bank=True)
    model = ng.synthetic_model(A, dims, cam, dev, seed=33)
    pose7 = np.array([0.1, -0.05, 0.02, 0.98, 0.05, -0.1, 0.15], dtype=np.float32)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(14))
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([A, W, H, dims.appearance_dim, int(dims.use_feat_bank), 3], np.int32).tofile(f)
        np.array([cam.tanfovx, cam.tanfovy, 0.01], np.float32).tofile(f)
        for name in ("anchor", "offset", "anchor_feat", "scaling"):
            model.param(name).cpu().numpy().astype(np.float32).tofile(f)
        model.mlp_params.cpu().numpy().astype(np.float32).tofile(f)
        for a in (cam.world_view_transform, cam.full_proj_transform, cam.camera_center, pose7, gt.numpy()):
            np.ascontiguousarray(a, np.float32).tofile(f)
    out = subprocess.check_output([os.path.join(TB, "trainer_test"), "--freq-cache", str(fin), str(fout)], text=True)
    assert "second target at the first one's address: 0" in out, out    # the cache entry owns target A, so B cannot land on it
    v = np.fromfile(fout, np.float32).reshape(3, 2)
    assert np.all(v > 0)
    assert np.array_equal(v[:, 0], v[:, 1]), v
    assert v[0, 0] != v[1, 0] and v[1, 0] != v[2, 0], v                  # the three targets are different images to the loss
