"""GPU tests of the optimizer kernel and the trainer step."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def adam_reference(p, g, m, v, lr, b1, b2, eps, step, gscale):
    """float32 restatement of LibTorch's C++ Adam step, one rounding per operation; hyper-parameters
    are doubles and every scalar is formed in double before it is rounded into the float32 tensor arithmetic."""
    f = np.float32
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    step_size = f(lr / bc1)
    sqrt_bc2 = f(np.sqrt(bc2))
    gr = g * f(gscale)
    m = m * f(b1) + gr * f(1.0 - b1)
    v = v * f(b2) + gr * gr * f(1.0 - b2)
    denom = np.sqrt(v) / sqrt_bc2 + f(eps)
    p = p - step_size * (m / denom)
    return p.astype(f), m.astype(f), v.astype(f)


@pytest.mark.parametrize("sizes", [(1000, 7, 64), (3, 1, 5), (4096 * 3 + 1, 1023, 2)])
def test_fused_adam_bit_exact(sizes):
    from segs_slam_amd import _capi
    rng = np.random.default_rng(sum(sizes))
    n = sum(sizes)
    p = rng.standard_normal(n).astype(np.float32)
    g = (rng.standard_normal(n) * 1e-3).astype(np.float32)
    m = (rng.standard_normal(n) * 1e-4).astype(np.float32)
    v = (rng.random(n) * 1e-6).astype(np.float32)
    lrs = [1.6e-4, 5e-3, 1e-3]
    tp, tg, tm, tv = (torch.from_numpy(a.copy()).to(DEV) for a in (p, g, m, v))
    segs = (_capi.AdamSegment * 3)()
    off = 0
    for i, (cnt, lr) in enumerate(zip(sizes, lrs)):
        segs[i].offset, segs[i].count, segs[i].lr = off, cnt, lr
        off += cnt
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = _capi.lib().segs_adam_step(ptr(tp), ptr(tg), ptr(tm), ptr(tv), segs, 3, 0.9, 0.999, 1e-15, 7, 0.5, 1,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _capi.check(st, "segs_adam_step")
    torch.cuda.synchronize()
    off = 0
    for cnt, lr in zip(sizes, lrs):
        sl = slice(off, off + cnt)
        rp, rm, rv = adam_reference(p[sl], g[sl], m[sl], v[sl], lr, 0.9, 0.999, 1e-15, 7, 0.5)
        assert np.array_equal(tm.cpu().numpy()[sl], rm) and np.array_equal(tv.cpu().numpy()[sl], rv)
        assert np.array_equal(tp.cpu().numpy()[sl], rp)
        off += cnt
    assert float(tg.abs().max()) == 0.0  # zero_grad folded in


def test_fused_adam_matches_libtorch_cpp_adam_fixture():
    """Five steps of the fused kernel against LibTorch's C++ torch::optim::Adam configured as src/gaussian_model.cpp:632-640
    does (tests/golden/adam_libtorch.npz, made by tests/golden/make_adam_golden.py with LibTorch 2.10 CPU; the reference
    pins 2.0.1).  LibTorch's CPU kernels may fuse multiply-adds, the HIP kernel is built without contraction: a few ulp."""
    import os
    from segs_slam_amd import _capi
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "adam_libtorch.npz"))
    n = g["p0"].size
    tp = torch.from_numpy(g["p0"].copy()).to(DEV)
    tm, tv = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    segs = (_capi.AdamSegment * 1)()
    segs[0].offset, segs[0].count, segs[0].lr = 0, n, float(g["lr"])
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for s in range(g["grads"].shape[0]):
        tg = torch.from_numpy(g["grads"][s].copy()).to(DEV)
        st = _capi.lib().segs_adam_step(ptr(tp), ptr(tg), ptr(tm), ptr(tv), segs, 1, 0.9, 0.999, 1e-15, s + 1, 1.0, 1,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _capi.check(st, "segs_adam_step")
    torch.cuda.synchronize()
    gmax = np.abs(g["grads"]).max(axis=0)          # exp_avg sums signed terms: rounding scales with the operands
    assert np.all(np.abs(tm.cpu().numpy() - g["exp_avg"]) <= 2e-6 * np.abs(g["exp_avg"]) + 1e-7 * gmax)
    np.testing.assert_allclose(tv.cpu().numpy(), g["exp_avg_sq"], rtol=2e-6, atol=1e-20)
    np.testing.assert_allclose(tp.cpu().numpy(), g["p"], rtol=0, atol=2e-6 * float(g["lr"]) * 5 + 1e-7)
    moved = np.abs(g["p"] - g["p0"]) > 0
    assert moved.sum() > 0.9 * n * 16 / 17 and not moved[::17].any()


def test_guarded_adam_skips_on_device_flag():
    """segs_adam_step_guarded: a non-zero device word drops the step (parameters and moments untouched, gradients cleared);
    a zero word gives exactly segs_adam_step."""
    from segs_slam_amd import _capi
    rng = np.random.default_rng(5)
    n = 10007
    arrs = [rng.standard_normal(n).astype(np.float32) for _ in range(3)] + [rng.random(n).astype(np.float32)]
    segs = (_capi.AdamSegment * 1)()
    segs[0].offset, segs[0].count, segs[0].lr = 0, n, 1e-3
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for flag in (None, 0, 1):
        tp, tg, tm, tv = (torch.from_numpy(a.copy()).to(DEV) for a in arrs)
        word = torch.tensor([flag or 0], dtype=torch.int32, device=DEV)
        if flag is None:
            st = _capi.lib().segs_adam_step(ptr(tp), ptr(tg), ptr(tm), ptr(tv), segs, 1, 0.9, 0.999, 1e-15, 3, 1.0, 1, stream)
        else:
            st = _capi.lib().segs_adam_step_guarded(ptr(tp), ptr(tg), ptr(tm), ptr(tv), segs, 1, 0.9, 0.999, 1e-15, 3, 1.0, 1,
                                                    ptr(word), stream)
        _capi.check(st, "segs_adam_step")
        torch.cuda.synchronize()
        out[flag] = [t.cpu().numpy() for t in (tp, tg, tm, tv)]
    for a, b in zip(out[None], out[0]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[1][0], arrs[0]) and np.array_equal(out[1][2], arrs[2]) and np.array_equal(out[1][3], arrs[3])
    assert float(np.abs(out[1][1]).max()) == 0.0


def test_fused_adam_matches_torch_optim():
    """Same step through torch.optim.Adam (the Python front end updates exp_avg with lerp_, a different association):
    agreement to a few ulp of the parameter (|p| ~ 1) -- 2e-6 relative + 5e-7 absolute."""
    from segs_slam_amd.gaussian_trainer import FusedAdam, OptimizationParams
    from segs_slam_amd.raster_engine import FIELDS, FLOATS_PER_GAUSSIAN
    P = 777
    opt = OptimizationParams()
    gen = torch.Generator().manual_seed(1)
    p0 = torch.randn(FLOATS_PER_GAUSSIAN * P, generator=gen)
    adam = FusedAdam(p0.numel(), DEV, opt)
    pg = p0.clone().to(DEV)
    ref_params, off = [], 0
    lrs = {"means3D": 1e-4, "scales": 5e-3, "rotations": 1e-3, "opacity": 5e-2, "colors": 2.5e-3}
    for name, n in FIELDS:
        ref_params.append(torch.nn.Parameter(p0[off:off + P * n].clone()))
        off += P * n
    topt = torch.optim.Adam([{"params": [q], "lr": lrs[name]} for q, (name, _) in zip(ref_params, FIELDS)], eps=1e-15)
    for it in range(3):
        g = torch.randn(p0.numel(), generator=gen) * 1e-3
        off = 0
        for q, (name, n) in zip(ref_params, FIELDS):
            q.grad = g[off:off + P * n].clone()
            off += P * n
        topt.step()
        adam.step(pg, g.to(DEV), lrs, P, 1.0)
    ref = torch.cat([q.detach() for q in ref_params])
    assert torch.allclose(pg.cpu(), ref, rtol=2e-6, atol=5e-7)


def test_trainer_step_reduces_loss():
    """render -> L1 + 0.2 (1-SSIM) -> raster backward -> fused Adam, a few steps on one keyframe: the loss must go down.
    (That the step's loss, gradients and update equal the oracle chain's is the subject of tests/test_step_parity_gpu.py.)"""
    from oracle import gs_oracle
    from segs_slam_amd import scenes
    from segs_slam_amd.gaussian_trainer import TrainerStep, keyframe_tensors
    sc = scenes.make_scene(20_000, 160, 120, 130.0, 130.0, seed=5, bg=(0.0, 0.0, 0.0))
    sc.scales *= 2.5
    # target = render of a perturbed copy of the scene
    tgt = scenes.make_scene(20_000, 160, 120, 130.0, 130.0, seed=5)
    tgt.scales *= 2.5
    tgt.colors[:] = np.clip(tgt.colors + 0.3 * (scenes.uniform01(tgt.colors.size, 70, 5).reshape(tgt.colors.shape) - 0.5), 0, 1)
    o, _ = gs_oracle.run_scene(tgt, backward=False)
    gt = torch.from_numpy(o.get("out_color")).to(DEV)
    step = TrainerStep.on_gpu(sc, DEV)
    kf = keyframe_tensors(sc.camera, DEV)
    losses = [float(step.training_once([kf], [gt])) for _ in range(8)]
    assert losses[-1] < losses[0], losses
    assert all(np.isfinite(losses))


def test_trainer_step_runs_a_dropped_iteration_again():
    """An iteration whose forward overflowed the resident capacity is dropped on the device; with one rank the host runs it again
    before the next one (the reference's trainer never skips an iteration, src/gaussian_trainer.cpp:47-117)."""
    from segs_slam_amd import scenes
    from segs_slam_amd.gaussian_trainer import TrainerStep, keyframe_tensors
    counts = {}
    for redo in (True, False):
        sc = scenes.make_scene(20_000, 160, 120, 130.0, 130.0, seed=5, bg=(0.0, 0.0, 0.0))
        sc.scales *= 2.5
        step = TrainerStep.on_gpu(sc, DEV)
        step.redo_dropped_steps = redo
        kf = keyframe_tensors(sc.camera, DEV)
        gt = torch.rand(3, 120, 160, generator=torch.Generator().manual_seed(1)).to(DEV)
        for it in range(6):
            if it == 3:
                assert step.engine.check() and step.engine.capacity > 0
                step.engine.capacity = max(step.engine.R // 3, 1024)     # the next forward overflows
            step.training_once([kf], [gt])
        torch.cuda.synchronize()
        counts[redo] = (step.optimizer.count.value(), step.optimizer.count.dropped(), getattr(step, "redone_steps", 0))
    assert counts[True] == (6, 1, 1), counts
    assert counts[False] == (5, 1, 0), counts


@pytest.mark.parametrize("H,W", [(48, 64), (37, 53), (680, 1200)])
def test_fused_l1_ssim_matches_loss_utils(H, W):
    """Fused HIP loss (forward value, L1, SSIM and dL/dimage) vs the reference's op chain (loss_utils mirror + autograd)."""
    from segs_slam_amd import loss_utils
    from segs_slam_amd.gaussian_trainer import FusedL1SSIM
    gen = torch.Generator().manual_seed(H * W)
    gt = torch.rand(3, H, W, generator=gen).to(DEV)
    img = (gt + 0.2 * torch.randn(3, H, W, generator=gen).to(DEV)).clamp(0, 1).contiguous()
    img[:, : H // 4] = gt[:, : H // 4]  # exact-equality region: sign(0) = 0 in the L1 gradient
    lam = 0.2
    fused = FusedL1SSIM(H, W, DEV, lam)
    loss, dL = fused(img, gt)
    x = img.clone().requires_grad_(True)
    l1 = loss_utils.l1_loss(x, gt)
    ss = loss_utils.ssim(x, gt)
    ref = (1.0 - lam) * l1 + lam * (1.0 - ss)
    (g,) = torch.autograd.grad(ref, x)
    torch.cuda.synchronize()
    assert abs(float(fused.out[1]) - float(l1)) < 1e-6 and abs(float(fused.out[2]) - float(ss)) < 2e-6
    assert abs(float(loss) - float(ref)) < 2e-6
    err = (dL - g).abs().max().item()
    assert err <= 2e-5 * g.abs().max().item() + 1e-12, (err, g.abs().max().item())
