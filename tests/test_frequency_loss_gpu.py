"""The mapper's frequency regulariser on the device (src/gaussian_mapper.cpp:930-945; include/loss_utils.h:126-237): the fused
path (segs-slam_amd/frequency_loss.py, csrc/freq_loss.hip) against the torch.fft + autograd mirror of the reference's op chain
at the sizes the step runs at, and inside the mapper step.  The mirror itself is pinned by the reference's compiled functions
(tests/test_loss_reference.py, small fixtures); this file carries that pin to BASELINE's image sizes.

Tolerances: value 1e-5 relative; dL/dimage 1e-4 relative + 1e-5 of the tensor's largest entry against a float64 evaluation,
with the sign of near-tie frequencies arbitrated as _check_against_float64 describes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(H, W, dev, seed):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(3, H, W, generator=g)
    img = (gt + 0.2 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    img[:, ::7, ::5] = gt[:, ::7, ::5]
    return img.to(dev).contiguous(), gt.to(dev).contiguous()


def _interp64(x, s):
    import torch.nn.functional as F
    if s == 1.0:
        return x
    return F.interpolate(x.unsqueeze(0), scale_factor=(s, s), mode="bilinear", align_corners=False,
                         recompute_scale_factor=True).squeeze(0)


def _check_against_float64(img, gt, lam, scales, val, got):
    """dL/dimage of the device against a float64 evaluation of the reference's formula.

    The regulariser's gradient is discontinuous where a spectrum magnitude crosses its target's: sign(|G_k| - |T_k|) is
    decided by rounding when the two agree to float32 precision, and each such frequency moves the whole gradient image by
    2 w_s.  The float64 side therefore lists its NEAR TIES (| |G_k| - |T_k| | <= 1e-5 |T_k|), and for each takes the device's
    sign if the device's gradient says so (projection of the residual on that frequency's own gradient image, obtained by
    autograd).  Nothing else is adjusted: after that the stated bar applies to every entry of dL/dimage -- 1e-4 relative
    + 1e-5 of its largest entry -- and the value to 1e-5 relative.  Returns (near ties, signs taken from the device).

    Sizes that are not multiples of 4 resize with non-trivial bilinear weights: those copies are taken with float32
    F.interpolate on the device, the op the reference runs (ATen forms the source coordinate scale * (dst + 0.5) - 0.5 in
    float32 for float32 images), and its own backward carries the level gradients to the image; exact 2x / 4x resizes
    (weights 1/2 in any precision) run in float64 throughout."""
    H, W = img.shape[-2:]
    exact = H % 4 == 0 and W % 4 == 0
    src = img.double() if exact else img
    a = src.clone().requires_grad_(True)
    levels = [_interp64(a, s) for s in scales]
    leaves = [x.detach().double().requires_grad_(True) for x in levels]
    tgt = [_interp64(gt.double() if exact else gt, s).double() for s in scales]

    def to_image(level_grads):
        g = torch.autograd.grad(levels, a, [x.to(lv.dtype) for x, lv in zip(level_grads, levels)], retain_graph=True)[0]
        return g.double()

    spectra, total = [], 0.0
    for s, lv, t in zip(scales, leaves, tgt):
        G, T = torch.fft.fft2(lv), torch.fft.fft2(t).abs()
        spectra.append((G, T))
        total = total + lam * s * torch.mean(torch.abs(G.abs() - T))
    ref = float(total.detach())
    assert abs(val - ref) <= 1e-5 * ref
    g_ref = to_image(torch.autograd.grad(total, leaves, retain_graph=True))
    resid = got.double() - g_ref
    ties = taken = 0
    for l, (s, (G, T)) in enumerate(zip(scales, spectra)):
        near = ((G.abs().detach() - T).abs() <= 1e-5 * T).nonzero()
        ties += len(near)
        assert len(near) <= 20 + 1e-4 * T.numel()
        for c, ky, kx in near.tolist():
            term = lam * s * torch.abs(G[c, ky, kx].abs() - T[c, ky, kx]) / T.numel()
            lg = [torch.zeros_like(x) for x in leaves]
            lg[l] = torch.autograd.grad(term, leaves[l], retain_graph=True)[0]
            g_k = to_image(lg)                      # this frequency's share of the gradient; the other sign gives -g_k
            if float((resid * (-2 * g_k)).sum()) > 0.5 * float(((2 * g_k) ** 2).sum()):
                g_ref = g_ref - 2 * g_k
                resid = resid + 2 * g_k
                taken += 1
    tol = 1e-4 * g_ref.abs() + 1e-5 * float(g_ref.abs().max())
    worst = float((resid.abs() - tol).max())
    assert worst <= 0, f"{int((resid.abs() > tol).sum())} entries outside the bar after {taken} of {ties} near ties took the device's sign"
    return ties, taken


_SIZES = [(680, 1200), (1080, 1920), (480, 640), (187, 333), (170, 300), (64, 72)]
# (the piecewise torch.fft form is covered at every size but the largest)
_CASES = [(sz, multi, tf) for tf in (False, True) for multi in (True, False) for sz in _SIZES if not (tf and sz[0] * sz[1] > 700 * 1300)]


@pytest.mark.parametrize("size,multi,torch_fft", _CASES)
def test_fused_frequency_loss_matches_float64_mirror_at_step_sizes(size, multi, torch_fft):
    from segs_slam_amd.frequency_loss import FusedFrequencyLoss
    dev = torch.device("cuda:0")
    H, W = size
    img, gt = _images(H, W, dev, 11 + H)
    lam = 0.01
    scales = (1.0, 0.5, 0.25) if multi else (1.0,)
    fl = FusedFrequencyLoss(H, W, dev, lambda_high=lam, scales=(1.0, 0.5, 0.25), multi_resolution=multi, torch_fft=torch_fft)
    assert fl.folded == (multi and not torch_fft and H % 4 == 0 and W % 4 == 0)
    base = torch.randn(3, H, W, device=dev) * 1e-9          # the L1/SSIM gradient the regulariser is added to
    for _ in range(2):
        dL = base.clone()
        loss_word = torch.full((1,), 0.5, device=dev)
        val = float(fl(img, gt, dL, loss_word))
    torch.cuda.synchronize()
    assert abs(float(loss_word) - 0.5 - val) <= 1e-6 + 1e-5 * val
    ties, taken = _check_against_float64(img, gt, lam, scales, val, dL - base)
    print(f"{W}x{H} multi={multi} torch_fft={torch_fft}: {ties} near ties, {taken} took the device's sign")


def test_folded_and_per_scale_plans_agree():
    """The alias-folded evaluation (one transform pair) against the per-scale evaluation (three pairs) of the same plan API:
    two float32 routes to the same numbers -- value 1e-6, coefficient-implied gradient within the flips' budget."""
    from segs_slam_amd.frequency_loss import FusedFrequencyLoss
    dev = torch.device("cuda:0")
    H, W = 680, 1200
    img, gt = _images(H, W, dev, 3)
    folded = FusedFrequencyLoss(H, W, dev, lambda_high=0.01)
    piecewise = FusedFrequencyLoss(H, W, dev, lambda_high=0.01, torch_fft=True)
    assert folded.folded and not piecewise.folded
    da, db = torch.zeros_like(img), torch.zeros_like(img)
    va, vb = float(folded(img, gt, da)), float(piecewise(img, gt, db))
    assert abs(va - vb) <= 2e-6 * vb
    # L2 distance: each flipped frequency moves the gradient by 2 w_l over the whole image
    rel = float((da - db).norm() / db.norm())
    assert rel < 5e-3, rel


def test_target_spectrum_cache_follows_the_target_tensor():
    """|FFT(gt)| is cached per target tensor (address, version): another keyframe or an in-place edit must not hit a stale entry."""
    from segs_slam_amd.frequency_loss import FusedFrequencyLoss
    dev = torch.device("cuda:0")
    H, W = 96, 128
    img, gt = _images(H, W, dev, 5)
    fl = FusedFrequencyLoss(H, W, dev, lambda_high=1.0, max_cached_targets=2)
    z = lambda: torch.zeros(3, H, W, device=dev)  # noqa: E731
    v0 = float(fl(img, gt, z()))
    assert float(fl(img, gt, z())) == v0 and len(fl._targets) == 1
    gt2 = gt.flip(-1).contiguous()
    v2 = float(fl(img, gt2, z()))
    assert v2 != v0 and len(fl._targets) == 2
    gt.mul_(0.5)                                   # in place: the version counter moves, the entry is re-made
    v3 = float(fl(img, gt, z()))
    assert v3 != v0 and len(fl._targets) == 2      # bounded: the oldest entry went
    fresh = FusedFrequencyLoss(H, W, dev, lambda_high=1.0)
    assert float(fresh(img, gt, z())) == v3
    assert float(fl(img, img.clone(), z())) == 0.0  # identical images: |G| == |T| everywhere, sign(0) = 0
    d = z()
    fl(img, img.clone(), d)
    assert not d.any()


def _step_pair(fused, iteration, dev):
    from segs_slam_amd import mapper_config as mc, neural_gaussians as ng, scenes
    cfg = mc.load_committed_config("cfg/gaussian_mapper/RGB-D/Replica/office0.yaml")
    cam = scenes.make_camera(320, 240, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    model = ng.synthetic_model(4000, cfg.model, cam, dev, seed=3)
    step = mc.make_mapper_step(cfg, model, cam.width, cam.height)
    step.freq_reg["fused"] = fused
    step.iteration = iteration
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.1, 0.2, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    g = torch.Generator().manual_seed(9)
    gt = torch.rand(3, cam.height, cam.width, generator=g).to(dev)
    gt[:, 100:104, :] = 0.0                      # rows the mapper's row mask blanks (src/gaussian_mapper.cpp:917-922)
    return step, kf, gt


@pytest.mark.parametrize("iteration,expect_on", [(10_000, True), (3_000, False), (26_000, False)])
def test_replica_mapper_step_with_fused_frequency_regulariser_matches_the_mirror(iteration, expect_on):
    """One forward/backward of the step the Replica cfg describes (Mapper.use_frequency_regularization 1, 3 scales,
    lambda_high 0.01, window (5 000, 25 500)): loss and every parameter gradient of the fused path against the path that
    evaluates the regulariser with the reference's op chain + autograd."""
    dev = torch.device("cuda:0")
    res = {}
    for fused in (True, False):
        step, kf, gt = _step_pair(fused, iteration, dev)
        assert step._freq_active() == (False, expect_on)
        step.iteration += 0
        loss = step._forward_backward(kf, gt)
        torch.cuda.synchronize()
        res[fused] = (float(loss), step.model.grads.clone(), step)
    (la, ga, sa), (lb, gb, _) = res[True], res[False]
    assert abs(la - lb) <= 1e-5 * abs(lb)
    for name in sa.model.widths:
        a, b = sa.model.grad(name), res[False][2].model.grad(name)
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-4 * scale, name
    if expect_on:
        # and the regulariser is really in there: without it the loss is smaller
        step, kf, gt = _step_pair(True, 3_000, dev)
        assert float(step._forward_backward(kf, gt)) < la - 1e-4


_LIBRARY_ARM = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from segs_slam_amd.frequency_loss import FusedFrequencyLoss
from tests.test_frequency_loss_gpu import _images
H, W = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
img, gt = _images(H, W, dev, 5)
fl = FusedFrequencyLoss(H, W, dev, lambda_high=0.01)
assert fl.folded
dL = torch.zeros_like(img)
val = float(fl(img, gt, dL))
torch.cuda.synchronize()
np.savez(sys.argv[4], val=val, dL=dL.cpu().numpy())
"""


@pytest.mark.parametrize("size", [(680, 1200), (480, 640)])
def test_own_transforms_and_the_vendor_library_fallback_agree(size, tmp_path):
    """The plan's own real transforms (csrc/real_fft.h; the default for sizes whose factors are in {2, 3, 5, 17}) against the same
    plan on the vendor FFT library (SEGS_FREQ_HIPFFT=1, read once per process: a child process): two float32 routes to the same
    numbers, held to the rule of test_folded_and_per_scale_plans_agree -- value 2e-6 relative, gradient within the sign flips'
    budget (L2 distance 5e-3)."""
    import os
    import subprocess
    import sys
    from segs_slam_amd.frequency_loss import FusedFrequencyLoss
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H, W = size
    out = tmp_path / "library.npz"
    env = dict(os.environ, SEGS_FREQ_HIPFFT="1")
    subprocess.check_call([sys.executable, "-c", _LIBRARY_ARM, root, str(H), str(W), str(out)], env=env)
    lib = np.load(out)
    dev = torch.device("cuda:0")
    img, gt = _images(H, W, dev, 5)
    fl = FusedFrequencyLoss(H, W, dev, lambda_high=0.01)
    da = torch.zeros_like(img)
    va = float(fl(img, gt, da))
    assert abs(va - float(lib["val"])) <= 2e-6 * float(lib["val"])
    db = torch.from_numpy(lib["dL"]).to(dev)
    rel = float((da - db).norm() / db.norm())
    assert rel < 5e-3, rel
    # the plan adds into dL/dimage: a second call on the same buffer doubles it (the inverse row pass accumulates)
    fl(img, gt, da)
    assert float((da - 2 * db).norm() / db.norm()) < 1e-2


@pytest.mark.parametrize("size", [(680, 1200), (480, 640), (1080, 1920), (240, 320), (136, 272)])
def test_own_forward_transform_against_a_float64_fft(size):
    """csrc/real_fft.h's forward pass alone (row pass: W reals as W/2 complex + split; column pass over tile-major storage), read
    back through the plan's target table |X| (segs_freq_target: the first table of the block, stored like the spectrum in tiles of 8
    columns, [c][kx // 8][ky][kx % 8]) against |torch.fft.rfft2| in float64.  Sizes: the three shipped ones, a small one, and one
    whose height has the radix-17 factor twice over (136 = 8.17, 272 / 2 = 8.17).
    Bar: a float32 Cooley-Tukey transform of N points errs by about eps log2(N) ||x||_2 per coefficient when no coefficient
    dominates (||x||_2 is the size of a typical coefficient, Parseval) -- 1.2e-6 ||x||_2 at these N; the input is zero-mean noise
    (an image's DC term, hundreds of times ||x||_2, would leave its own rounding, eps |DC|, on the coefficients that share its
    butterflies) and every coefficient must be inside 1e-5 ||x||_2 + 1e-5 |X|."""
    from segs_slam_amd.frequency_loss import FusedFrequencyLoss
    dev = torch.device("cuda:0")
    H, W = size
    g = torch.Generator().manual_seed(H * 7 + W)
    gt = (torch.rand(3, H, W, generator=g) - 0.5).to(dev).contiguous()
    fl = FusedFrequencyLoss(H, W, dev, lambda_high=0.01)
    assert fl.folded
    block = fl.target_block(gt)
    torch.cuda.synchronize()
    wc = W // 2 + 1
    ntiles = (wc + 7) // 8
    t0 = block[:3 * ntiles * H * 8].view(3, ntiles, H, 8).permute(0, 2, 1, 3).reshape(3, H, ntiles * 8)[:, :, :wc].double()
    ref = torch.fft.rfft2(gt.double()).abs()
    norm = float(gt.double().norm() / 3 ** 0.5)          # per channel, on average
    err = (t0 - ref).abs()
    assert bool((err <= 1e-5 * norm + 1e-5 * ref).all()), (float(err.max()), norm)
    assert float(err.max()) > 0                            # (it IS a float32 transform, not the reference looked up)
