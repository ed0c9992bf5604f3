"""Step-level parity of the trainer / mapper iteration on the GPU.

Reference chain (TEST ONLY; each piece is the restatement that already pins its own kernel):
    prefilter_voxel          oracle/gs_oracle.visible_filter             src/gaussian_renderer.cpp:131-199
    generate_neural_gaussians oracle/neural_ref.py, float64 autograd      src/gaussian_renderer.cpp:214-334
    rasterizer fwd / bwd     oracle/gs_oracle (CPU restatement of K1-K13) cuda_rasterizer/*.cu
    loss                     segs-slam_amd/loss_utils.py (pinned by the reference's own include/loss_utils.h,
                             tests/golden/loss_reference.npz) + 0.01 * scaling.prod(1).mean()
                                                                          src/gaussian_mapper.cpp:917-928, src/gaussian_trainer.cpp:89-90
    optimizer                torch.optim.Adam(eps=1e-15), one group per tensor with the step's learning rates
                                                                          src/gaussian_model.cpp:620-690, src/gaussian_mapper.cpp:1027-1030
against ScaffoldTrainerStep.training_once / TrainerStep.training_once (the HIP path through the C ABI).

Tolerances: loss 1e-5 relative; bucket gradient 1e-4 relative with a floor of 1e-5 * max|ref| per tensor (float atomics sum in
arbitrary order, the oracle in double); parameter UPDATE after two steps 1e-3 relative wherever both steps' gradients are more
than rounding noise (Adam with eps 1e-15 turns noise into full-size steps; the first step alone only tests signs).  As in the
raster tests, dL/dimage is zeroed on BOTH sides on the few pixels whose compositing decisions sit within 1e-5 of a threshold
(the oracle reports them; v_exp_f32 and glibc expf differ in the last bits there) -- here together with every pixel whose
11x11 SSIM window contains one, because the loss gradient of a pixel reads the image over that window.
"""
import numpy as np
import pytest
import torch

from oracle import gs_oracle, neural_ref

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _kf(cam, dev, pose7=(0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0)):
    from segs_slam_amd import neural_gaussians as ng
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    return ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                       torch.tensor(pose7, dtype=torch.float32, device=dev), cam.tanfovx, cam.tanfovy)


def _grad_check(name, got, ref, rel=1e-4, floor=1e-5):
    """Same bar as tests/test_raster_gpu.py::assert_grad_close (where the floor is derived from measurements)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    tol = rel * np.abs(ref) + floor * (np.abs(ref).max() + 1e-300)
    bad = err > tol
    assert not bad.any(), (name, int(bad.sum()), bad.size, float(err.max()), float(np.abs(ref).max()))
    nz = ref != 0
    if nz.sum() >= 1000:
        assert float((err[nz] <= rel * np.abs(ref[nz])).mean()) >= 0.98, (name, float((err[nz] <= rel * np.abs(ref[nz])).mean()))


def _dilate(unstable: torch.Tensor) -> torch.Tensor:
    """Pixels whose 11x11 SSIM window (include/loss_utils.h:51-124) contains an unstable pixel: there the two sides' images may
    legitimately differ, and with them dL/dimage of every pixel of the window."""
    return torch.nn.functional.max_pool2d(unstable[None, None].float(), 11, stride=1, padding=5)[0, 0] > 0


class ReferenceScaffoldStep:
    """The float64 / oracle chain of one mapper iteration over CPU copies of the model's tensors."""

    GROUPS = (("anchor", "anchor"), ("offset", "offset"), ("anchor_feat", "anchor_feat"), ("scaling", "scaling"))

    def __init__(self, model, dims_kw, cam, lambda_dssim, reg_weight):
        self.rd = neural_ref.NeuralDims(**dims_kw)
        self.cam, self.lam, self.reg_w = cam, lambda_dssim, reg_weight
        self.p = {n: model.param(n).detach().cpu().clone() for n in ("anchor", "offset", "anchor_feat", "scaling")}
        self.mlp = {n: model.param(n).detach().cpu().clone() for n in model.mlp_layout}
        self.rotation = torch.nn.functional.normalize(model.rotation[:model.A].cpu())
        self.opt = None

    def sync_params(self, model):
        """Start the iteration from the GPU model's current parameters (bit-identical inputs on both sides; the optimizer
        state of this chain stays its own)."""
        for n in self.p:
            self.p[n] = model.param(n).detach().cpu().clone()
        for n in self.mlp:
            self.mlp[n] = model.param(n).detach().cpu().clone()
        # get_scaling of the prefilter, evaluated where the step evaluates it (torch.exp on the device)
        self.exp_scales = torch.exp(model.param("scaling")[:, :3]).cpu().numpy()
        if self.opt is not None:
            with torch.no_grad():
                for n, q in self.params.items():
                    q.copy_((self.p if n in self.p else self.mlp)[n])

    def sync_params_from_flat(self, model, flat: np.ndarray):
        """Parameters of an Adam-only instance <- a host copy of the model's bucket (the optimizer state stays this chain's)."""
        t = torch.from_numpy(flat)
        for n in self.p:
            self.p[n] = model._view(t, n).clone()
        for n in self.mlp:
            self.mlp[n] = model._view(t, n).clone()
        if self.opt is not None:
            with torch.no_grad():
                for n, q in self.params.items():
                    q.copy_((self.p if n in self.p else self.mlp)[n].reshape(q.shape))

    def visible(self):
        cam = self.cam
        return gs_oracle.visible_filter(self.p["anchor"].numpy(), self.exp_scales, self.rotation.numpy(), 1.0,
                                        cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.height, cam.width)

    def forward(self, gt, pose7):
        """-> (loss value, image (3,H,W) f32, unstable-pixel mask); keeps what backward() needs."""
        from segs_slam_amd import loss_utils
        cam = self.cam
        d64 = lambda t: t.double().requires_grad_(True)  # noqa: E731
        self.leaf = {n: d64(t) for n, t in self.p.items()}
        self.leaf_mlp = {n: d64(t) for n, t in self.mlp.items()}
        self.radii = self.visible()
        vis = torch.from_numpy(self.radii > 0)
        xyz, color, opacity, scaling, rot, neural_opacity, mask = neural_ref.generate_neural_gaussians(
            self.rd, self.leaf["anchor"], self.leaf["offset"], self.leaf["anchor_feat"], self.leaf["scaling"], self.leaf_mlp,
            torch.from_numpy(cam.camera_center).double(), torch.tensor(pose7, dtype=torch.float64), vis)
        self.outs = (xyz, color, opacity, scaling, rot)
        f32 = lambda t: t.detach().float().numpy()  # noqa: E731
        self.o = gs_oracle.Oracle()
        self.o.forward(np.zeros(3, dtype=np.float32), f32(xyz), f32(color), f32(opacity), f32(scaling), 1.0, f32(rot),
                       cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.height, cam.width)
        image = torch.from_numpy(self.o.get("out_color"))
        self.unstable = _dilate(torch.from_numpy(self.o.unstable_pixels(1e-5)))
        img = image.double().requires_grad_(True)
        g64 = gt.cpu().double()
        self.reg = scaling.prod(1).mean() if self.reg_w else torch.zeros((), dtype=torch.float64)
        base = (1.0 - self.lam) * loss_utils.l1_loss(img, g64) + self.lam * (1.0 - loss_utils.ssim(img, g64))
        (self.dL,) = torch.autograd.grad(base, img)
        return float(base) + self.reg_w * float(self.reg), image, self.unstable

    def backward(self):
        """-> dict of gradients (float64 numpy) for the four anchor tensors and every MLP tensor."""
        dL = self.dL.clone()
        dL[:, self.unstable] = 0
        g = self.o.backward(dL.float().numpy())
        xyz, color, opacity, scaling, rot = self.outs
        t64 = lambda a: torch.from_numpy(a).double()  # noqa: E731
        heads = [xyz, color, opacity, scaling, rot]
        gin = [t64(g["dL_dmean3D"]), t64(g["dL_dcolor"]), t64(g["dL_dopacity"]), t64(g["dL_dscale"]), t64(g["dL_drot"])]
        if self.reg_w:
            heads.append(self.reg)
            gin.append(torch.tensor(self.reg_w, dtype=torch.float64))
        torch.autograd.backward(heads, gin)
        out = {n: (t.grad if t.grad is not None else torch.zeros_like(t)).numpy() for n, t in self.leaf.items()}
        out.update({n: (t.grad if t.grad is not None else torch.zeros_like(t)).numpy() for n, t in self.leaf_mlp.items()})
        return out

    def adam(self, grads, lrs):
        """torch.optim.Adam in float32 over the tensors, lr per reference group (src/gaussian_model.cpp:632-690)."""
        lr_of = lambda n: {"mlp_opacity": lrs["mlp_opacity"], "mlp_cov": lrs["mlp_cov"], "mlp_color": lrs["mlp_color"],  # noqa: E731
                           "mlp_apperance": lrs["appearance"], "mlp_feature_bank": lrs["mlp_featurebank"]}[n.split(".")[0]]
        if self.opt is None:
            self.params = {n: torch.nn.Parameter(t.clone()) for n, t in {**self.p, **self.mlp}.items()}
            groups = [{"params": [self.params[n]], "lr": lrs[k]} for n, k in self.GROUPS]
            groups += [{"params": [self.params[n]], "lr": lr_of(n)} for n in self.mlp]
            self.opt = torch.optim.Adam(groups, eps=1e-15)
        else:
            for grp, (n, k) in zip(self.opt.param_groups, self.GROUPS):
                grp["lr"] = lrs[k]
            for grp, n in zip(self.opt.param_groups[4:], self.mlp):
                grp["lr"] = lr_of(n)
        for n, q in self.params.items():
            q.grad = torch.from_numpy(np.asarray(grads[n])).float().reshape(q.shape)
        self.opt.step()
        for n in self.p:
            self.p[n] = self.params[n].detach().clone()
        for n in self.mlp:
            self.mlp[n] = self.params[n].detach().clone()


DIMS = {
    "replica": dict(feat_dim=32, n_offsets=10, appearance_dim=32, use_feat_bank=True, add_opacity_dist=False, add_cov_dist=False, add_color_dist=False),
    # cfg/gaussian_mapper/RGB-D/ScanNet/scannet_rgbd.yaml:20-21 (BASELINE config 5)
    "scannet": dict(feat_dim=32, n_offsets=10, appearance_dim=16, use_feat_bank=False, add_opacity_dist=False, add_cov_dist=False, add_color_dist=False),
}


@pytest.mark.parametrize("cfg", ["replica", "scannet"])
def test_scaffold_training_once_matches_reference_chain(cfg):
    """Two whole mapper iterations (loss with the 0.01 scaling regulariser): loss, exchanged-bucket gradient and the
    parameters after Adam against the float64 / oracle chain."""
    from segs_slam_amd import neural_gaussians as ng, scenes
    dev = torch.device(DEV)
    W, H = 320, 240
    cam = scenes.make_camera(W, H, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    dims = ng.ModelDims(**DIMS[cfg])
    A = 3000
    pose7 = (0.1, -0.05, 0.02, 0.98, 0.05, -0.1, 0.15)
    kf = _kf(cam, dev, pose7)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(11)).to(dev)

    model = ng.synthetic_model(A, dims, cam, dev, seed=21)
    step = ng.ScaffoldTrainerStep(model, W, H, scaling_reg_weight=0.01)
    ref = ReferenceScaffoldStep(model, DIMS[cfg], cam, step.opt.lambda_dssim, 0.01)
    # a second torch.optim.Adam that is fed the DEVICE's own bucket gradient: the optimizer wiring (groups, learning rates,
    # step counts, bias corrections, eps) checked on every live parameter, independent of gradient noise (see below)
    ref_same_grad = ReferenceScaffoldStep(model, DIMS[cfg], cam, step.opt.lambda_dssim, 0.01)
    p_init = model.params.cpu().numpy().copy()

    # the step's loss object, with dL/dimage zeroed on the reference's unstable pixels (set before each iteration)
    mask_dev = torch.ones(H, W, device=dev)
    fused = step.loss_fn
    step.loss_fn = lambda img, g: (lambda l, d: (l, d * mask_dev))(*fused(img, g))

    captured = {}
    orig_adam = step._adam

    def spy(groups, count, guard):
        if "g" not in captured:
            torch.cuda.synchronize()
            captured["g"] = model.grads.clone()      # the bucket exactly as the optimizer receives it
        return orig_adam(groups, count, guard)
    step._adam = spy

    solid = zero_both = None
    for it in range(2):
        ref.sync_params(model)
        p_before = model.params.cpu().numpy().copy()
        loss_ref, image_ref, unstable = ref.forward(gt, pose7)
        assert unstable.float().mean() < 0.1
        mask_dev.copy_((~unstable).float().to(dev))
        grads_ref = ref.backward()
        captured.clear()
        loss = step.training_once([kf], [gt])
        torch.cuda.synchronize()
        # prefilter: identical anchor visibility
        assert np.array_equal(step.visible_radii[:A].cpu().numpy(), ref.radii), "prefilter_voxel radii differ"
        total = float(loss) + float(step.neural.scaling_reg)
        assert abs(total - loss_ref) <= 1e-5 * abs(loss_ref), (it, total, loss_ref)
        ok = ~unstable.numpy()
        img = step.engine.out_color.cpu().numpy()
        assert np.all(np.abs(img - image_ref.numpy())[:, ok] <= 1e-4 * np.abs(image_ref.numpy())[:, ok] + 2e-5), it
        g = captured["g"]
        this_solid = np.zeros(model.params.numel(), dtype=bool)
        for name in ("anchor", "offset", "anchor_feat", "scaling"):
            got = model._view(g, name).cpu().numpy()
            _grad_check(f"{name}@{it}", got, grads_ref[name].reshape(got.shape))
        for name in model.mlp_layout:
            got = model._view(g, name).cpu().numpy()
            _grad_check(f"{name}@{it}", got, grads_ref[name].reshape(got.shape))
        gflat = g.cpu().numpy()
        this_solid = np.abs(gflat) > 1e-6 * np.abs(gflat).max()
        solid = this_solid if solid is None else (solid & this_solid)
        # the reference's gradient in the bucket's layout: entries it leaves at exactly zero (anchors outside the frustum,
        # offsets whose Gaussian was masked out) must not move on the device either
        gref_flat = torch.zeros(model.params.numel())
        for name in list(("anchor", "offset", "anchor_feat", "scaling")) + list(model.mlp_layout):
            v = model._view(gref_flat, name)
            v.copy_(torch.as_tensor(np.asarray(grads_ref[name], dtype=np.float32)).reshape(v.shape))
        this_zero = (gref_flat.numpy() == 0) & (gflat == 0)
        zero_both = this_zero if zero_both is None else (zero_both & this_zero)
        ref.adam(grads_ref, step.learning_rates(step.iteration))
        ref_same_grad.sync_params_from_flat(model, p_before)
        ref_same_grad.adam({n: model._view(g, n).cpu().numpy() for n in list(ref.p) + list(ref.mlp)}, step.learning_rates(step.iteration))
    assert step._mlp_count.value() == 2
    # the second step's update (first and second moments carry the first step's gradient)
    p_init = p_before
    p_gpu = model.params.cpu().numpy()
    p_ref = np.zeros_like(p_gpu)
    ref_model_view = torch.from_numpy(p_ref)
    for name in ("anchor", "offset", "anchor_feat", "scaling"):
        model._view(ref_model_view, name).copy_(ref.p[name].reshape(model._view(ref_model_view, name).shape))
    for name in model.mlp_layout:
        model._view(ref_model_view, name).copy_(ref.mlp[name])
    # Every live entry of the bucket falls in one of three classes, and the first two must cover nearly all of them:
    #   zero  -- gradient exactly 0 in both steps on both sides (moments stay 0): the parameter must not move at all;
    #   solid -- gradient above 1e-6 of the bucket's largest in both steps: the update agrees to 1e-3 relative;
    #   noise -- a non-zero gradient that is itself rounding noise: Adam with eps 1e-15 turns it into a full-size step of either
    #            sign, so only its size is bounded (|update| <= 2 steps of the largest learning rate).
    live = np.zeros(model.params.numel(), dtype=bool)
    for name in ("anchor", "offset", "anchor_feat", "scaling"):
        o, n = model.segments[name]
        live[o:o + n] = True
    live[model.mlp_offset:] = True
    zero_both &= live
    solid &= live
    assert np.array_equal(p_gpu[zero_both], p_init[zero_both]) and np.array_equal(p_ref[zero_both], p_init[zero_both])
    upd, upd_ref = (p_gpu - p_init)[solid], (p_ref - p_init)[solid]
    bad = np.abs(upd - upd_ref) > 1e-3 * np.abs(upd_ref) + 2e-7
    assert bad.mean() < 1e-4, (float(bad.mean()), float(np.abs(upd - upd_ref).max()), float(np.abs(upd_ref).max()))
    noise = live & ~zero_both & ~solid
    lr_max = max(step.learning_rates(step.iteration).values())
    assert np.all(np.abs(p_gpu - p_init)[noise] <= 2.0 * lr_max * 1.001), float(np.abs(p_gpu - p_init)[noise].max())
    # How much of the bucket the oracle-gradient comparison above can speak for is a property of the scene, not of the code:
    # 58 % solid, 24 % zero, 18 % noise here (offsets of Gaussians a few pixels see).  Round 3 asserted "covered > 0.90", saw
    # 0.823 on the GPU box and lowered the bound to 0.80 / 0.50 -- a threshold fitted to its own measurement.  It is printed
    # now, not asserted; what IS asserted for every live parameter is the optimizer itself: torch.optim.Adam fed the bucket
    # gradient the device's Adam received must land on the device's parameters (1e-3 of the update + 1e-9: two float32
    # evaluations of LibTorch's formula), for both steps' moments, all groups, all learning rates.
    print(f"oracle-gradient Adam comparison covers {float((zero_both | solid)[live].mean()):.3f} of the live parameters "
          f"(solid {float(solid[live].mean()):.3f}, zero {float(zero_both[live].mean()):.3f})")
    p_same = np.zeros_like(p_gpu)
    same_view = torch.from_numpy(p_same)
    for name in ("anchor", "offset", "anchor_feat", "scaling"):
        model._view(same_view, name).copy_(ref_same_grad.p[name].reshape(model._view(same_view, name).shape))
    for name in model.mlp_layout:
        model._view(same_view, name).copy_(ref_same_grad.mlp[name])
    upd_dev, upd_same = (p_gpu - p_init)[live], (p_same - p_init)[live]
    off = np.abs(upd_dev - upd_same) > 1e-3 * np.abs(upd_same) + 1.2e-7 * np.abs(p_init[live]) + 1e-9    # (+ one float32 ulp of the parameter itself)
    assert not off.any(), (int(off.sum()), int(live.sum()), float(np.abs(upd_dev - upd_same).max()))
    assert float((upd_same != 0).mean()) > 0.7      # and it is not a comparison of zeros


def test_config4_trainer_step_gradients_match_oracle():
    """BASELINE config 4's single-rank unit of work: TrainerStep on the `c4` scene (TUM fr3 camera, 640x480, 200 k Gaussians):
    loss and the five parameter gradients of one render + L1/SSIM + backward against the oracle; the Adam update against
    torch.optim.Adam fed the same gradient.  (The 8-rank exchange around it: tests/test_distributed_cpu.py,
    tests/test_bench_contract.py.)"""
    from segs_slam_amd import loss_utils, scenes
    from segs_slam_amd.gaussian_trainer import TrainerStep, field_segments, keyframe_tensors
    from segs_slam_amd.raster_engine import FIELDS
    dev = torch.device(DEV)
    sc = scenes.make_config_scene("c4")
    cam = sc.camera
    gt = torch.rand(3, cam.height, cam.width, generator=torch.Generator().manual_seed(4)).to(dev)
    step = TrainerStep.on_gpu(sc, dev)
    p_init = step.params_flat.cpu().numpy().copy()

    o = gs_oracle.Oracle()
    o.forward(sc.bg, sc.means3D, sc.colors, sc.opacity, sc.scales, 1.0, sc.rotations, cam.world_view_transform,
              cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.height, cam.width)
    unstable = _dilate(torch.from_numpy(o.unstable_pixels(1e-5)))
    assert unstable.float().mean() < 0.1
    img = torch.from_numpy(o.get("out_color")).double().requires_grad_(True)
    g64 = gt.cpu().double()
    lam = step.opt.lambda_dssim
    loss_ref = (1.0 - lam) * loss_utils.l1_loss(img, g64) + lam * (1.0 - loss_utils.ssim(img, g64))
    (dL,) = torch.autograd.grad(loss_ref, img)
    dL[:, unstable] = 0
    gref = o.backward(dL.float().numpy())

    mask_dev = (~unstable).float().to(dev)
    fused = step.fused_loss
    step.fused_loss = lambda im, g: (lambda l, d: (l, d * mask_dev))(*fused(im, g))
    captured = {}
    orig = step.optimizer.step

    def spy(p, g, lrs, P, scale, exchange=None, guard=None):
        torch.cuda.synchronize()
        captured["g"] = g.clone()
        return orig(p, g, lrs, P, scale, exchange=exchange, guard=guard)
    step.optimizer.step = spy
    loss = step.training_once([keyframe_tensors(cam, dev)], [gt])
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * float(loss_ref), (float(loss), float(loss_ref))
    from segs_slam_amd.raster_engine import split_flat
    got = split_flat(captured["g"], sc.P)
    for name, key in (("means3D", "dL_dmean3D"), ("scales", "dL_dscale"), ("rotations", "dL_drot"), ("opacity", "dL_dopacity"),
                      ("colors", "dL_dcolor")):
        _grad_check(name, got[name].cpu().numpy(), gref[key])
    # the update: torch.optim.Adam on the GPU's own gradient (the fused kernel's arithmetic is pinned bit-exactly elsewhere)
    lrs = step.learning_rates(1)
    params, off = [], 0
    for (name, n) in FIELDS:
        params.append(torch.nn.Parameter(torch.from_numpy(p_init[off:off + sc.P * n].copy())))
        off += sc.P * n
    topt = torch.optim.Adam([{"params": [q], "lr": lrs[name]} for q, (name, _) in zip(params, FIELDS)], eps=1e-15)
    gcpu, off = captured["g"].cpu(), 0
    for q, (o_, n_, _) in zip(params, field_segments(lrs, sc.P)):
        q.grad = gcpu[o_:o_ + n_].clone()
    topt.step()
    want = torch.cat([q.detach() for q in params]).numpy()
    assert np.allclose(step.params_flat.cpu().numpy(), want, rtol=2e-6, atol=5e-7)


def test_config5_scaffold_step_at_full_size():
    """BASELINE config 5 itself: 300 k anchors (3 M candidate Gaussians) with the ScanNet model dimensions (appearance_dim 16,
    no feature bank) at 1200x680.  No CPU reference can carry this size in test time, so the checks are size-independent
    properties: everything finite; the resident (no host sync, tight binning) image equals the reference-shaped synchronising
    path's bit for bit on the same neural Gaussians; the whole backward (raster + neural, into the flat bucket) is linear in
    dL/dimage; one training_once moves the parameters and leaves the gradient bucket clean.  Oracle parity of this exact model
    configuration at a size the float64 chain can carry: test_scaffold_training_once_matches_reference_chain[scannet]."""
    from segs_slam_amd import neural_gaussians as ng, scenes
    from segs_slam_amd.raster_engine import RasterEngine
    dev = torch.device(DEV)
    W, H = 1200, 680
    cam = scenes.make_camera(W, H, 600.0, 600.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    dims = ng.ModelDims(**DIMS["scannet"])
    A = 300_000
    model = ng.synthetic_model(A, dims, cam, dev, seed=5)
    step = ng.ScaffoldTrainerStep(model, W, H, scaling_reg_weight=0.01)
    kf = _kf(cam, dev, (0.2, 0.0, -0.1, 1.0, 0.0, 0.0, 0.0))
    gen = torch.Generator().manual_seed(55)
    gt = torch.rand(3, H, W, generator=gen).to(dev)

    for _ in range(2):                      # calibrating pass, then a resident one
        image = step.render(kf).clone()
    assert step.engine.check() and step.engine._last_resident
    ngs = step.neural
    assert torch.isfinite(image).all() and float(image.abs().max()) > 0
    sync = RasterEngine(ngs.P, W, H, dev, resident=False, skip_nonpositive_opacity=True)
    want = sync.forward(step.bg, ngs.means3D[:ngs.P], ngs.colors[:ngs.P], ngs.opacity[:ngs.P], ngs.scales[:ngs.P], ngs.rotations[:ngs.P],
                        kf.view, kf.proj, kf.campos, kf.tanfovx, kf.tanfovy)
    assert torch.equal(image, want), "resident image differs from the reference-shaped path"
    assert 0 < step.engine.R <= sync.R and torch.equal(step.engine.radii[:ngs.P], sync.radii)
    del sync

    def bucket_gradient(dL):
        model.grads.zero_()
        g = step.engine.backward(dL.contiguous())
        ngs.backward(g["means3D"], g["colors"], g["opacity"], g["scales"], g["rotations"], 0.0)
        torch.cuda.synchronize()
        return model.grads.clone()
    d1 = (torch.rand(3, H, W, generator=gen).to(dev) * 2 - 1) / (3 * H * W)
    d2 = (torch.rand(3, H, W, generator=gen).to(dev) * 2 - 1) / (3 * H * W)
    g1, g2, g12 = bucket_gradient(d1), bucket_gradient(d2), bucket_gradient(d1 + 0.5 * d2)
    assert torch.isfinite(g12).all() and float(g12.abs().max()) > 0
    lin = g1 + 0.5 * g2
    scale = float(lin.abs().max())
    err = (g12 - lin).abs()
    assert float(err.max()) <= 2e-4 * scale, (float(err.max()), scale)
    assert float((err > 1e-4 * lin.abs() + 1e-6 * scale).float().mean()) < 1e-4
    model.grads.zero_()

    before = model.params.clone()
    loss = step.training_once([kf], [gt])
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and torch.isfinite(model.params).all()
    assert step._mlp_count.value() == 1 and float(model.grads.abs().max()) == 0.0
    moved = (model.params != before).float().mean()
    assert float(moved) > 0.05, float(moved)
