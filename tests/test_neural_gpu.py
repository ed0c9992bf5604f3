"""GPU parity of the fused neural-Gaussian generation (include/segs_neural.h) against the torch restatement of
src/gaussian_renderer.cpp:214-334 (oracle/neural_ref.py), forward and backward.

Tolerances: the reference computes these MLPs with cuBLAS GEMMs whose summation order is unspecified, so parity is
tolerance-based: forward 2e-5 absolute on O(1) outputs, gradients 1e-4 relative to the largest entry of each tensor
(the bar north_star states for gradients)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import neural_ref  # noqa: E402


def _setup(dims_kw, A, seed, device):
    from segs_slam_amd import neural_gaussians as ng
    rd = neural_ref.NeuralDims(**dims_kw)
    md = ng.ModelDims(**dims_kw)
    anchor, offset, feat, scaling_log, mlp = neural_ref.random_model(rd, A, seed)
    model = ng.ScaffoldModel(A, md, device)
    model.load(anchor, offset, feat, scaling_log, mlp)
    return rd, model, (anchor, offset, feat, scaling_log, mlp)


CASES = [
    dict(feat_dim=32, n_offsets=10, appearance_dim=32, use_feat_bank=True, add_opacity_dist=False, add_cov_dist=False, add_color_dist=False),
    dict(feat_dim=32, n_offsets=10, appearance_dim=16, use_feat_bank=False, add_opacity_dist=False, add_cov_dist=False, add_color_dist=False),
    dict(feat_dim=32, n_offsets=10, appearance_dim=0, use_feat_bank=False, add_opacity_dist=True, add_cov_dist=True, add_color_dist=True),
    dict(feat_dim=32, n_offsets=10, appearance_dim=8, use_feat_bank=True, add_opacity_dist=True, add_cov_dist=False, add_color_dist=True),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("A", [1, 700])
def test_forward_and_backward_match_restatement(case, A):
    from segs_slam_amd import neural_gaussians as ng
    dev = torch.device("cuda:0")
    rd, model, (anchor, offset, feat, scaling_log, mlp) = _setup(CASES[case], A, 300 + case, dev)
    g = torch.Generator().manual_seed(7 + case)
    campos = torch.tensor([0.1, -0.2, -0.5])
    pose7 = torch.tensor([0.3, -0.1, 0.2, 0.9, 0.1, -0.3, 0.2])
    visible = torch.rand(A, generator=g) < 0.7
    if A == 1:
        visible[:] = True
    radii = torch.where(visible, torch.tensor(3), torch.tensor(0)).to(torch.int32)

    gen = ng.NeuralGaussians(model)
    gen.forward(campos.to(dev), pose7.to(dev), radii.to(dev))
    torch.cuda.synchronize()

    # ---- reference (float64 autograd)
    d64 = lambda t: t.double().requires_grad_(True)  # noqa: E731
    r_anchor, r_offset, r_feat, r_scal = d64(anchor), d64(offset), d64(feat), d64(scaling_log)
    r_mlp = {k: d64(v) for k, v in mlp.items()}
    xyz, color, opacity, scaling, rot, neural_opacity, mask = neural_ref.generate_neural_gaussians(
        rd, r_anchor, r_offset, r_feat, r_scal, r_mlp, campos.double(), pose7.double(), visible)

    # candidate-domain rows of the visible anchors, then the reference's mask
    vis_rows = visible.repeat_interleave(10)
    nop = gen.neural_opacity.cpu().view(-1)
    assert torch.all(nop[~vis_rows] == 0)
    np.testing.assert_allclose(nop[vis_rows].numpy(), neural_opacity.detach().view(-1).numpy(), atol=2e-5)
    dmask = gen.mask().cpu()
    # a sign flip of an opacity within the forward tolerance of zero is not an error of either side
    ref_full_mask = torch.zeros(A * 10, dtype=torch.bool)
    ref_full_mask[vis_rows] = mask
    near_zero = torch.zeros(A * 10, dtype=torch.bool)
    near_zero[vis_rows] = neural_opacity.detach().view(-1).abs() < 2e-5
    assert torch.all((dmask == ref_full_mask) | near_zero)
    if near_zero.any():
        pytest.skip("an opacity within tolerance of 0 makes the masks incomparable for this seed")
    for name, ours, ref in (("xyz", gen.means3D, xyz), ("color", gen.colors, color), ("opacity", gen.opacity, opacity),
                            ("scaling", gen.scales, scaling), ("rot", gen.rotations, rot)):
        np.testing.assert_allclose(ours.cpu()[dmask].numpy(), ref.detach().numpy(), atol=2e-5, rtol=2e-5, err_msg=name)

    # ---- backward: random candidate-domain gradients; the reference sees them through its compaction
    P = A * 10
    gm, gc, go, gs, gr = (torch.randn(P, n, generator=g) for n in (3, 3, 1, 3, 4))
    reg_w = 0.01 if case % 2 == 0 else 0.0       # the mapper's scaling regulariser (src/gaussian_mapper.cpp:926-928)
    reg = reg_w * scaling.prod(1).mean()
    loss = ((xyz * gm[dmask].double()).sum() + (color * gc[dmask].double()).sum() + (opacity * go[dmask].double()).sum()
            + (scaling * gs[dmask].double()).sum() + (rot * gr[dmask].double()).sum()) + 1e4 * reg
    loss.backward()
    model.grads.zero_()
    gen.backward(gm.to(dev), gc.to(dev), go.to(dev), gs.to(dev), gr.to(dev), scaling_reg_weight=1e4 * reg_w)
    torch.cuda.synchronize()
    if reg_w:
        assert abs(gen.scaling_reg.item() - 1e4 * reg.item()) <= 1e-5 * 1e4 * reg.item()

    def close(name, ours, ref):
        ref = ref if ref is not None else torch.zeros_like(ours, dtype=torch.float64)
        scale = max(float(ref.abs().max()), 1e-12)
        err = float((ours.cpu().double() - ref).abs().max()) / scale
        assert err < 1e-4, f"{name}: max err / max|ref| = {err:.3e}"

    close("anchor", model.grad("anchor"), r_anchor.grad)
    close("offset", model.grad("offset"), r_offset.grad)
    close("anchor_feat", model.grad("anchor_feat"), r_feat.grad)
    close("scaling", model.grad("scaling"), r_scal.grad)
    for n in model.dims.mlp_tensor_names():
        close(n, model.grad(n), r_mlp[n].grad)

    # gradients accumulate: a second backward doubles them
    gen.backward(gm.to(dev), gc.to(dev), go.to(dev), gs.to(dev), gr.to(dev), scaling_reg_weight=1e4 * reg_w)
    torch.cuda.synchronize()
    close("anchor_feat x2", model.grad("anchor_feat"), 2 * r_feat.grad)
    close("mlp_cov.2.weight x2", model.grad("mlp_cov.2.weight"), 2 * r_mlp["mlp_cov.2.weight"].grad)


def test_no_visible_anchor_is_a_no_op():
    from segs_slam_amd import neural_gaussians as ng
    dev = torch.device("cuda:0")
    rd, model, _ = _setup(CASES[0], 300, 5, dev)
    gen = ng.NeuralGaussians(model)
    radii = torch.zeros(300, dtype=torch.int32, device=dev)
    gen.opacity.fill_(5.0)
    gen.forward(torch.zeros(3, device=dev), torch.zeros(7, device=dev), radii)
    assert float(gen.opacity.abs().max()) == 0.0
    z = torch.zeros(3000, 4, device=dev)
    gen.backward(z[:, :3].contiguous(), z[:, :3].contiguous(), z[:, :1].contiguous(), z[:, :3].contiguous(), z)
    torch.cuda.synchronize()
    assert float(model.grads.abs().max()) == 0.0


def test_unsupported_dims_are_rejected():
    from segs_slam_amd import _capi, neural_gaussians as ng
    with pytest.raises(_capi.SegsError):
        ng.ScaffoldModel(10, ng.ModelDims(feat_dim=64), "cuda:0")


def test_scaffold_render_equals_compacted_render():
    """Rendering the candidate-domain arrays with SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY equals rendering the
    reference's compacted Gaussians (array[mask]) through the plain engine: same image, same gradients."""
    from segs_slam_amd import neural_gaussians as ng, scenes
    from segs_slam_amd.raster_engine import RasterEngine
    dev = torch.device("cuda:0")
    A = 4000
    rd, model, (anchor, *_rest) = _setup(CASES[0], A, 11, dev)
    cam = scenes.make_camera(320, 240, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.1, 0.2, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    step = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
    img = step.render(kf).clone()
    dL = torch.randn(3, cam.height, cam.width, device=dev) / (3 * cam.height * cam.width)
    g = {k: v.clone() for k, v in step.engine.backward(dL).items()}
    mask = step.neural.mask()
    assert int(mask.sum()) > 1000 and int((step.engine.radii > 0).sum()) > 100
    n = step.neural
    c = lambda x: x[mask].contiguous()  # noqa: E731
    Pc = int(mask.sum())
    eng = RasterEngine(Pc, cam.width, cam.height, dev)
    img2 = eng.forward(step.bg, c(n.means3D), c(n.colors), c(n.opacity), c(n.scales), c(n.rotations), kf.view, kf.proj,
                       kf.campos, kf.tanfovx, kf.tanfovy)
    assert torch.equal(img, img2)
    g2 = eng.backward(dL)
    for k in g:
        ref = g2[k]
        scale = max(float(ref.abs().max()), 1e-20)
        assert float((g[k][mask] - ref).abs().max()) / scale < 1e-4, k
        assert float(g[k][~mask].abs().max()) == 0.0, k


def test_scaffold_trainer_with_mapper_loss_terms_runs():
    """Scaling regulariser + FFT high-frequency regulariser (the mapper's loss, src/gaussian_mapper.cpp:924-945) wired in."""
    from segs_slam_amd import neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    rd, model, _ = _setup(CASES[0], 3000, 23, dev)
    cam = scenes.make_camera(320, 240, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.1, 0.2, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    step = ng.ScaffoldTrainerStep(model, cam.width, cam.height, scaling_reg_weight=0.01)
    step.enable_frequency_regularization(start=0)
    gt = torch.rand((3, cam.height, cam.width), device=dev)
    losses = [float(step.training_once([kf], [gt])) for _ in range(5)]
    assert np.isfinite(losses).all() and float(step.neural.scaling_reg) > 0


def test_scaffold_trainer_reduces_loss():
    from segs_slam_amd import neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    A = 6000
    rd, model, _ = _setup(CASES[0], A, 21, dev)
    cam = scenes.make_camera(320, 240, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.1, 0.2, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    step = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
    gt = torch.full((3, cam.height, cam.width), 0.4, device=dev)
    losses = [float(step.training_once([kf], [gt])) for _ in range(40)]
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.8 * losses[0], losses[::8]


def test_model_io_round_trip(tmp_path):
    """save_ply writes the reference's property list (src/gaussian_model.cpp:1179-1261) and reads back bit-exactly;
    the MLP text files have the reference's names and `%.5f` rows."""
    from segs_slam_amd import model_io, neural_gaussians as ng
    dev = torch.device("cuda:0")
    _, model, (anchor, offset, feat, scaling_log, mlp) = _setup(CASES[0], 37, 9, dev)
    p = str(tmp_path / "point_cloud.ply")
    model_io.save_ply(model, p)
    head = open(p, "rb").read(4096).split(b"end_header\n")[0].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 37\nproperty float x\n")
    back = model_io.load_ply(p)
    assert back["names"] == model_io.ply_property_names(32, 10)
    assert back["names"][6] == "anchor_feat_0" and back["names"][38] == "offset_0" and back["names"][68] == "opacity"
    np.testing.assert_array_equal(back["anchor"], anchor.numpy())
    np.testing.assert_array_equal(back["offset"], offset.numpy())
    np.testing.assert_array_equal(back["anchor_feat"], feat.numpy())
    np.testing.assert_array_equal(back["scaling"], scaling_log.numpy())
    np.testing.assert_array_equal(back["rotation"][:, 0], np.ones(37, np.float32))
    model_io.save_mlp_checkpoints(model, str(tmp_path / "mlp"))
    rows = open(tmp_path / "mlp" / "cov_weight2.txt").read().strip().split("\n")
    assert len(rows) == 70 and len(rows[0].split(" ")) == 32
    assert rows[3].split(" ")[5] == f"{float(mlp['mlp_cov.2.weight'][3, 5]):.5f}"
    for fn in ("opacity_weight1.txt", "opacity_bias2.txt", "color_bias1.txt", "feat_weight2.txt"):
        assert (tmp_path / "mlp" / fn).exists()
    # and back: the anchor tensors bit-exactly, the MLPs to the 5 decimals the text format keeps
    again = model_io.load_model(p, str(tmp_path / "mlp"), model.dims, dev)
    assert again.A == 37
    for name in ("anchor", "offset", "anchor_feat", "scaling"):
        assert torch.equal(again.param(name), model.param(name)), name
    assert torch.equal(again.rotation[:37], model.rotation[:37]) and torch.equal(again.opacity[:37], model.opacity[:37])
    for name in model.dims.mlp_tensor_names():
        assert float((again.param(name) - model.param(name)).abs().max()) <= 5.1e-6, name
    with pytest.raises(ValueError):
        model_io.load_model(p, str(tmp_path / "mlp"), ng.ModelDims(appearance_dim=16), dev)


@pytest.mark.parametrize("case", [0, 1])   # 0: feature bank (one-kernel backward), 1: plain model (chain + weight-gradient wave pairs)
def test_second_slab_of_persistent_workgroups_is_consistent(case):
    """A = 140 001 anchors exceed the forward's 512 workgroups x 128 anchors (and the backward's 256), so the persistent
    workgroups loop over further slabs.  Checked
    by consistency (no float64 reference at this size): the same anchors processed as two smaller models give bit-identical
    per-anchor outputs and gradients (the arithmetic of an anchor does not depend on the wave that carries it) and MLP weight
    gradients that add up (different summation order: 1e-5 of the largest entry)."""
    from segs_slam_amd import neural_gaussians as ng
    dev = torch.device("cuda:0")
    A, A1 = 140001, 70000
    rd = neural_ref.NeuralDims(**CASES[case])
    md = ng.ModelDims(**CASES[case])
    anchor, offset, feat, scaling_log, mlp = neural_ref.random_model(rd, A, 77)
    campos = torch.tensor([0.1, -0.2, -0.5], device=dev)
    pose7 = torch.tensor([0.3, -0.1, 0.2, 0.9, 0.1, -0.3, 0.2], device=dev)
    g = torch.Generator().manual_seed(3)
    grads = [torch.randn(A * 10, n, generator=g).to(dev) for n in (3, 3, 1, 3, 4)]

    def run(lo, hi):
        m = ng.ScaffoldModel(hi - lo, md, dev)
        m.load(anchor[lo:hi], offset[lo:hi], feat[lo:hi], scaling_log[lo:hi], mlp)
        gen = ng.NeuralGaussians(m)
        gen.forward(campos, pose7, None)
        outs = [t.clone() for t in (gen.means3D, gen.colors, gen.opacity, gen.scales, gen.rotations)]
        gen.backward(*[x[lo * 10:hi * 10].contiguous() for x in grads])
        torch.cuda.synchronize()
        return m, outs

    big, ob = run(0, A)
    s1, o1 = run(0, A1)
    s2, o2 = run(A1, A)
    for tb, t1, t2 in zip(ob, o1, o2):
        assert torch.equal(tb, torch.cat([t1, t2], dim=0))
    for name in ("anchor", "offset", "anchor_feat", "scaling"):
        assert torch.equal(big.grad(name), torch.cat([s1.grad(name), s2.grad(name)], dim=0)), name
    ref = s1.mlp_grads + s2.mlp_grads
    assert float((big.mlp_grads - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("case", [1, 0])        # plain model (ScanNet dimensions), feature-bank model (Replica dimensions)
def test_wave_pair_backward_equals_the_one_kernel_backward(case):
    """The backward runs as pairs of chain / weight-gradient waves (neural_bwd_pair_kernel; since round 4 for the feature-bank
    model too); the one-kernel form it replaced stays reachable under SEGS_NEURAL_ONE_KERNEL_BACKWARD (include/segs_neural.h).
    The arithmetic of an anchor is the same in both for the plain model: per-anchor gradients must be bit-identical.  The
    feature-bank pairs form dL/dfeat through partial sums (register pressure): equal to rounding there."""
    from segs_slam_amd import _capi, neural_gaussians as ng
    dev = torch.device("cuda:0")
    kw, A = CASES[case], 40001
    rd, md = neural_ref.NeuralDims(**kw), ng.ModelDims(**kw)
    anchor, offset, feat, scaling_log, mlp = neural_ref.random_model(rd, A, 91)
    g = torch.Generator().manual_seed(5)
    radii = torch.where(torch.rand(A, generator=g) < 0.8, torch.tensor(3), torch.tensor(0)).to(torch.int32).to(dev)
    grads = [torch.randn(A * 10, n, generator=g).to(dev) for n in (3, 3, 1, 3, 4)]
    lib = _capi.lib()
    outs = []
    for flags in (0, 1):
        m = ng.ScaffoldModel(A, md, dev)
        m.load(anchor, offset, feat, scaling_log, mlp)
        gen = ng.NeuralGaussians(m)
        gen.forward(torch.tensor([0.1, -0.2, -0.5], device=dev), torch.tensor([0.3, -0.1, 0.2, 0.9, 0.1, -0.3, 0.2], device=dev), radii)
        old = lib.segs_neural_set_flags(flags)
        try:
            gen.backward(*grads, scaling_reg_weight=0.01)
        finally:
            assert lib.segs_neural_set_flags(old) == flags
        torch.cuda.synchronize()
        outs.append(m.grads.cpu().numpy().copy())
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    n_anchor = A * (3 + 30 + 32 + 6)   # the four per-anchor segments of the bucket; the MLP block follows
    if not kw["use_feat_bank"]:
        assert np.array_equal(outs[0][:n_anchor], outs[1][:n_anchor])
    else:
        a, b = outs[0][:n_anchor], outs[1][:n_anchor]
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max() and np.mean(a != b) < 0.5
    # the visible-anchor list is compacted with one atomic per 2048 anchors, so its block order -- and with it the order in which
    # the MLP weight gradients are summed -- differs from run to run: same bound as the split-model test above
    a, b = outs[0][n_anchor:], outs[1][n_anchor:]
    assert np.abs(a).max() > 0 and np.abs(a - b).max() <= 1e-5 * np.abs(a).max()


def test_scaffold_step_alternates_pyramid_levels_on_one_step_object():
    """Gaussian-pyramid training (src/gaussian_mapper.cpp:837-858: a keyframe is trained at the size of its current pyramid
    level): one ScaffoldTrainerStep takes targets of two sizes in turn.  Every iteration's render must equal, bit for bit, a
    reference-shaped render of the same neural Gaussians at that size; the optimizer takes every step; the loss of each level
    goes down; and the first low-resolution iteration equals the first iteration of a step built for that size alone."""
    from segs_slam_amd import neural_gaussians as ng, scenes
    from segs_slam_amd.raster_engine import RasterEngine
    dev = torch.device("cuda:0")
    sizes = [(320, 240), (160, 120)]
    cam = scenes.make_camera(320, 240, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    g = torch.Generator().manual_seed(5)
    full = torch.rand(3, 240, 320, generator=g)
    gts = {sizes[0]: full.to(dev), sizes[1]: torch.nn.functional.avg_pool2d(full[None], 2)[0].contiguous().to(dev)}

    model = ng.synthetic_model(3000, ng.ModelDims(), cam, dev, seed=9)
    step = ng.ScaffoldTrainerStep(model, *sizes[0])
    alone = ng.ScaffoldTrainerStep(ng.synthetic_model(3000, ng.ModelDims(), cam, dev, seed=9), *sizes[1])
    loss_alone = float(alone.training_once([kf], [gts[sizes[1]]]))
    losses = {s: [] for s in sizes}
    order = [sizes[1], sizes[0], sizes[1], sizes[0], sizes[1], sizes[0]]
    for it, sz in enumerate(order):
        loss = float(step.training_once([kf], [gts[sz]]))
        assert (step.W, step.H) == sz and step.engine.out_color.shape == (3, sz[1], sz[0])
        losses[sz].append(loss)
        if it == 0:
            assert loss == loss_alone
        # the image this iteration rendered (from the parameters BEFORE its update): re-render now-current parameters instead and
        # compare with the reference-shaped path at the same size
        img = step.render(kf).clone()              # (the projecting forward once the level's engine is calibrated)
        step.fuse_projection = False               # the reference-shaped render below reads the candidate colours / opacities
        assert torch.equal(step.render(kf), img)
        step.fuse_projection = True
        ngs = step.neural
        ref = RasterEngine(ngs.P_capacity, sz[0], sz[1], dev, resident=False, skip_nonpositive_opacity=True)
        ref.set_active(ngs.P)
        img_ref = ref.forward(step.bg, ngs.means3D, ngs.colors, ngs.opacity, ngs.scales, ngs.rotations, kf.view, kf.proj, kf.campos,
                              kf.tanfovx, kf.tanfovy)
        assert torch.equal(img, img_ref), (it, sz)
    torch.cuda.synchronize()
    assert step._mlp_count.value() == len(order) and step.dropped_steps() == 0
    for sz in sizes:
        assert losses[sz][-1] < losses[sz][0], (sz, losses[sz])
    assert len(step._levels) == 2


def _projecting_pair(case, A, seed, W, H, keep_dead, z_shift=0.0):
    """Two step objects over identical models: `a` renders with segs_neural_forward + the resident rasterizer (K1 included),
    `b` with segs_neural_forward_projected + the rasterizer without K1."""
    from segs_slam_amd import neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    steps = []
    for fuse in (False, True):
        rd, model, _ = _setup(CASES[case], A, seed, dev)
        if z_shift:
            model.param("anchor")[:, 2] += z_shift
        cam = scenes.make_camera(W, H, 0.9 * W, 0.9 * W, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
        step = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
        step.fuse_projection = fuse
        if keep_dead:
            step.engine.flags |= 2       # SEGS_RASTER_KEEP_DEAD_INSTANCES: the reference's rectangles, nothing dropped
        steps.append(step)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.1, 0.2, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    return steps[0], steps[1], kf, cam


@pytest.mark.parametrize("case,A,size,keep_dead", [(1, 4000, (320, 240), False), (0, 4001, (333, 187), False), (2, 37, (64, 72), False),
                                                  (3, 9000, (640, 480), True), (1, 70_000, (1200, 680), False)])
def test_projecting_forward_equals_forward_plus_k1(case, A, size, keep_dead):
    """SURVEY 8f n3 (src/gaussian_renderer.cpp:299-333 -> cuda_rasterizer/forward.cu:155-256): the neural forward that runs the
    rasterizer's per-Gaussian stage itself leaves, bit for bit, what the two separate kernels leave -- candidate geometry, radii,
    instance counts, image -- and the same gradients (up to the summation order of the tile backward's atomics)."""
    a, b, kf, cam = _projecting_pair(case, A, 40 + case, size[0], size[1], keep_dead)
    dL = torch.randn(3, cam.height, cam.width, device="cuda:0") / (3 * cam.height * cam.width)
    for it in range(3):                 # the first render calibrates (unfused on both sides), the others are resident
        ia, ib = a.render(kf), b.render(kf)
        torch.cuda.synchronize()
        assert a.engine.check() and b.engine.check()
        assert b.engine._last_resident == (it > 0)
        P = a.neural.P
        assert torch.equal(ia, ib), it
        assert torch.equal(a.engine.radii[:P], b.engine.radii[:P]), it
        assert torch.equal(a.visible_radii, b.visible_radii), it        # (b's were filled by the projecting forward itself once resident)
        assert (a.engine.R, a.engine.R_live) == (b.engine.R, b.engine.R_live), it
        for name in ("means3D", "scales", "rotations", "neural_opacity"):
            xa, xb = getattr(a.neural, name)[:P], getattr(b.neural, name)[:P]
            rows = (a.visible_radii[:a.model.A] > 0).repeat_interleave(10)       # rows of invisible anchors are not written
            assert torch.equal(xa[rows], xb[rows]), (it, name)
        ga = {k: v.clone() for k, v in a.engine.backward(dL).items()}
        gb = b.engine.backward(dL)
        for k in ga:
            scale = max(float(ga[k].abs().max()), 1e-20)
            assert float((ga[k][:P] - gb[k][:P]).abs().max()) <= 1e-5 * scale, (it, k)
        a.neural.backward(ga["means3D"], ga["colors"], ga["opacity"], ga["scales"], ga["rotations"], 0.0)
        b.neural.backward(gb["means3D"], gb["colors"], gb["opacity"], gb["scales"], gb["rotations"], 0.0)
        torch.cuda.synchronize()
        scale = float(a.model.grads.abs().max())
        assert scale > 0 and float((a.model.grads - b.model.grads).abs().max()) <= 1e-4 * scale
        a.model.grads.zero_(); b.model.grads.zero_()
    assert int((a.visible_radii[:a.model.A] > 0).sum()) > 0 and int((b.engine.radii[:P] > 0).sum()) > 0
    if A > 1000:
        assert int((a.visible_radii[:a.model.A] == 0).sum()) > 0        # some anchors are outside the frustum: their rows get radius 0
        inv = (b.visible_radii[:b.model.A] == 0).repeat_interleave(10)
        assert not bool(b.engine.radii[:P][inv].any())


def test_projecting_forward_flags_a_depth_beyond_the_sort_key_range():
    """A binned Gaussian deeper than the resident sort's 27-bit key range (13 107 m) must flag the step exactly as K1 does: the
    engine then drops it and re-calibrates through the exact-range path, and training goes on."""
    a, b, kf, cam = _projecting_pair(1, 3000, 77, 320, 240, False, z_shift=20_000.0)
    gt = torch.rand(3, cam.height, cam.width, device="cuda:0")
    for s in (a, b):
        s.opt.position_lr_init = s.opt.position_lr_final = 0.0
    la = [float(a.training_once([kf], [gt])) for _ in range(4)]
    lb = [float(b.training_once([kf], [gt])) for _ in range(4)]
    torch.cuda.synchronize()
    assert int((a.engine.radii > 0).sum()) > 0          # (something is rendered out there)
    assert a.dropped_steps() == b.dropped_steps() and a.dropped_steps() >= 1
    np.testing.assert_allclose(lb, la, rtol=1e-5)


def test_scaffold_step_with_projecting_forward_trains_like_the_unfused_step():
    a, b, kf, cam = _projecting_pair(0, 6000, 21, 320, 240, False)
    gt = torch.full((3, cam.height, cam.width), 0.4, device="cuda:0")
    la = [float(a.training_once([kf], [gt])) for _ in range(30)]
    lb = [float(b.training_once([kf], [gt])) for _ in range(30)]
    np.testing.assert_allclose(lb[:3], la[:3], rtol=1e-5)
    np.testing.assert_allclose(lb, la, rtol=2e-3)       # (the trajectories drift by the atomics' noise through Adam)
    assert lb[-1] < 0.85 * lb[0]


def _three_keyframe_step(seed):
    from segs_slam_amd import mapper_config as mc, neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    cfg = mc.load_committed_config("cfg/gaussian_mapper/RGB-D/Replica/office0.yaml")
    cfg.densify.update_until = 0
    cam = scenes.make_camera(320, 240, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    step = mc.make_mapper_step(cfg, ng.synthetic_model(4000, cfg.model, cam, dev, seed=seed), cam.width, cam.height)
    step.keyframe_for = lambda it, n: it % n
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kfs, gts = [], []
    g = torch.Generator().manual_seed(3)
    for k in range(3):
        ang = 0.05 * k
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=np.float32)
        c = scenes.make_camera(320, 240, 300.0, 300.0, R, np.array([0.03 * k, 0.0, 0.0], dtype=np.float32))
        kfs.append(ng.Keyframe(t(c.world_view_transform), t(c.full_proj_transform), t(c.camera_center),
                               torch.tensor([0.03 * k, 0.0, 0.0, 1.0, 0.0, ang, 0.0], device=dev), c.tanfovx, c.tanfovy))
        gts.append(torch.rand(3, 240, 320, generator=g).to(dev))
    return step, kfs, gts


@pytest.mark.parametrize("redo", [True, False])
def test_iteration_dropped_on_the_device_is_run_again_before_the_next_one(redo):
    """A forward whose instance count outgrows the resident capacity is dropped on the device (no statistics, no optimizer
    step).  The reference never skips an iteration (src/gaussian_mapper.cpp:823-1032): with one rank the host runs the dropped
    iteration again -- same keyframe, same iteration number -- as soon as it resolves that forward's status word, i.e. before the
    next iteration is queued, so the optimizer takes the reference's steps in the reference's order."""
    a, kfs, gts = _three_keyframe_step(5)
    b, _, _ = _three_keyframe_step(5)
    b.redo_dropped_steps = redo
    la, lb = [], []
    for it in range(8):
        if it == 4:
            assert b.engine.check() and b.engine.capacity > 0
            b.engine.capacity = max(b.engine.R // 3, 1024)       # the next forward overflows (its buffers stay as large as they are)
        la.append(float(a.training_once(kfs, gts)))
        lb.append(float(b.training_once(kfs, gts)))
    la.append(float(a.training_once(kfs, gts)))
    lb.append(float(b.training_once(kfs, gts)))                  # (the call that notices a drop of the last iteration, if any)
    torch.cuda.synchronize()
    assert a.dropped_steps() == 0 and a.redone_steps == 0 and a._mlp_count.value() == 9
    assert b.dropped_steps() == 1
    if redo:
        assert b.redone_steps == 1 and b._mlp_count.value() == 9
        keep = [i for i in range(9) if i != 4]                   # (the dropped pass returned the loss of an invalid image)
        np.testing.assert_allclose(np.array(lb)[keep], np.array(la)[keep], rtol=2e-4)
        pa, pb = a.model.params.cpu().numpy(), b.model.params.cpu().numpy()
        moved = np.abs(pa - pb) > 0
        assert np.isfinite(pb).all() and float(np.abs(pa - pb).max()) <= 2 * 9 * 0.02
        assert float((np.abs(pa - pb) > 1e-4 + 1e-2 * np.abs(pa)).mean()) < 2e-2, float(moved.mean())
    else:
        assert b.redone_steps == 0 and b._mlp_count.value() == 8   # one optimizer step fewer than the reference takes


def test_projecting_forward_rejects_inconsistent_arguments():
    """segs_neural_forward_projected fails loudly (status + message, nothing launched) on targets made for another image size,
    on a prefilter request without a radii array to fill, and on missing camera matrices."""
    import ctypes as C
    from segs_slam_amd import _capi, neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    a, b, kf, cam = _projecting_pair(1, 2000, 3, 320, 240, False)
    for _ in range(2):
        b.render(kf)
    torch.cuda.synchronize()
    assert b.engine.check() and b.engine.can_take_projected()
    lib, m, n = b._lib, b.model, b.neural
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731

    def call(tg, width, height, radii, rots, view):
        return lib.segs_neural_forward_projected(
            C.byref(m._cdims), m.A, p(m.param("anchor")), p(m.param("offset")), p(m.param("anchor_feat")), p(m.param("scaling")),
            p(radii), p(rots), p(m.mlp_params), p(kf.campos), p(kf.pose7), p(n.means3D), p(n.scales), p(n.rotations),
            p(n.neural_opacity), C.byref(tg) if tg is not None else None, p(view), p(kf.proj), width, height, float(kf.tanfovx),
            float(kf.tanfovy), 1.0, p(n.temp), None)

    tg = b.engine.projection_targets()
    rots = b._anchor_rotations()
    assert call(tg, 320, 240, b.visible_radii, rots, kf.view) == 0
    for bad in (call(tg, 640, 480, b.visible_radii, rots, kf.view),      # targets of a 320x240 engine
                call(tg, 320, 240, None, rots, kf.view),                 # prefilter folded in, nowhere to put the radii
                call(tg, 320, 240, b.visible_radii, rots, None),         # no view matrix
                call(None, 320, 240, b.visible_radii, rots, kf.view)):   # no targets
        assert bad != 0
        with pytest.raises(_capi.SegsError):
            _capi.check(bad, "segs_neural_forward_projected")
    torch.cuda.synchronize()
    assert torch.equal(a.render(kf), b.render(kf))             # (and the step still renders, bit for bit, afterwards)
    assert bool(torch.isfinite(b.engine.out_color).all())
