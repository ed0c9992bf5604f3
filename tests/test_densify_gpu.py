"""GPU parity of the anchor statistics / densification (include/segs_densify.h, segs-slam_amd/densify.py) against the
torch restatement of src/gaussian_model.cpp:1459-1762 (oracle/densify_ref.py).

Integer / index work (which voxels receive an anchor, their order, the per-voxel feature maximum, row compaction) is
compared exactly; the accumulated gradient norms to 1e-6 relative (sqrt rounding)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import densify_ref, neural_ref  # noqa: E402

DIMS = dict(feat_dim=32, n_offsets=10, appearance_dim=0, use_feat_bank=False)


def _model(A, seed, dev, capacity=None):
    from segs_slam_amd import neural_gaussians as ng
    rd = neural_ref.NeuralDims(**DIMS)
    g = torch.Generator().manual_seed(seed)
    anchor = torch.rand(A, 3, generator=g) - 0.5
    offset = torch.randn(A, 10, 3, generator=g)
    offset[::7] = 0.0                                        # candidates that fall into their parent's voxel
    feat = torch.randn(A, 32, generator=g)
    scaling_log = torch.log(0.02 + 0.08 * torch.rand(A, 6, generator=g))
    _, _, _, _, mlp = neural_ref.random_model(rd, 1, seed)
    model = ng.ScaffoldModel(A, ng.ModelDims(**DIMS), dev, capacity=capacity)
    model.load(anchor, offset, feat, scaling_log, mlp)
    return model, (anchor, offset, feat, scaling_log), g


def test_training_statis_matches_restatement():
    from segs_slam_amd import densify, neural_gaussians as ng
    dev = torch.device("cuda:0")
    A, no = 900, 10
    model, _, g = _model(A, 3, dev)
    dens = densify.AnchorDensifier(model)
    gen = ng.NeuralGaussians(model)
    visible = torch.rand(A, generator=g) < 0.6
    nop = torch.tanh(torch.randn(A * no, generator=g))
    nop[~visible.repeat_interleave(no)] = 0.0
    radii = (torch.rand(A * no, generator=g) < 0.5).to(torch.int32) * 4
    g2d = torch.randn(A * no, 3, generator=g)
    gen.neural_opacity.copy_(nop.view(-1, 1))
    vis_radii = visible.to(torch.int32) * 2
    init = {k: torch.rand_like(v.cpu()) for k, v in dens._stats.items()}
    for k, v in init.items():
        dens._stats[k].copy_(v)
    dens.training_statis(gen, vis_radii.to(dev), radii.to(dev), g2d.to(dev))
    torch.cuda.synchronize()

    st = densify_ref.DensifyState(params={}, exp_avg={}, exp_avg_sq={}, opacity_accum=init["opacity_accum"].view(-1, 1).clone(),
                                  anchor_demon=init["anchor_demon"].view(-1, 1).clone(),
                                  offset_gradient_accum=init["offset_gradient_accum"].view(-1, 1).clone(),
                                  offset_denom=init["offset_denom"].view(-1, 1).clone())
    vis_rows = visible.repeat_interleave(no)
    mask = nop[vis_rows] > 0                                   # offset_selection_mask over the visible anchors' slots
    full_mask = torch.zeros(A * no, dtype=torch.bool)
    full_mask[vis_rows] = mask
    densify_ref.training_statis(st, g2d[full_mask], nop[vis_rows].view(-1, 1), radii[full_mask] > 0, mask, visible)
    np.testing.assert_array_equal(dens.stat("anchor_demon").cpu().numpy(), st.anchor_demon.numpy())
    np.testing.assert_array_equal(dens.stat("offset_denom").cpu().numpy(), st.offset_denom.numpy())
    np.testing.assert_allclose(dens.stat("opacity_accum").cpu().numpy(), st.opacity_accum.numpy(), rtol=1e-6)
    np.testing.assert_allclose(dens.stat("offset_gradient_accum").cpu().numpy(), st.offset_gradient_accum.numpy(), rtol=1e-6)


@pytest.mark.parametrize("A,seed,capacity", [(600, 1, None), (2500, 2, 9000), (40, 5, None)])
def test_adjust_anchor_matches_restatement(A, seed, capacity):
    from segs_slam_amd import densify
    dev = torch.device("cuda:0")
    no = 10
    model, (anchor, offset, feat, scaling_log), g = _model(A, seed, dev, capacity)
    P = densify.DensifyParams(voxel_size=0.01, update_depth=3, update_init_factor=16, update_hierachy_factor=4)
    dens = densify.AnchorDensifier(model, P)
    # statistics: about half of the offsets eligible, a third of the anchors old enough to be judged
    denom = torch.floor(torch.rand(A * no, generator=g) * 100)
    accum = torch.rand(A * no, generator=g) * denom * 0.0005
    demon = torch.floor(torch.rand(A, generator=g) * 160)
    opac = torch.rand(A, generator=g) * demon * 0.01
    for k, v in (("offset_denom", denom), ("offset_gradient_accum", accum), ("anchor_demon", demon), ("opacity_accum", opac)):
        dens._stats[k][:v.numel()] = v.to(dev)
    m_rand = {n: torch.randn(A, w, generator=g) for n, w in model.widths.items()}
    v_rand = {n: torch.rand(A, w, generator=g) for n, w in model.widths.items()}
    for n in model.widths:
        model._view(model.exp_avg, n).copy_(m_rand[n].view(model._view(model.exp_avg, n).shape))
        model._view(model.exp_avg_sq, n).copy_(v_rand[n].view(model._view(model.exp_avg_sq, n).shape))
    rands = [torch.rand(A * no, generator=g) for _ in range(3)]

    rot = torch.zeros(A, 4); rot[:, 0] = 1.0
    ref = densify_ref.DensifyState(
        params={"anchor": anchor.clone(), "offset": offset.clone(), "anchor_feat": feat.clone(), "opacity": torch.zeros(A, 1),
                "scaling": scaling_log.clone(), "rotation": rot},
        exp_avg={"anchor": m_rand["anchor"].clone(), "offset": m_rand["offset"].view(A, no, 3).clone(),
                 "anchor_feat": m_rand["anchor_feat"].clone(), "scaling": m_rand["scaling"].clone()},
        exp_avg_sq={"anchor": v_rand["anchor"].clone(), "offset": v_rand["offset"].view(A, no, 3).clone(),
                    "anchor_feat": v_rand["anchor_feat"].clone(), "scaling": v_rand["scaling"].clone()},
        opacity_accum=opac.view(-1, 1).clone(), anchor_demon=demon.view(-1, 1).clone(),
        offset_gradient_accum=accum.view(-1, 1).clone(), offset_denom=denom.view(-1, 1).clone(),
        voxel_size=0.01, update_depth=3, update_init_factor=16, update_hierachy_factor=4)
    ref_prune = densify_ref.adjust_anchor(ref, 100, 0.8, 0.0002, 0.005, rands)

    prune = dens.adjust_anchor(100, 0.8, 0.0002, 0.005, rands=[r.to(dev) for r in rands])
    torch.cuda.synchronize()
    A1 = ref.params["anchor"].shape[0]
    grown = ref_prune.shape[0] - A
    assert grown > 0 and int(ref_prune.sum()) > 0, "the case must exercise both growing and pruning"
    assert model.A == A1
    np.testing.assert_array_equal(prune.cpu().numpy(), ref_prune.numpy())
    eq = lambda a, b, msg: np.testing.assert_array_equal(a.cpu().numpy(), b.numpy(), err_msg=msg)  # noqa: E731
    eq(model.param("anchor"), ref.params["anchor"], "anchor")
    eq(model.param("offset"), ref.params["offset"], "offset")
    eq(model.param("anchor_feat"), ref.params["anchor_feat"], "anchor_feat")
    np.testing.assert_allclose(model.param("scaling").cpu().numpy(), ref.params["scaling"].numpy(), rtol=0, atol=1e-6)
    eq(model.rotation[:A1], ref.params["rotation"], "rotation")
    np.testing.assert_allclose(model.opacity[:A1].cpu().numpy(), ref.params["opacity"].numpy(), atol=1e-6)
    for n in ("anchor", "offset", "anchor_feat", "scaling"):
        eq(model._view(model.exp_avg, n), ref.exp_avg[n], "exp_avg " + n)
        eq(model._view(model.exp_avg_sq, n), ref.exp_avg_sq[n], "exp_avg_sq " + n)
    for k in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        eq(dens.stat(k), getattr(ref, k), k)


def test_trainer_keeps_running_through_densification():
    from segs_slam_amd import densify, neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    cam = scenes.make_camera(320, 240, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    model = ng.synthetic_model(3000, ng.ModelDims(), cam, dev, seed=4)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    step = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
    dens = densify.AnchorDensifier(model, densify.DensifyParams(voxel_size=0.01, start_stat=2, update_from=5, update_interval=10,
                                                                update_until=1000, densify_grad_threshold=1e-7))
    step.enable_densification(dens, seed=0)
    gt = torch.full((3, cam.height, cam.width), 0.4, device=dev)
    sizes = []
    for it in range(1, 32):
        loss = step.training_once([kf], [gt])
        sizes.append(model.A)
        assert np.isfinite(float(loss))
    assert sizes[-1] != 3000, sizes[::5]      # the map changed size and the step kept running
    # anchor tensors are skipped by Adam at the 3 densify iterations; a pass right after the map grew may outgrow the resident
    # scratch and is then dropped ON THE DEVICE (both counts stay put), so the counts say how many steps were really taken
    mlp, anchor = step._mlp_count.value(), step._anchor_count.value()
    assert mlp - anchor == 3 and 28 <= mlp <= 31, (mlp, anchor)


def test_trainer_survives_a_map_pruned_to_nothing():
    """All anchors pruned: the reference's rasterizer short-circuits P == 0 to a zero image (src/rasterize_points.cu:81);
    the step must keep running (loss against the zero image, no gradient) rather than fault."""
    from segs_slam_amd import densify, neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    cam = scenes.make_camera(96, 64, 90.0, 90.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    model = ng.synthetic_model(200, ng.ModelDims(), cam, dev, seed=5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    step = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
    dens = densify.AnchorDensifier(model, densify.DensifyParams(voxel_size=0.01, start_stat=2, update_from=5, update_interval=10,
                                                                update_until=1000))
    step.enable_densification(dens, seed=0)
    gt = torch.full((3, cam.height, cam.width), 0.4, device=dev)
    step.training_once([kf], [gt])
    dens.prune_anchor(torch.ones(model.A, dtype=torch.bool, device=dev))
    assert model.A == 0
    for _ in range(12):                      # crosses a densify iteration with A == 0
        loss = step.training_once([kf], [gt])
    torch.cuda.synchronize()
    assert abs(float(loss) - (0.8 * 0.4 + 0.2 * 1.0)) < 0.05      # L1 = 0.4 against zeros, SSIM ~ 0


def test_create_from_pcd_and_increase_pcd_match_restatement():
    """GaussianModel::createFromPcd / increasePcd (src/gaussian_model.cpp:327-381, 443-520) restated on the CPU: voxel centres
    by torch.unique over round(points / voxel_size), scales from the oracle's simple-knn (tests/test_points.py pins the GPU
    kernel to it bit for bit)."""
    from oracle import gs_oracle
    from segs_slam_amd import densify, neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    vs = 0.05
    pts = torch.rand(6000, 3, generator=g) * torch.tensor([2.0, 1.5, 1.0]) + torch.tensor([-1.0, -0.75, 2.0])
    pts = torch.cat([pts, pts[:1500] + 0.004])                 # near-duplicates: many land in an occupied voxel

    def restate(p):
        u = (torch.unique(torch.round(p / vs), dim=0, sorted=True) * vs).to(torch.float32)
        d2 = torch.from_numpy(gs_oracle.knn_mean_dist2(u.numpy())).clamp_min(0.0000001)
        return u, torch.log(torch.sqrt(d2)).unsqueeze(1).repeat(1, 6)

    model = ng.create_from_pcd(pts, ng.ModelDims(), vs, dev)
    u, sc = restate(pts)
    A = u.shape[0]
    assert model.A == A and A < pts.shape[0]
    assert torch.equal(model.param("anchor").cpu(), u)
    assert torch.allclose(model.param("scaling").cpu(), sc, rtol=1e-6, atol=1e-6)
    assert float(model.param("offset").abs().max()) == 0.0 and float(model.param("anchor_feat").abs().max()) == 0.0
    assert torch.allclose(model.opacity[:A].cpu(), torch.full((A, 1), float(np.log(0.1 / 0.9))), atol=1e-6)
    assert torch.equal(model.rotation[:A].cpu(), torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(A, 1))
    w = model.param("mlp_cov.0.weight")
    assert float(w.abs().max()) <= 1.0 / np.sqrt(w.shape[1]) and float(w.abs().max()) > 0.0      # nn::Linear's default range

    # increasePcd: new voxels appended (unique among themselves only), counters and Adam moments of the new rows zero
    dens = densify.AnchorDensifier(model, densify.DensifyParams(voxel_size=vs))
    model.exp_avg.fill_(0.5)
    new = torch.rand(900, 3, generator=g) * torch.tensor([1.0, 1.0, 0.5]) + torch.tensor([1.2, -0.5, 2.2])
    new = torch.cat([new, pts[:50]])                            # some fall into voxels that already hold an anchor: kept, like the reference
    n_new = dens.increase_pcd(new.to(dev))
    u2, sc2 = restate(new)
    assert n_new == u2.shape[0] and model.A == A + n_new
    assert torch.equal(model.param("anchor")[A:].cpu(), u2) and torch.equal(model.param("anchor")[:A].cpu(), u)
    assert torch.allclose(model.param("scaling")[A:].cpu(), sc2, rtol=1e-6, atol=1e-6)
    for name in model.widths:
        assert float(model._view(model.exp_avg, name)[A:].abs().max()) == 0.0
        assert float(model._view(model.exp_avg, name)[:A].min()) == 0.5
    assert float(dens.stat("offset_denom")[A * 10:].abs().max()) == 0.0 and float(dens.stat("anchor_demon")[A:].abs().max()) == 0.0
    assert dens.increase_pcd(torch.zeros(0, 3, device=dev)) == 0

    # the grown model trains
    cam = scenes.make_camera(160, 120, 150.0, 150.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    model.exp_avg.zero_()
    step = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
    gt = torch.full((3, cam.height, cam.width), 0.5, device=dev)
    losses = [float(step.training_once([kf], [gt])) for _ in range(30)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_coarse_anchor_set_is_created_and_grown_like_the_reference():
    """Model.use_coarse_anchor = 1: createCoarseAnchorFromPcd / increasePcdCoarse (src/gaussian_model.cpp:288-325, 383-441)
    restated with torch.unique and the oracle's simple-knn, quirks included -- the coarse set is rounded at coarse_voxel_size but
    PLACED at unique * voxel_size (:290-291), and grown at the fine voxel size (:385-386); offsets and features take the FINE
    n_offsets / feat_dim.  The set is an inert payload: the step trains exactly as without it and never touches it."""
    from oracle import gs_oracle
    from segs_slam_amd import coarse_anchors as ca, densify, mapper_config as mc, neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    cfg = mc.load_committed_config("cfg/colmap/gaussian_splatting.yaml")
    g = torch.Generator().manual_seed(21)
    vs, cvs = 0.05, cfg.coarse.coarse_voxel_size
    pts = torch.rand(8000, 3, generator=g) * torch.tensor([2.0, 1.5, 1.0]) + torch.tensor([-1.0, -0.75, 2.0])

    def restate(p, round_size, place_size):
        u = (torch.unique(torch.round(p / round_size), dim=0, sorted=True) * place_size).to(torch.float32)
        d2 = torch.from_numpy(gs_oracle.knn_mean_dist2(u.numpy())).clamp_min(0.0000001)
        return u, torch.log(torch.sqrt(d2)).unsqueeze(1).repeat(1, 6)

    model = ng.create_from_pcd(pts, cfg.model, vs, dev, coarse=cfg.coarse)
    plain = ng.create_from_pcd(pts, cfg.model, vs, dev)
    c = model.coarse
    u, sc = restate(pts, cvs, vs)
    assert isinstance(c, ca.CoarseAnchors) and c.n == u.shape[0] and 0 < c.n < model.A          # 0.2-m voxels: far fewer than the fine set
    assert torch.equal(c.anchor.cpu(), u) and torch.allclose(c.scaling.cpu(), sc, rtol=1e-6, atol=1e-6)
    assert c.offset.shape == (c.n, cfg.model.n_offsets, 3) and c.anchor_feat.shape == (c.n, cfg.model.feat_dim)
    assert float(c.offset.abs().max()) == 0.0 and float(c.anchor_feat.abs().max()) == 0.0
    assert torch.equal(c.rotation.cpu(), torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(c.n, 1))
    assert torch.allclose(c.opacity.cpu(), torch.full((c.n, 1), float(np.log(0.1 / 0.9))), atol=1e-6)
    for name, shape in ca.coarse_mlp_shapes(cfg.model, cfg.coarse).items():
        assert tuple(c.mlp[name].shape) == shape and float(c.mlp[name].abs().max()) > 0
    names = [n for n, _, _ in c.optimizer_groups(0)]
    assert names == ["anchor_c", "offset_c", "anchor_feat_c", "opacity_c", "scaling_c", "rotation_c", "mlp_opacity_c", "mlp_cov_c",
                     "mlp_color_c", "mlp_apperance_c"]                                           # :730-758 (appearance, no feature bank)
    lr = {n: v for n, _, v in c.optimizer_groups(15000)}
    assert lr["anchor_c"] == 0.0 and lr["offset_c"] == pytest.approx(0.001) and lr["mlp_cov_c"] == pytest.approx(0.004)
    # the fine set is what it is without the coarse one
    assert torch.equal(model.params, plain.params)

    # increasePcd -> increasePcdCoarse: the same new points, rounded and placed at the FINE voxel size, appended
    dens = densify.AnchorDensifier(model, densify.DensifyParams(voxel_size=vs))
    new = torch.rand(700, 3, generator=g) * torch.tensor([1.0, 1.0, 0.5]) + torch.tensor([1.2, -0.5, 2.2])
    n0 = c.n
    n_new = dens.increase_pcd(new.to(dev))
    u2, sc2 = restate(new, vs, vs)
    assert n_new == u2.shape[0] and c.n == n0 + u2.shape[0]
    assert torch.equal(c.anchor[n0:].cpu(), u2) and torch.equal(c.anchor[:n0].cpu(), u)
    assert torch.allclose(c.scaling[n0:].cpu(), sc2, rtol=1e-6, atol=1e-6) and c.max_radii2D.shape == (c.n,)

    # a mapper step of this configuration takes the model; training leaves the coarse set bit for bit alone
    cam = scenes.make_camera(160, 120, 150.0, 150.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    step = mc.make_mapper_step(cfg, model, cam.width, cam.height)
    before = {k: v.clone() for k, v in (("anchor", c.anchor), ("scaling", c.scaling), ("w", c.mlp["mlp_cov_c.0.weight"]))}
    gt = torch.full((3, cam.height, cam.width), 0.5, device=dev)
    losses = [float(step.training_once([kf], [gt])) for _ in range(10)]
    assert np.isfinite(losses).all()
    assert torch.equal(before["anchor"], c.anchor) and torch.equal(before["scaling"], c.scaling) and torch.equal(before["w"], c.mlp["mlp_cov_c.0.weight"])
