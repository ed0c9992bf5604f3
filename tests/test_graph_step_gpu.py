"""Whole-iteration hipGraph replay of the trainer / mapper steps (ScaffoldTrainerStep.enable_graph, TrainerStep.enable_graph)
against their eager paths.

The two paths launch the same kernels with the same arguments in the same order; what can differ between ANY two runs of this
step -- eager against eager as well -- is the summation order of the float atomics in the tile backward, which Adam (eps 1e-15)
turns into full-size steps of either sign on parameters whose gradient is rounding noise.  So: the forward (deterministic) is
compared bit for bit at the iteration the replay starts, losses to 1e-5 relative over the run, parameters with the bound of
tests/test_cpp_trainer.py (host against host), and the device-side step counts / learning-rate refresh exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _scaffold(seed, graph, densify):
    from segs_slam_amd import mapper_config as mc, neural_gaussians as ng, scenes
    dev = torch.device(DEV)
    cfg = mc.load_committed_config("cfg/gaussian_mapper/RGB-D/Replica/office0.yaml")
    cam = scenes.make_camera(320, 240, 300.0, 300.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    model = ng.synthetic_model(4000, cfg.model, cam, dev, seed=seed)
    if densify:
        cfg.densify.start_stat, cfg.densify.update_from, cfg.densify.update_interval = 3, 10, 12
        cfg.densify.voxel_size, cfg.densify.densify_grad_threshold = 0.01, 1e-7
    else:
        cfg.densify.update_until = 0
    step = mc.make_mapper_step(cfg, model, cam.width, cam.height)
    step.enable_graph(graph)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kfs, gts = [], []
    g = torch.Generator().manual_seed(3)
    for k in range(3):
        ang = 0.03 * k
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=np.float32)
        c = scenes.make_camera(320, 240, 300.0, 300.0, R, np.array([0.02 * k, 0.0, 0.0], dtype=np.float32))
        kfs.append(ng.Keyframe(t(c.world_view_transform), t(c.full_proj_transform), t(c.camera_center),
                               torch.tensor([0.02 * k, 0.0, 0.0, 1.0, 0.0, ang, 0.0], device=dev), c.tanfovx, c.tanfovy))
        gts.append(torch.rand(3, 240, 320, generator=g).to(dev))
    return step, kfs, gts


@pytest.mark.parametrize("start_iteration", [0, 10_000])       # without / with the frequency regulariser's window open
def test_scaffold_graph_replay_matches_eager(start_iteration):
    runs = {}
    for graph in (False, True):
        step, kfs, gts = _scaffold(5, graph, densify=False)
        step.iteration = start_iteration
        losses = [float(step.training_once(kfs, gts)) for _ in range(40)]
        torch.cuda.synchronize()
        runs[graph] = (losses, step.model.params.cpu().numpy().copy(), step)
    (le, pe, se), (lg, pg, sg) = runs[False], runs[True]
    assert se.graph_replays == 0 and sg.graph_replays >= 37         # the first iteration calibrates the rasterizer eagerly
    assert sg._mlp_count.value() == 40 == se._mlp_count.value() and sg.dropped_steps() == 0
    np.testing.assert_allclose(lg, le, rtol=1e-3)                   # (the trajectories drift apart by atomics noise through Adam)
    np.testing.assert_allclose(lg[:3], le[:3], rtol=1e-5)
    assert lg[-1] < lg[0]
    # (no per-parameter bound after 40 steps on a random target: two EAGER runs differ on most entries by then -- the update of
    # a parameter whose gradient is atomics noise has a random sign every step; the single-step comparison is in the next test)
    assert np.isfinite(pg).all() and np.isfinite(pe).all()


def test_scaffold_graph_forward_is_bit_identical_and_learning_rates_follow():
    """From identical parameters, the iteration replayed from the graph renders the eager iteration's image bit for bit (the
    forward has no atomics), for a keyframe and a target other than the ones the graph was captured with; and the learning
    rates the replay uses are this iteration's (an offset_lr schedule that moves visibly per step)."""
    from segs_slam_amd import neural_gaussians as ng
    step_g, kfs, gts = _scaffold(7, True, densify=False)
    step_e, _, _ = _scaffold(7, False, densify=False)
    for s in (step_g, step_e):
        s.opt.offset_lr_max_steps = 20            # a steep schedule: a stale learning rate would show
        s.keyframe_for = lambda it, n: it % n
    for it in range(6):
        # make both sides start this iteration from the SAME parameters and moments (the eager side's)
        for name in ("params", "exp_avg", "exp_avg_sq"):
            getattr(step_g.model, name).copy_(getattr(step_e.model, name))
        p_before = step_e.model.params.clone()
        le, lg = step_e.training_once(kfs, gts), step_g.training_once(kfs, gts)
        torch.cuda.synchronize()
        assert torch.equal(step_e.engine.out_color, step_g.engine.out_color), it
        assert float(le) == float(lg), it
        # the offset group's step is lr * m_hat / (sqrt(v_hat) + eps): its largest entry is this iteration's learning rate
        o, n = step_e.model.segments["offset"]
        ue = (step_e.model.params - p_before)[o:o + n].abs().max()
        ug = (step_g.model.params - p_before)[o:o + n].abs().max()
        lr = step_e.learning_rates(step_e.iteration)["offset"]
        assert abs(float(ue) - float(ug)) <= 1e-3 * float(ue) and 0.2 * lr < float(ug) <= 2.0 * lr, (it, float(ue), float(ug), lr)   # (|m_hat / sqrt(v_hat)| is of order 1, not bounded by it)
        # one step from identical parameters and moments: the whole update agrees entry by entry up to the atomics' noise
        upd_e, upd_g = (step_e.model.params - p_before).cpu().numpy(), (step_g.model.params - p_before).cpu().numpy()
        bad = np.abs(upd_g - upd_e) > 2e-3 * np.abs(upd_e) + 1e-6
        assert (np.abs(upd_e) > 0).mean() > 0.05 and bad.mean() < 1e-2, (it, float(bad.mean()))
    assert step_g.graph_replays >= 4


def test_scaffold_graph_survives_adjust_anchor_iterations():
    """adjust_anchor iterations (every 12th here) run eagerly, re-size the model and invalidate the captured graphs; replay
    resumes afterwards with the split anchor / MLP step counts."""
    step, kfs, gts = _scaffold(9, True, densify=True)
    A0 = step.model.A
    losses = [float(step.training_once(kfs, gts)) for _ in range(40)]
    torch.cuda.synchronize()
    assert np.isfinite(losses).all() and step.model.A != A0
    assert step.graph_replays >= 30 and step.dropped_steps() == 0
    assert step._anchor_count is not None and step._mlp_count.value() == 40 and step._anchor_count.value() == 37   # iterations 12, 24, 36 adjust
    assert bool(torch.isfinite(step.model.params).all())


def test_trainer_step_graph_replay_matches_eager():
    from segs_slam_amd import scenes
    from segs_slam_amd.gaussian_trainer import TrainerStep, keyframe_tensors
    dev = torch.device(DEV)
    runs = {}
    for graph in (False, True):
        sc = scenes.make_scene(20_000, 320, 240, 260.0, 260.0, seed=77)
        sc.scales *= 2.0
        step = TrainerStep.on_gpu(sc, dev)
        if graph:
            step.enable_graph()
        kfs = [keyframe_tensors(scenes.keyframe_camera(320, 240, 260.0, 260.0, 77, k), dev) for k in range(3)]
        gts = [torch.rand(3, 240, 320, generator=torch.Generator().manual_seed(k)).to(dev) for k in range(3)]
        p0 = step.params_flat.cpu().numpy().copy()
        losses = [float(step.training_once(kfs, gts)) for _ in range(30)]
        torch.cuda.synchronize()
        runs[graph] = (losses, step.params_flat.cpu().numpy().copy(), step, p0)
    (le, pe, se, p0), (lg, pg, sg, _) = runs[False], runs[True]
    assert sg.graph_replays >= 27 and sg.optimizer.step_count == 30 == se.optimizer.step_count
    np.testing.assert_allclose(lg, le, rtol=1e-3)
    np.testing.assert_allclose(lg[:3], le[:3], rtol=1e-5)
    assert np.isfinite(pg).all() and np.abs(pe - p0).max() > 0
