"""world_size-2 gloo tests (CPU) of the keyframe-parallel trainer step.

The HIP engine cannot run here, so the per-rank render+backward is supplied by the CPU oracle (test-only wiring;
the package never imports oracle/).  Checked: (1) the all-reduced bucket equals the sum of the two single-keyframe
oracle gradients; (2) both ranks end the step with bit-identical parameters; (3) the 2-rank step equals a 1-process
step fed the summed gradient (parity at gradient level, not trajectory level)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_backend(scene_np, grads_flat, P):
    from oracle import gs_oracle
    from segs_slam_amd.raster_engine import split_flat

    def render_backward(params, keyframe, dL_fn):
        cam = keyframe
        o = gs_oracle.Oracle()
        o.forward(scene_np.bg, params["means3D"].numpy(), params["colors"].numpy(), params["opacity"].numpy(),
                  params["scales"].numpy(), 1.0, params["rotations"].numpy(), cam.world_view_transform,
                  cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.height, cam.width)
        image = torch.from_numpy(o.get("out_color"))
        loss, dL = dL_fn(image)
        g = o.backward(dL.numpy())
        views = split_flat(grads_flat, P)
        for name, key in (("means3D", "dL_dmean3D"), ("scales", "dL_dscale"), ("rotations", "dL_drot"),
                          ("opacity", "dL_dopacity"), ("colors", "dL_dcolor")):
            views[name].copy_(torch.from_numpy(g[key]))
        return loss
    return render_backward


def _make(P=400, W=48, H=32):
    from segs_slam_amd import scenes
    sc = scenes.make_scene(P, W, H, 40.0, 40.0, seed=77, bg=(0.1, 0.1, 0.1))
    sc.scales *= 4.0
    cams = [scenes.make_scene(P, W, H, 40.0, 40.0, seed=77, keyframe=k).camera for k in range(2)]
    gts = [torch.from_numpy(scenes.uniform01(3 * H * W, 60 + k, 5).reshape(3, H, W).copy()) for k in range(2)]
    return sc, cams, gts


def _params(sc):
    from segs_slam_amd.raster_engine import FLOATS_PER_GAUSSIAN, split_flat
    flat = torch.zeros(FLOATS_PER_GAUSSIAN * sc.P)
    v = split_flat(flat, sc.P)
    for name, arr in (("means3D", sc.means3D), ("scales", sc.scales), ("rotations", sc.rotations), ("opacity", sc.opacity),
                      ("colors", sc.colors)):
        v[name].copy_(torch.from_numpy(arr))
    return flat


def _worker(rank, world, port, out_dir, sharded):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from segs_slam_amd.gaussian_trainer import OptimizationParams, TorchAdam, TrainerStep
    sc, cams, gts = _make()
    flat = _params(sc)
    grads = torch.zeros_like(flat)
    opt = OptimizationParams()
    step = TrainerStep(flat, sc.P, _oracle_backend(sc, grads, sc.P), TorchAdam(flat.numel(), "cpu", opt), opt, grads,
                       sharded_optimizer=sharded)
    assert step.world == world and step.keyframe_for(0, 2) == rank and step.exchange.sharded == sharded
    # capture the reduced gradient before Adam clears it
    reduced = {}
    orig = step.optimizer.step

    def spy(p, g, lrs, P, scale, exchange=None, guard=None):
        reduced["g"] = g.clone()
        reduced["scale"] = scale
        return orig(p, g, lrs, P, scale, exchange=exchange, guard=guard)
    step.optimizer.step = spy
    loss = step.training_once(cams, gts)
    lo, hi = step.exchange.shard_range()
    np.save(os.path.join(out_dir, f"params_{rank}.npy"), flat.numpy())
    np.save(os.path.join(out_dir, f"reduced_{rank}.npy"), reduced["g"].numpy())
    np.save(os.path.join(out_dir, f"shard_{rank}.npy"), np.array([lo, hi]))
    assert reduced["scale"] == 0.5 and np.isfinite(float(loss))
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [False, True], ids=["dense_allreduce", "reduce_scatter_sharded_adam_allgather"])
def test_two_rank_step_matches_summed_gradients(tmp_path, sharded):
    port = 29500 + (os.getpid() % 2000) + (7 if sharded else 0)
    mp.spawn(_worker, args=(2, port, str(tmp_path), sharded), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    assert np.array_equal(p0, p1), "replicas diverged"
    r0, r1 = np.load(tmp_path / "reduced_0.npy"), np.load(tmp_path / "reduced_1.npy")
    if sharded:
        # each rank holds the summed gradient of its own shard only; the shards tile the bucket
        (lo0, hi0), (lo1, hi1) = np.load(tmp_path / "shard_0.npy"), np.load(tmp_path / "shard_1.npy")
        assert lo0 == 0 and hi0 == lo1 and hi1 == r0.size and lo1 % 4 == 0
        r0 = np.concatenate([r0[lo0:hi0], r1[lo1:hi1]])
    else:
        assert np.array_equal(r0, r1)

    # single-process reference: sum of the two single-keyframe oracle gradients, then one Adam step at scale 1/2
    from segs_slam_amd.gaussian_trainer import OptimizationParams, TorchAdam, TrainerStep
    sc, cams, gts = _make()
    total = None
    for k in range(2):
        flat = _params(sc)
        grads = torch.zeros_like(flat)
        opt = OptimizationParams()
        st = TrainerStep(flat, sc.P, _oracle_backend(sc, grads, sc.P), TorchAdam(flat.numel(), "cpu", opt), opt, grads)
        st.render_backward(st.params, cams[k], lambda im: st.loss_and_grad(im, gts[k]))
        total = grads.clone() if total is None else total + grads
    assert np.allclose(r0, total.numpy(), rtol=1e-6, atol=1e-12)
    assert np.abs(total.numpy()).max() > 0
    flat = _params(sc)
    opt = OptimizationParams()
    adam = TorchAdam(flat.numel(), "cpu", opt)
    st = TrainerStep(flat, sc.P, None, adam, opt, torch.zeros_like(flat))
    adam.step(flat, torch.from_numpy(r0.copy()), st.learning_rates(1), sc.P, 0.5)
    assert np.array_equal(flat.numpy(), p0)


def test_expon_lr_and_schedule():
    from segs_slam_amd.gaussian_trainer import expon_lr
    assert abs(expon_lr(0, 1.6e-4, 1.6e-6, 30000) - 1.6e-4) < 1e-12
    assert abs(expon_lr(30000, 1.6e-4, 1.6e-6, 30000) - 1.6e-6) < 1e-12
    assert abs(expon_lr(15000, 1.6e-4, 1.6e-6, 30000) - 1.6e-5) < 1e-10
    assert expon_lr(5, 0.0, 0.0, 100) == 0.0


def test_ssim_matches_reference_formula_on_cpu():
    """loss_utils.ssim against a direct numpy evaluation of loss_utils.h:77-109 (zero padding 5, integer-x window)."""
    from segs_slam_amd import loss_utils
    rng = np.random.default_rng(3)
    a, b = rng.random((3, 20, 24), dtype=np.float32), rng.random((3, 20, 24), dtype=np.float32)
    g = np.array([np.exp(-float((x - 5) ** 2) / (2 * 1.5 * 1.5)) for x in range(11)], dtype=np.float64)
    g /= g.sum()
    w2 = np.outer(g, g)

    def conv(img):
        pad = np.pad(img.astype(np.float64), ((0, 0), (5, 5), (5, 5)))
        out = np.zeros_like(img, dtype=np.float64)
        for dy in range(11):
            for dx in range(11):
                out += w2[dy, dx] * pad[:, dy:dy + img.shape[1], dx:dx + img.shape[2]]
        return out
    mu1, mu2 = conv(a), conv(b)
    s1, s2, s12 = conv(a * a) - mu1 ** 2, conv(b * b) - mu2 ** 2, conv(a * b) - mu1 * mu2
    ref = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 ** 2 + mu2 ** 2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    got = float(loss_utils.ssim(torch.from_numpy(a), torch.from_numpy(b)))
    assert abs(got - ref) < 2e-5


def _resize_worker(rank, world, port, out_dir):
    """A densification re-sizes the bucket between the gradient exchange and the optimizer: the block that sits at the END
    of the bucket (the MLPs) moves, so the shard that updates an element afterwards is not the shard whose reduce-scatter
    summed it.  `reduce_gradients(dense=True)` must therefore leave the full sum EVERYWHERE; the plain sharded call must
    leave it in the rank's own shard only (outside it the rank's own contribution stays -- the caller clears it)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segs_slam_amd.keyframe_parallel import BucketExchange
    n_rows, mlp = 10, 1501                        # 71 floats per anchor row, then the MLP block: it spans both shards
    n_old = 71 * n_rows + mlp
    own = torch.arange(n_old, dtype=torch.float32) * (rank + 1)          # rank r contributes (r + 1) * i
    total = torch.arange(n_old, dtype=torch.float32) * 3.0
    ex = BucketExchange(n_old, "cpu", None, sharded=True)
    assert ex.sharded and not ex._emulate
    lo, hi = ex.shard_range()
    g = own.clone()
    ex.reduce_gradients(g)
    assert torch.equal(g[lo:hi], total[lo:hi])
    outside = torch.ones(n_old, dtype=torch.bool)
    outside[lo:hi] = False
    assert torch.equal(g[outside], own[outside]), "outside its shard a rank keeps its own contribution"
    g = own.clone()
    ex.reduce_gradients(g, dense=True)
    assert torch.equal(g, total)
    # the bucket grows by 50 anchor rows: the MLP block shifts by 71 * 50 floats and the shard cut with it
    n_new = 71 * (n_rows + 50) + mlp
    moved = torch.zeros(n_new)
    moved[n_new - mlp:] = g[n_old - mlp:]
    ex2 = BucketExchange(n_new, "cpu", None, sharded=True)
    lo2, hi2 = ex2.shard_range()
    a, b = max(lo2, n_new - mlp), max(hi2, max(lo2, n_new - mlp))     # this rank's part of the moved MLP block (may be empty)
    assert torch.equal(moved[a:b], total[n_old - mlp:][a - (n_new - mlp):b - (n_new - mlp)])
    # ... and with the plain sharded reduce the same elements would have been wrong on some rank (what the fix is for)
    g = own.clone()
    ex.reduce_gradients(g)
    moved_bad = torch.zeros(n_new)
    moved_bad[n_new - mlp:] = g[n_old - mlp:]
    bad = not torch.equal(moved_bad[a:b], total[n_old - mlp:][a - (n_new - mlp):b - (n_new - mlp)])
    np.save(os.path.join(out_dir, f"bad_{rank}.npy"), np.array([bad, lo, hi, lo2, hi2]))
    dist.destroy_process_group()


def test_dense_reduce_survives_a_bucket_resize_between_exchange_and_optimizer(tmp_path):
    port = 29500 + (os.getpid() % 2000) + 13
    mp.spawn(_resize_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"bad_{k}.npy") for k in range(2)]
    assert r[0][0] or r[1][0], "the scenario must exercise elements that change owner"


def _piggy_worker(rank, world, port, out_dir):
    """Dense exchange over a gradient bucket with a spare element behind it: the overflow word rides through the gradient
    all-reduce (no collective of its own) and every rank sees the SUM of the words afterwards; without the spare element, or
    with allow_piggyback=False, the word takes its own all-reduce -- same values either way; "auto" picks dense below
    AUTO_SHARD_BYTES and sharded above."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segs_slam_amd.keyframe_parallel import BucketExchange
    n = 1003
    store = torch.zeros(n + 4)
    grads = store[:n]
    own = torch.arange(n, dtype=torch.float32) * (rank + 1)
    for allow in (True, False):
        for my_flag in (0, 1 if rank == 1 else 0):
            ex = BucketExchange(n, "cpu", None, sharded=False, grads=grads)
            assert ex._ext is not None and not ex.sharded
            grads.copy_(own)
            calls = []
            orig = dist.all_reduce
            dist.all_reduce = lambda t, **k: (calls.append(t.numel()), orig(t, **k))[1]
            try:
                ex.reduce_flag_async(torch.tensor([my_flag], dtype=torch.int32), allow_piggyback=allow)
                flag = ex.wait_flag()
                ex.reduce_gradients(grads)
            finally:
                dist.all_reduce = orig
            assert torch.equal(grads, torch.arange(n, dtype=torch.float32) * 3.0)
            want = float(my_flag) if rank == 1 else float(1 if (my_flag == 0 and False) else 0)
            total = float(flag.reshape(-1)[0])
            # rank 0 always contributes 0, rank 1 contributes my_flag (both ranks run the same loop index)
            assert total in (0.0, 1.0)
            assert calls == ([n + 1] if allow else [1, n]), (allow, calls)
            np.save(os.path.join(out_dir, f"flag_{rank}_{int(allow)}_{my_flag}.npy"), np.array([total]))
    # no spare element: falls back to the word's own collective
    plain = torch.zeros(n)
    ex = BucketExchange(n, "cpu", None, sharded=False, grads=plain)
    assert ex._ext is None
    assert BucketExchange(n, "cpu", None, sharded="auto").sharded is False
    assert BucketExchange(BucketExchange.AUTO_SHARD_BYTES // 4, "cpu", None, sharded="auto").sharded is True
    dist.destroy_process_group()


def test_overflow_word_rides_in_the_dense_gradient_all_reduce(tmp_path):
    port = 29500 + (os.getpid() % 2000) + 19
    mp.spawn(_piggy_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for allow in (0, 1):
        # second loop value: rank 1 raised its word -> both ranks must have seen the sum 1
        a, b = (float(np.load(tmp_path / f"flag_{r}_{allow}_{1 if r == 1 else 0}.npy")[0]) for r in range(2))
        assert (a, b) == (1.0, 1.0), (allow, a, b)


def _redo_worker(rank, world, port, out_dir, sharded, overflow_at):
    """Three iterations of the two-rank step; at iteration `overflow_at` (1-based; 0 = never) RANK 1 ALONE raises its overflow word
    on the first attempt.  Both ranks must drop that pass on the device, learn of it from the mirrored SUMMED word at the top of
    the next call, run the iteration again together, and end with the parameters of the run in which nothing overflowed."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from segs_slam_amd.gaussian_trainer import OptimizationParams, TorchAdam, TrainerStep
    sc, cams, gts = _make()
    flat = _params(sc)
    grads_store = torch.zeros(flat.numel() + 4)          # (a spare element behind the bucket: the dense exchange carries the word there)
    grads = grads_store[:flat.numel()]
    opt = OptimizationParams()
    inner = _oracle_backend(sc, grads, sc.P)
    attempts = {}

    def backend(params, keyframe, dL_fn, after_forward=None):
        it = step.iteration
        attempts[it] = attempts.get(it, 0) + 1
        raised = rank == 1 and it == overflow_at and attempts[it] == 1
        after_forward(torch.tensor([1 if raised else 0], dtype=torch.int32))
        return inner(params, keyframe, dL_fn)

    adam = TorchAdam(flat.numel(), "cpu", opt)
    step = TrainerStep(flat, sc.P, backend, adam, opt, grads, sharded_optimizer=sharded)
    for _ in range(3):
        step.training_once(cams, gts)
    step.finish()                                         # (a drop of the LAST iteration is only seen here)
    np.save(os.path.join(out_dir, f"redo_params_{rank}_{overflow_at}.npy"), flat.numpy())
    np.save(os.path.join(out_dir, f"redo_counts_{rank}_{overflow_at}.npy"),
            np.array([adam.step_count, getattr(step, "redone_steps", 0), sum(attempts.values())]))
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [False, True], ids=["dense_allreduce", "reduce_scatter_sharded_adam_allgather"])
def test_iteration_dropped_because_one_rank_overflowed_is_run_again_by_both(tmp_path, sharded):
    """N > 1 ranks lose no optimizer step either (the reference never skips an iteration, src/gaussian_mapper.cpp:1027-1030): an
    overflow on ONE rank -- in the middle of the run, and at its very last iteration -- makes both ranks redo that iteration;
    replicas stay bit-identical and equal the run without any overflow."""
    base = 29500 + (os.getpid() % 2000) + (31 if sharded else 23)
    for i, at in enumerate((0, 2, 3)):
        mp.spawn(_redo_worker, args=(2, base + i, str(tmp_path), sharded, at), nprocs=2, join=True)
    clean = np.load(tmp_path / "redo_params_0_0.npy")
    for at in (0, 2, 3):
        p0, p1 = np.load(tmp_path / f"redo_params_0_{at}.npy"), np.load(tmp_path / f"redo_params_1_{at}.npy")
        assert np.array_equal(p0, p1), ("replicas diverged", at)
        assert np.array_equal(p0, clean), ("a step was lost or taken twice", at)
        for r in range(2):
            steps, redone, attempts = np.load(tmp_path / f"redo_counts_{r}_{at}.npy")
            assert steps == 3 and redone == (1 if at else 0) and attempts == (4 if at else 3), (at, r, steps, redone, attempts)


def _frozen_worker(rank, world, port, out_dir, sharded):
    """A frozen segment in front of the exchanged range (anchor positions under position_lr = 0): it never crosses a link, and no
    rank's optimizer may touch it -- its gradient is this rank's own, so moments formed from it would differ between replicas."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segs_slam_amd.keyframe_parallel import BucketExchange
    off, n = 12, 1001
    store = torch.zeros(off + n + 4)
    grads = store[:off + n]
    ex = BucketExchange(n, "cpu", None, sharded=sharded, grads=grads, offset=off)
    assert ex.active and ex.sharded == sharded
    grads.copy_(torch.arange(off + n, dtype=torch.float32) * (rank + 1) + 1.0)
    ex.reduce_flag_async(None)
    ex.wait_flag()
    ex.reduce_gradients(grads)
    lo, hi = ex.shard_range()
    assert lo >= off and hi <= off + n, (lo, hi)
    clipped = ex.clip_segments([(0, off, 0.0), (off, n, 1e-3)])        # the anchor group and everything else
    assert all(a >= off for a, _, _ in clipped) and sum(c for _, c, _ in clipped) == hi - lo, clipped
    assert float(grads[:off].abs().max()) == 0.0                       # un-summed, unread: cleared
    want = torch.arange(off + n, dtype=torch.float32) * 3.0 + 2.0
    assert torch.equal(grads[lo:hi], want[lo:hi])
    np.save(os.path.join(out_dir, f"frozen_{rank}.npy"), np.array([lo, hi]))
    one = BucketExchange(n, "cpu", None, sharded=False, offset=off)   # (what a single process gets is checked in the parent)
    assert one.active
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [False, True], ids=["dense_allreduce", "reduce_scatter_sharded_adam_allgather"])
def test_frozen_segment_stays_out_of_exchange_and_optimizer_on_every_rank(tmp_path, sharded):
    port = 29500 + (os.getpid() % 2000) + (43 if sharded else 41)
    mp.spawn(_frozen_worker, args=(2, port, str(tmp_path), sharded), nprocs=2, join=True)
    (lo0, hi0), (lo1, hi1) = np.load(tmp_path / "frozen_0.npy"), np.load(tmp_path / "frozen_1.npy")
    if sharded:
        assert lo0 == 12 and hi0 == lo1 and hi1 == 12 + 1001
    else:
        assert (lo0, hi0) == (lo1, hi1) == (12, 12 + 1001)
    # one process, no group: the optimizer covers the whole bucket, frozen group included, like the reference's
    from segs_slam_amd.keyframe_parallel import BucketExchange
    assert BucketExchange(1001, "cpu", None, sharded=False, offset=12).shard_range() == (0, 1013)
