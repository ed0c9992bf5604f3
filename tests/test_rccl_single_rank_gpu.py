"""The collectives of the sharded exchange under the REAL backend ("nccl" = RCCL) on a one-GPU box: a process group of ONE
rank with `single_rank_collectives=True` runs reduce_scatter_tensor -> fused Adam on the shard -> all_gather_into_tensor
(keyframe_parallel.BucketExchange), which every N > 1 GPU test before only emulated under gloo.  With one rank the sums are
identities, so the parameters must equal the dense / no-exchange path BIT FOR BIT on the same gradients -- for a bucket whose
length needs the padded staging buffers and for one that does not."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from segs_slam_amd.gaussian_trainer import FusedAdam, OptimizationParams
    from segs_slam_amd.keyframe_parallel import BucketExchange
    from segs_slam_amd.raster_engine import FLOATS_PER_GAUSSIAN
    out = {}
    for P in (1000, 1001):                      # 14 * 1001 = 14 014 is not a multiple of 4: padded staging buffers
        n = FLOATS_PER_GAUSSIAN * P
        g = torch.Generator(device="cpu").manual_seed(5 + P)
        p0 = torch.randn(n, generator=g).to(dev)
        grads = [torch.randn(n, generator=g).to(dev) * 1e-3 for _ in range(3)]
        lrs = {"means3D": 1.6e-4, "scales": 5e-3, "rotations": 1e-3, "opacity": 5e-2, "colors": 2.5e-3}
        res = {}
        for mode in ("plain", "dense", "dense_word_in_bucket", "sharded"):
            params = p0.clone()
            opt = OptimizationParams()
            adam = FusedAdam(n, dev, opt)
            ex = None
            gb = torch.zeros(n + 4, device=dev)[:n]           # a bucket with the spare tail the engines allocate
            if mode != "plain":
                ex = BucketExchange(n, dev, None, sharded=(mode == "sharded"), single_rank_collectives=True,
                                    grads=gb if mode == "dense_word_in_bucket" else None)
                assert ex.active and ex.sharded == (mode == "sharded") and not ex._emulate
                assert (ex._ext is not None) == (mode == "dense_word_in_bucket")
            for gi in grads:
                gb.copy_(gi)
                if ex is not None:
                    ex.reduce_flag_async(torch.zeros(1, dtype=torch.int32, device=dev))
                    flag = ex.wait_flag()
                    ex.reduce_gradients(gb)
                    adam.step(params, gb, lrs, P, 1.0, exchange=ex, guard=flag)
                    ex.gather(params)
                else:
                    adam.step(params, gb, lrs, P, 1.0)
            torch.cuda.synchronize()
            res[mode] = params.cpu().numpy()
            assert adam.step_count == 3
        out[f"plain_{P}"], out[f"dense_{P}"], out[f"sharded_{P}"] = res["plain"], res["dense"], res["sharded"]
        out[f"piggy_{P}"] = res["dense_word_in_bucket"]
    # a raised overflow word riding in the bucket drops the step on the device
    n = FLOATS_PER_GAUSSIAN * 1000
    gb = torch.zeros(n + 4, device=dev)[:n]
    ex = BucketExchange(n, dev, None, sharded=False, single_rank_collectives=True, grads=gb)
    params = torch.ones(n, device=dev)
    adam = FusedAdam(n, dev, OptimizationParams())
    gb.fill_(1e-3)
    ex.reduce_flag_async(torch.ones(1, dtype=torch.int32, device=dev))
    flag = ex.wait_flag()
    ex.reduce_gradients(gb)
    adam.step(params, gb, lrs, 1000, 1.0, exchange=ex, guard=flag)
    torch.cuda.synchronize()
    out["dropped"] = np.array([float(flag.reshape(-1)[0]), float((params != 1).sum()), adam.step_count])
    # the whole trainer step once through the same collectives (no crash, finite parameters, sharded exchange in use)
    from segs_slam_amd import scenes
    from segs_slam_amd.gaussian_trainer import TrainerStep, keyframe_tensors
    sc = scenes.make_scene(4000, 160, 96, 150.0, 150.0, seed=11)
    sc.scales *= 3.0
    step = TrainerStep.on_gpu(sc, dev, single_rank_collectives=True)
    assert step.exchange.sharded
    kf = [keyframe_tensors(sc.camera, dev)]
    gt = [torch.rand(3, 96, 160, device=dev)]
    for _ in range(3):
        loss = step.training_once(kf, gt)
    torch.cuda.synchronize()
    out["trainer_finite"] = np.array([bool(torch.isfinite(step.params_flat).all()), bool(torch.isfinite(loss))])
    out["trainer_steps"] = np.array([step.optimizer.step_count])
    np.savez(os.path.join(outdir, "out.npz"), **out)
    dist.destroy_process_group()


def test_sharded_exchange_under_rccl_with_one_rank_equals_dense_path():
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(1, 29611 + os.getpid() % 300, d), nprocs=1, join=True)
        r = dict(np.load(os.path.join(d, "out.npz")))
    for P in (1000, 1001):
        assert np.array_equal(r[f"plain_{P}"], r[f"dense_{P}"]), P
        assert np.array_equal(r[f"plain_{P}"], r[f"sharded_{P}"]), P
        assert np.array_equal(r[f"plain_{P}"], r[f"piggy_{P}"]), P
    assert list(r["dropped"]) == [1.0, 0.0, 0.0]
    assert r["trainer_finite"].all() and int(r["trainer_steps"][0]) == 3
