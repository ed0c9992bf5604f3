"""Trainer step around the rasterizer: render -> L1 + lambda*(1-SSIM) -> backward -> [RCCL all-reduce] -> Adam.

Mirror of the loop body of GaussianTrainer::trainingOnce (src/gaussian_trainer.cpp:47-117) -- the minimal
"render + L1/SSIM + Adam" step of SURVEY section 3.2 -- over the rasterizer's own inputs as trainable parameters
(means3D, scales, rotations, opacity, colors = 14 floats per Gaussian, one flat fp32 bucket).

Keyframe-parallel training (new functionality: the reference is single-GPU, F3): one process
per GPU, every rank holds a full replica, rank r renders keyframe r of the step; the only exchange is the one of
keyframe_parallel.BucketExchange over the flat gradient bucket (RCCL over xGMI when the backend is "nccl"):
reduce-scatter -> fused Adam on this rank's shard -> all-gather of the parameters (or, dense, one all-reduce and a full
Adam on every rank); replicas stay bit-identical either way.

The raster backend and the optimizer are injectable so that the distributed logic can be exercised on CPU
(gloo, world_size 2) in tests; the product wiring is `TrainerStep.on_gpu(...)` = HIP engine + fused HIP Adam.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

from . import loss_utils
from .raster_engine import FIELDS, FLOATS_PER_GAUSSIAN, split_flat


@dataclass
class OptimizationParams:
    """The subset of GaussianOptimizationParams (include/gaussian_parameters.h) the step uses."""
    lambda_dssim: float = 0.2           # cfg/gaussian_mapper/RGB-D/Replica/office0.yaml:89
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_max_steps: int = 30000
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    opacity_lr: float = 0.05
    feature_lr: float = 0.0025          # colours
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-15                  # src/gaussian_model.cpp:632-661


def expon_lr(step: int, lr_init: float, lr_final: float, max_steps: int) -> float:
    """getExponLrFunc (src/gaussian_model.cpp:1393-1409) with lr_delay_steps = 0 (delay_rate == 1)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class DeviceStepCount:
    """Step count of one set of torch::optim::Adam parameter groups, kept on the device (segs_adam_step_device,
    include/segs_train.h): two int64 words used in turn, `calls` says which one the next launch reads.  A guarded step that
    the device drops does not advance it, and the host never has to find out."""

    def __init__(self, device):
        # [0], [1]: the pair; [2]: the call count as the hipGraph-capturable form keeps it on the device (segs_adam_step_graph)
        self.words = torch.zeros(3, dtype=torch.int64, device=device)
        self.calls = 0
        self._device_calls_current = True     # words[2] == calls (the eager form does not touch words[2])

    def clone(self) -> "DeviceStepCount":
        c = DeviceStepCount(self.words.device)
        c.words.copy_(self.words)
        c.calls = self.calls
        c._device_calls_current = self._device_calls_current
        return c

    def eager_call(self) -> int:
        """Call index for segs_adam_step_device (host-side parity); the device-side call count falls behind."""
        call = self.calls
        self.calls += 1
        self._device_calls_current = False
        return call

    def sync_device_calls(self):
        """Before capturing / replaying segs_adam_step_graph after eager calls: bring words[2] up to the host's count."""
        if not self._device_calls_current:
            self.words[2:3].fill_(self.calls)
            self._device_calls_current = True

    def value(self) -> int:
        """Steps taken so far (synchronises; for tests and checkpoints)."""
        return int(self.words[self.calls & 1].item())

    def dropped(self) -> int:
        """Launches the device dropped (guard word set: an iteration whose pass overflowed the resident capacity on some
        rank): launches issued minus steps taken (synchronises).  The reference never drops an iteration; the loss returned
        for a dropped one comes from a truncated image and the iteration counter / LR schedule still advanced."""
        return self.calls - self.value()


def field_segments(lrs: Dict[str, float], P: int):
    """(offset, count, lr) of the five per-Gaussian fields inside the flat bucket (raster_engine.FIELDS order)."""
    out, off = [], 0
    for name, n in FIELDS:
        out.append((off, P * n, float(lrs[name])))
        off += P * n
    return out


class FusedAdam:
    """segs_adam_step_device over the flat bucket (include/segs_train.h): one launch, step count on the device, optionally
    guarded by a device word (the all-reduced overflow word of the resident rasterizer) and restricted to a shard."""

    def __init__(self, n_params: int, device, opt: OptimizationParams):
        from . import _capi
        self._capi = _capi
        self._lib = _capi.lib()
        self.opt = opt
        self.exp_avg = torch.zeros(n_params, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n_params, dtype=torch.float32, device=device)
        self.count = DeviceStepCount(device)

    @property
    def step_count(self) -> int:
        return self.count.value()

    def step(self, params_flat: torch.Tensor, grads_flat: torch.Tensor, lrs: Dict[str, float], P: int, grad_scale: float,
             exchange=None, guard: Optional[torch.Tensor] = None):
        groups = field_segments(lrs, P)
        if exchange is not None:
            groups = exchange.clip_segments(groups) or [(0, 0, 0.0)]
        segs = (self._capi.AdamSegment * len(groups))()
        for i, (o, n, lr) in enumerate(groups):
            segs[i].offset, segs[i].count, segs[i].lr = o, n, lr
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        call = self.count.eager_call()
        st = self._lib.segs_adam_step_device(p(params_flat), p(grads_flat), p(self.exp_avg), p(self.exp_avg_sq), segs, len(groups),
                                             self.opt.beta1, self.opt.beta2, self.opt.eps, p(self.count.words), call,
                                             float(grad_scale), 1, p(guard) if guard is not None else None,
                                             C.c_void_p(torch.cuda.current_stream(params_flat.device).cuda_stream))
        self._capi.check(st, "segs_adam_step_device")


class FusedL1SSIM:
    """segs_l1_ssim_loss (include/segs_train.h): loss and dL/dimage in two HBM-streaming kernels."""

    def __init__(self, H: int, W: int, device, lambda_dssim: float):
        from . import _capi
        self._capi, self._lib = _capi, _capi.lib()
        self.H, self.W, self.lam = int(H), int(W), float(lambda_dssim)
        self.temp = torch.empty(self._lib.segs_l1_ssim_temp_bytes(H, W), dtype=torch.uint8, device=device)
        self.out = torch.zeros(3, dtype=torch.float32, device=device)     # loss, l1, ssim
        self.dL = torch.empty((3, H, W), dtype=torch.float32, device=device)

    def __call__(self, image: torch.Tensor, gt: torch.Tensor):
        assert image.is_contiguous() and gt.is_contiguous() and image.shape == (3, self.H, self.W) == gt.shape
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        st = self._lib.segs_l1_ssim_loss(p(image), p(gt), self.H, self.W, self.lam, p(self.out), p(self.dL), p(self.temp),
                                         C.c_void_p(torch.cuda.current_stream(image.device).cuda_stream))
        self._capi.check(st, "segs_l1_ssim_loss")
        return self.out[0], self.dL


class TorchAdam:
    """Same arithmetic with torch ops (LibTorch C++ Adam formula); used for CPU tests of the distributed logic."""

    def __init__(self, n_params: int, device, opt: OptimizationParams):
        self.opt = opt
        self.exp_avg = torch.zeros(n_params, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n_params, dtype=torch.float32, device=device)
        self.step_count = 0

    def step(self, params_flat, grads_flat, lrs, P, grad_scale, exchange=None, guard=None):
        """`exchange`: update only this rank's shard (moments outside it are not kept current, as in FusedAdam)."""
        o = self.opt
        if guard is not None and float(guard.reshape(-1)[0]) != 0:
            grads_flat.zero_()
            return
        self.step_count += 1
        bc1 = 1.0 - o.beta1 ** self.step_count
        bc2 = 1.0 - o.beta2 ** self.step_count
        groups = field_segments(lrs, P)
        if exchange is not None:
            groups = exchange.clip_segments(groups)
        for off, cnt, lr in groups:
            sl = slice(off, off + cnt)
            g = grads_flat[sl] * grad_scale
            self.exp_avg[sl].mul_(o.beta1).add_(g, alpha=1 - o.beta1)
            self.exp_avg_sq[sl].mul_(o.beta2).addcmul_(g, g, value=1 - o.beta2)
            denom = (self.exp_avg_sq[sl].sqrt() / math.sqrt(bc2)).add_(o.eps)
            params_flat[sl].addcdiv_(self.exp_avg[sl], denom, value=-(lr / bc1))
        grads_flat.zero_()


class TrainerStep:
    """One keyframe-parallel training step.  `render_backward(params: dict, keyframe, dL_fn) -> (image, loss)` must
    render the keyframe, call dL_fn(image) -> (loss, dL_dimage) and leave parameter gradients in `grads_flat`."""

    def __init__(self, params_flat: torch.Tensor, P: int, render_backward: Callable, optimizer, opt: OptimizationParams,
                 grads_flat: torch.Tensor, process_group=None, sharded_optimizer: bool = True,
                 single_rank_collectives: bool = False):
        import inspect
        from .keyframe_parallel import BucketExchange
        assert params_flat.numel() == FLOATS_PER_GAUSSIAN * P and grads_flat.numel() == params_flat.numel()
        self.params_flat, self.grads_flat, self.P = params_flat, grads_flat, P
        self.params = split_flat(params_flat, P)
        self.render_backward, self.optimizer, self.opt = render_backward, optimizer, opt
        self.pg = process_group
        self.exchange = BucketExchange(params_flat.numel(), params_flat.device, process_group, sharded=sharded_optimizer,
                                       single_rank_collectives=single_rank_collectives, grads=grads_flat)
        self.world, self.rank = self.exchange.world, self.exchange.rank
        self.iteration = 0
        # a backend that knows an overflow word (the HIP engine) takes a hook it calls right after its forward
        self._backend_takes_hook = render_backward is not None and "after_forward" in inspect.signature(render_backward).parameters

    def learning_rates(self, iteration: int) -> Dict[str, float]:
        o = self.opt  # updateLearningRate (src/gaussian_model.cpp:874-998): only the position group is scheduled
        return {"means3D": expon_lr(iteration, o.position_lr_init, o.position_lr_final, o.position_lr_max_steps),
                "scales": o.scaling_lr, "rotations": o.rotation_lr, "opacity": o.opacity_lr, "colors": o.feature_lr}

    fused_loss = None  # set by on_gpu(): FusedL1SSIM

    def loss_and_grad(self, image: torch.Tensor, gt: torch.Tensor):
        """Ll1 and (1-lambda) Ll1 + lambda (1 - SSIM)  (src/gaussian_trainer.cpp:89-90) and its gradient wrt image."""
        if self.fused_loss is not None:
            return self.fused_loss(image, gt)
        img = image.detach().requires_grad_(True)
        Ll1 = loss_utils.l1_loss(img, gt)
        loss = (1.0 - self.opt.lambda_dssim) * Ll1 + self.opt.lambda_dssim * (1.0 - loss_utils.ssim(img, gt))
        (dL,) = torch.autograd.grad(loss, img)
        return loss.detach(), dL.contiguous()

    def _exchange_sharded(self) -> bool:
        return self.exchange.sharded

    def keyframe_for(self, step: int, n_keyframes: int) -> int:
        """Deterministic shared schedule: rank r takes keyframe (step * world + r) mod n."""
        return (step * self.world + self.rank) % n_keyframes

    def training_once(self, keyframes, gt_images) -> torch.Tensor:
        """Nothing here waits for the device: a pass whose instance count outgrew some rank's resident capacity is dropped by
        every rank on the device (the optimizer is guarded by the all-reduced overflow word and its step count lives there);
        the rank concerned re-sizes its scratch at its next forward.  The dropped iteration is then run again by every rank before
        the next one (`redo_dropped_steps`), so no optimizer step of the reference's sequence is lost
        (src/gaussian_mapper.cpp:1027-1030 never skips one).  CONTRACT: the keyframe's and the target's tensors of a call must
        stay unchanged until the next call (or finish()) has returned -- a redo trains on them again."""
        # An iteration the device dropped is run again -- same keyframe, same iteration number -- as soon as the host resolves that
        # step's overflow word, which is before the next iteration is queued (ScaffoldTrainerStep.training_once).  One rank: the
        # engine's own status word.  N > 1: the SUMMED word every rank mirrored to its host after the gradient exchange
        # (BucketExchange.mirror_flag), so all ranks redo the same iteration together and replicas stay bit-identical.
        self._redo_if_dropped()
        self.iteration += 1
        k = self.keyframe_for(self.iteration - 1, len(keyframes))
        self._last_iteration = (keyframes[k], gt_images[k], self.iteration)
        return self._iteration_body(keyframes[k], gt_images[k], self.iteration)

    def _redo_if_dropped(self):
        prev = getattr(self, "_last_iteration", None)
        if prev is None or not getattr(self, "redo_dropped_steps", True) or getattr(self, "use_graph", False):
            return
        eng = getattr(self, "engine", None)
        for _ in range(4):
            if self.world == 1:
                dropped = eng is not None and eng.resident and not eng.check(raise_on_overflow=False)
            else:
                dropped = bool(self.exchange.step_dropped())
                if eng is not None and eng.resident:
                    eng.check(raise_on_overflow=False)      # the rank that overflowed re-calibrates in its next forward
            if not dropped:
                break
            self.redone_steps = getattr(self, "redone_steps", 0) + 1
            self._iteration_body(*prev)
            if self.world == 1:
                break                                       # (a re-calibrating forward cannot overflow)
        else:
            raise RuntimeError("an iteration kept being dropped by the device")
        self._last_iteration = None

    def finish(self):
        """Resolve the LAST iteration's status word and run that iteration again if the device dropped it (training_once only
        learns of a drop at the next call).  Call once after the last training_once of a run."""
        self._redo_if_dropped()

    def _iteration_body(self, keyframe, gt, it: int) -> torch.Tensor:
        lrs = self.learning_rates(it)
        ex = self.exchange
        if getattr(self, "use_graph", False) and not ex.active:
            loss = self._training_once_graph(keyframe, gt, lrs)
            if loss is not None:
                return loss
        dL_fn = lambda im: self.loss_and_grad(im, gt)  # noqa: E731
        if self._backend_takes_hook:
            loss = self.render_backward(self.params, keyframe, dL_fn, after_forward=ex.reduce_flag_async)
        else:
            loss = self.render_backward(self.params, keyframe, dL_fn)
            ex.reduce_flag_async(None)
        flag = ex.wait_flag()
        ex.reduce_gradients(self.grads_flat)  # sum over the keyframes of this step
        if ex.active and getattr(self, "redo_dropped_steps", True):
            ex.mirror_flag()
        self.optimizer.step(self.params_flat, self.grads_flat, lrs, self.P, 1.0 / self.world, exchange=ex, guard=flag)
        ex.gather(self.params_flat)
        return loss

    # ---- whole-iteration hipGraph (the on_gpu wiring: HIP raster engine + fused loss + fused Adam) -------------------------
    def enable_graph(self, on: bool = True):
        """Replay the iteration from a captured hipGraph (single rank, calibrated resident rasterizer): the keyframe matrices,
        the target image and the learning rates go through staging buffers refreshed before each replay; same kernels, same
        arguments, same order as the eager path."""
        assert getattr(self, "engine", None) is not None and isinstance(self.optimizer, FusedAdam), "graph mode needs TrainerStep.on_gpu()"
        self.use_graph, self._graphs, self._stage, self.graph_replays = bool(on), {}, None, 0

    def _training_once_graph(self, keyframe, gt, lrs):
        from . import _capi
        eng, opt = self.engine, self.optimizer
        if not eng.resident or eng.capacity <= 0 or not eng.poll() or eng.capacity <= 0:
            return None
        view, proj, campos, tanx, tany = keyframe
        dev = self.params_flat.device
        if self._stage is None:
            self._stage = dict(packed=torch.zeros(35, dtype=torch.float32, device=dev), gt=torch.empty_like(gt),
                               lr=torch.zeros(16, dtype=torch.float64, device=dev))
        st = self._stage
        groups = field_segments(lrs, self.P)
        key = (eng.capacity, eng._bin_r.data_ptr(), float(tanx), float(tany), tuple(gt.shape))
        pk = st["packed"]
        pk[0:16].copy_(view.reshape(-1)); pk[16:32].copy_(proj.reshape(-1)); pk[32:35].copy_(campos.reshape(-1))
        st["gt"].copy_(gt)
        vals = (C.c_double * len(groups))(*[float(g[2]) for g in groups])
        stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)  # noqa: E731
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        _capi.check(_capi.lib().segs_set_doubles(p(st["lr"]), vals, len(groups), stream()), "segs_set_doubles")
        opt.count.sync_device_calls()
        g = self._graphs.get(key)
        if g is None:
            skf = (pk[0:16].view(4, 4), pk[16:32].view(4, 4), pk[32:35], tanx, tany)
            guard = eng._status[3:4]
            segs = (_capi.AdamSegment * len(groups))()
            for i, (o, n, _) in enumerate(groups):
                segs[i].offset, segs[i].count, segs[i].lr = o, n, 0.0

            def body():
                self.render_backward(self.params, skf, lambda im: self.loss_and_grad(im, st["gt"]))
                rc = _capi.lib().segs_adam_step_graph(p(self.params_flat), p(self.grads_flat), p(opt.exp_avg), p(opt.exp_avg_sq), segs,
                                                      len(groups), p(st["lr"]), opt.opt.beta1, opt.opt.beta2, opt.opt.eps, p(opt.count.words),
                                                      1.0, 1, p(guard), stream())
                _capi.check(rc, "segs_adam_step_graph")

            eng.check(raise_on_overflow=False)
            if eng.capacity <= 0:
                return None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            self._graphs = {key: g}
        g.replay()
        opt.count.calls += 1
        eng.after_graph_replay()
        self.graph_replays += 1
        return self.fused_loss.out[0]

    # ---- product wiring -------------------------------------------------------------------------------
    @staticmethod
    def on_gpu(scene, device, opt: Optional[OptimizationParams] = None, process_group=None, sharded_optimizer: bool = True,
               single_rank_collectives: bool = False):
        """HIP raster engine + fused HIP Adam over a segs_slam_amd.scenes.Scene's Gaussians."""
        import numpy as np
        from .raster_engine import RasterEngine
        opt = opt or OptimizationParams()
        P = scene.P
        cam = scene.camera
        eng = RasterEngine(P, cam.width, cam.height, device, resident=True)
        params_flat = torch.empty(FLOATS_PER_GAUSSIAN * P, dtype=torch.float32, device=device)
        views = split_flat(params_flat, P)
        for name, arr in (("means3D", scene.means3D), ("scales", scene.scales), ("rotations", scene.rotations),
                          ("opacity", scene.opacity), ("colors", scene.colors)):
            views[name].copy_(torch.from_numpy(np.ascontiguousarray(arr)))
        bg = torch.from_numpy(scene.bg).to(device)

        def render_backward(params, keyframe, dL_fn, after_forward=None):
            view, proj, campos, tanx, tany = keyframe
            # (a forward that finds the previous pass overflowed goes through the synchronising path and re-sizes the scratch)
            image = eng.forward(bg, params["means3D"], params["colors"], params["opacity"], params["scales"],
                                params["rotations"], view, proj, campos, tanx, tany)
            if after_forward is not None:
                status = getattr(eng, "_status", None)
                after_forward(status[3:4] if (status is not None and eng._last_resident) else None)
            loss, dL = dL_fn(image)
            eng.backward(dL)
            return loss

        step = TrainerStep(params_flat, P, render_backward, FusedAdam(params_flat.numel(), device, opt), opt, eng.grads_flat,
                           process_group, sharded_optimizer=sharded_optimizer, single_rank_collectives=single_rank_collectives)
        step.engine = eng
        step.fused_loss = FusedL1SSIM(cam.height, cam.width, device, opt.lambda_dssim)
        return step


def keyframe_tensors(cam, device):
    """(view, proj, campos, tanfovx, tanfovy) of a scenes.Camera on `device`."""
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return (t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center), cam.tanfovx, cam.tanfovy)
