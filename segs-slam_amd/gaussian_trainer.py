"""Trainer step around the rasterizer: render -> L1 + lambda*(1-SSIM) -> backward -> [RCCL all-reduce] -> Adam.

Mirror of the loop body of GaussianTrainer::trainingOnce (src/gaussian_trainer.cpp:47-117) -- the minimal
"render + L1/SSIM + Adam" step of SURVEY section 3.2 -- over the rasterizer's own inputs as trainable parameters
(means3D, scales, rotations, opacity, colors = 14 floats per Gaussian, one flat fp32 bucket).

Keyframe-parallel training (new functionality: the reference is single-GPU, F3): one process
per GPU, every rank holds a full replica, rank r renders keyframe r of the step; the only exchange is ONE
all-reduce(sum) of the flat gradient bucket (RCCL over xGMI when the backend is "nccl"); the fused Adam then
applies the same averaged gradient on every rank, so replicas stay bit-identical.

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


class FusedAdam:
    """segs_adam_step over the flat bucket (include/segs_train.h)."""

    def __init__(self, n_params: int, device, opt: OptimizationParams):
        from . import _capi
        self._capi = _capi
        self._lib = _capi.lib()
        self.opt = opt
        self.exp_avg = torch.zeros(n_params, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n_params, dtype=torch.float32, device=device)
        self.step_count = 0

    def step(self, params_flat: torch.Tensor, grads_flat: torch.Tensor, lrs: Dict[str, float], P: int, grad_scale: float):
        self.step_count += 1
        segs = (self._capi.AdamSegment * len(FIELDS))()
        off = 0
        for i, (name, n) in enumerate(FIELDS):
            segs[i].offset, segs[i].count, segs[i].lr = off, P * n, float(lrs[name])
            off += P * n
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        st = self._lib.segs_adam_step(p(params_flat), p(grads_flat), p(self.exp_avg), p(self.exp_avg_sq), segs, len(FIELDS),
                                      self.opt.beta1, self.opt.beta2, self.opt.eps, self.step_count, float(grad_scale), 1,
                                      C.c_void_p(torch.cuda.current_stream(params_flat.device).cuda_stream))
        self._capi.check(st, "segs_adam_step")


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

    def step(self, params_flat, grads_flat, lrs, P, grad_scale):
        o = self.opt
        self.step_count += 1
        bc1 = 1.0 - o.beta1 ** self.step_count
        bc2 = 1.0 - o.beta2 ** self.step_count
        g = grads_flat * grad_scale
        self.exp_avg.mul_(o.beta1).add_(g, alpha=1 - o.beta1)
        self.exp_avg_sq.mul_(o.beta2).addcmul_(g, g, value=1 - o.beta2)
        denom = (self.exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(o.eps)
        off = 0
        for name, n in FIELDS:
            sl = slice(off, off + P * n)
            params_flat[sl].addcdiv_(self.exp_avg[sl], denom[sl], value=-(lrs[name] / bc1))
            off += P * n
        grads_flat.zero_()


class TrainerStep:
    """One keyframe-parallel training step.  `render_backward(params: dict, keyframe, dL_fn) -> (image, loss)` must
    render the keyframe, call dL_fn(image) -> (loss, dL_dimage) and leave parameter gradients in `grads_flat`."""

    def __init__(self, params_flat: torch.Tensor, P: int, render_backward: Callable, optimizer, opt: OptimizationParams,
                 grads_flat: torch.Tensor, process_group=None):
        assert params_flat.numel() == FLOATS_PER_GAUSSIAN * P and grads_flat.numel() == params_flat.numel()
        self.params_flat, self.grads_flat, self.P = params_flat, grads_flat, P
        self.params = split_flat(params_flat, P)
        self.render_backward, self.optimizer, self.opt = render_backward, optimizer, opt
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.iteration = 0

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

    def keyframe_for(self, step: int, n_keyframes: int) -> int:
        """Deterministic shared schedule: rank r takes keyframe (step * world + r) mod n."""
        return (step * self.world + self.rank) % n_keyframes

    def training_once(self, keyframes, gt_images) -> torch.Tensor:
        self.iteration += 1
        lrs = self.learning_rates(self.iteration)
        k = self.keyframe_for(self.iteration - 1, len(keyframes))
        loss = self.render_backward(self.params, keyframes[k], lambda im: self.loss_and_grad(im, gt_images[k]))
        if self.world > 1:
            dist.all_reduce(self.grads_flat, group=self.pg)  # sum over keyframes of this step
        self.optimizer.step(self.params_flat, self.grads_flat, lrs, self.P, 1.0 / self.world)
        return loss

    # ---- product wiring -------------------------------------------------------------------------------
    @staticmethod
    def on_gpu(scene, device, opt: Optional[OptimizationParams] = None, process_group=None):
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

        def render_backward(params, keyframe, dL_fn):
            view, proj, campos, tanx, tany = keyframe
            image = eng.forward(bg, params["means3D"], params["colors"], params["opacity"], params["scales"],
                                params["rotations"], view, proj, campos, tanx, tany)
            loss, dL = dL_fn(image)
            eng.backward(dL)
            if not eng.check(raise_on_overflow=False):
                # the instance count outgrew the resident capacity (the map changed): redo this keyframe through the
                # synchronising path, which re-sizes the scratch; gradients are fully overwritten by the second pass
                image = eng.forward(bg, params["means3D"], params["colors"], params["opacity"], params["scales"],
                                    params["rotations"], view, proj, campos, tanx, tany)
                loss, dL = dL_fn(image)
                eng.backward(dL)
            return loss

        step = TrainerStep(params_flat, P, render_backward, FusedAdam(params_flat.numel(), device, opt), opt, eng.grads_flat,
                           process_group)
        step.engine = eng
        step.fused_loss = FusedL1SSIM(cam.height, cam.width, device, opt.lambda_dssim)
        return step


def keyframe_tensors(cam, device):
    """(view, proj, campos, tanfovx, tanfovy) of a scenes.Camera on `device`."""
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return (t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center), cam.tanfovx, cam.tanfovy)
