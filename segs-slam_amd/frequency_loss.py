"""The mapper's frequency regulariser (src/gaussian_mapper.cpp:930-945) as a device path: loss value and dL/dimage.

    loss += lambda_high * multi_scale_loss(image, gt, scales)          (Mapper.use_multi_resolution, include/loss_utils.h:216-237)
    loss += lambda_high * high_frequency_loss(image, gt)               (otherwise, :147-165)
    loss += lambda_low  * low_freq_loss(image, gt)                     (:188-213; lambda_low = 0 in every shipped cfg)

The reference evaluates these with LibTorch op chains + autograd (per scale: interpolate x2, fft2 x2, fftshift x2, mask x2,
abs x3, sub, mean and the backward of each).  Here (include/segs_train.h, csrc/freq_loss.hip): |FFT(gt)| is constant per
keyframe and scale and cached; the backward of mean(| |F g| - |F t| |) is ONE unnormalised inverse real FFT of
sign(.) F g / |F g|; and for image sizes that are multiples of 4 the half- and quarter-size spectra are alias folds of the
full-size one, so the whole three-scale regulariser is one forward transform, one kernel, one inverse transform.  The FFTs
themselves stay library calls (hipFFT), as in the reference.

The gradient is DISCONTINUOUS in the image wherever a spectrum magnitude crosses its target's (sign(|G_k| - |T_k|) flips): two
float32 evaluations of the same formula may legitimately disagree by 2 w_l on such a frequency.  The tests therefore compare
dL/dimage against a float64 evaluation in which only frequencies whose float64 margin | |G_k| - |T_k| | / |T_k| is below 1e-5
may take the device's sign (tests/test_frequency_loss_gpu.py).

low_freq_loss: the reference's mask quirk multiplies both spectra by an all-zero mask, so its gradient is
identically zero (tests/golden/loss_reference.npz: dL_low == 0 from the reference's own compiled code) while its VALUE is a
sign-pattern artefact of angle(-0 + 0i) = pi.  With lambda_low != 0 the value is therefore evaluated with the tensor mirror
(loss_utils.low_freq_loss, no gradient) so the reported loss still matches; with lambda_low == 0 nothing is computed.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import List, Optional, Sequence, Tuple

import torch

from . import _capi


def level_sizes(H: int, W: int, scales: Sequence[float]) -> List[Tuple[int, int]]:
    """Output sizes of F.interpolate(scale_factor=s, recompute_scale_factor=True): floor(size * s) in double."""
    return [(int(math.floor(float(H) * float(s))), int(math.floor(float(W) * float(s)))) for s in scales]


class FusedFrequencyLoss:
    """One object per image size.  `__call__(image, gt, dL_inout, loss_inout)` adds the regulariser's gradient to
    `dL_inout` (3,H,W) and its value to the device scalar `loss_inout`, and returns the value (a device scalar view).

    Default: the plan entry points of include/segs_train.h (segs_freq_plan_create / segs_freq_target / segs_freq_loss) --
    hipFFT driven from inside the library, ONE transform pair per step for the shipped sizes (alias folding).
    `torch_fft=True`: the piecewise entry points around torch.fft.rfft2 / irfft2 (one pair per scale), the form a host
    that brings its own FFT uses; kept as the A/B partner of the plan path."""

    def __init__(self, H: int, W: int, device, lambda_high: float = 0.01, scales: Sequence[float] = (1.0, 0.5, 0.25),
                 multi_resolution: bool = True, lambda_low: float = 0.0, max_cached_targets: int = 1024, torch_fft: bool = False):
        self._lib = _capi.lib()
        self.H, self.W, self.dev = int(H), int(W), torch.device(device)
        self.lambda_high, self.lambda_low = float(lambda_high), float(lambda_low)
        # high_frequency_loss alone is multi_scale_loss's s = 1 term with weight 1 (loss_utils.h:235 multiplies by `scale`)
        self.scales = tuple(float(s) for s in scales) if multi_resolution else (1.0,)
        self.sizes = level_sizes(self.H, self.W, self.scales)
        n = len(self.sizes)
        if n > 4 or any(h <= 0 or w <= 0 for h, w in self.sizes):
            raise ValueError(f"unsupported scales {self.scales} for {self.W}x{self.H}")
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.value = torch.zeros(1, **f32)
        self._targets: "OrderedDict[tuple, list]" = OrderedDict()
        self._max_cached = int(max_cached_targets)
        self.torch_fft = bool(torch_fft)
        self._plan = None
        if not self.torch_fft:
            plan = C.c_void_p()
            sc = (C.c_float * n)(*self.scales)
            with torch.cuda.device(self.dev):
                _capi.check(self._lib.segs_freq_plan_create(self.H, self.W, n, sc, self.lambda_high, C.byref(plan)),
                            "segs_freq_plan_create")
            self._plan = plan
            hs, ws, folded = (C.c_int * n)(), (C.c_int * n)(), C.c_int()
            assert self._lib.segs_freq_plan_levels(plan, hs, ws, C.byref(folded)) == n
            assert [(hs[i], ws[i]) for i in range(n)] == self.sizes
            self.folded = bool(folded.value)
            self._target_floats = int(self._lib.segs_freq_target_floats(plan))
            return
        self.folded = False
        self._h = (C.c_int * n)(*[h for h, _ in self.sizes])
        self._w = (C.c_int * n)(*[w for _, w in self.sizes])
        self._weight = (C.c_float * n)(*[self.lambda_high * s / float(3 * h * w) for s, (h, w) in zip(self.scales, self.sizes)])
        full = lambda h, w: (h, w) == (self.H, self.W)  # noqa: E731
        self._level = [None if full(h, w) else torch.empty((3, h, w), **f32) for h, w in self.sizes]
        self._spec = [torch.empty((3, h, w // 2 + 1), dtype=torch.complex64, device=self.dev) for h, w in self.sizes]
        self._grad = [torch.empty((3, h, w), **f32) for h, w in self.sizes]
        self._temp = torch.empty(self._lib.segs_freq_temp_bytes(3, n, self._h, self._w), dtype=torch.uint8, device=self.dev)
        self._level_ptrs, self._spec_ptrs, self._grad_ptrs = self._ptrs(self._level), self._ptrs(self._spec), self._ptrs(self._grad)

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan is not None and _capi is not None:
            try:
                self._lib.segs_freq_plan_destroy(plan)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    @staticmethod
    def _ptrs(ts):
        return (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _pyramid(self, image: torch.Tensor):
        st = self._lib.segs_freq_pyramid(C.c_void_p(image.data_ptr()), 3, self.H, self.W, len(self.sizes), self._h, self._w,
                                         self._level_ptrs, self._stream())
        _capi.check(st, "segs_freq_pyramid")
        return [image if lv is None else lv for lv in self._level]

    def target_magnitudes(self, gt: torch.Tensor) -> list:
        """|FFT(resize_s(gt))| per scale, cached per target tensor (a keyframe's image does not change)."""
        key = (gt.data_ptr(), gt._version)
        hit = self._targets.get(key)
        if hit is not None:
            self._targets.move_to_end(key)
            return hit
        if self._plan is not None:
            block = torch.empty(self._target_floats, dtype=torch.float32, device=self.dev)
            _capi.check(self._lib.segs_freq_target(self._plan, C.c_void_p(gt.data_ptr()), C.c_void_p(block.data_ptr()),
                                                   self._stream()), "segs_freq_target")
            hit = [block, None, gt]                     # gt kept alive: the key is its address
        else:
            mags = []
            for lv in self._pyramid(gt):
                spec = torch.fft.rfft2(lv)
                mag = torch.empty(spec.shape, dtype=torch.float32, device=self.dev)
                st = self._lib.segs_spectrum_magnitude(C.c_void_p(spec.data_ptr()), spec.numel(), C.c_void_p(mag.data_ptr()),
                                                       self._stream())
                _capi.check(st, "segs_spectrum_magnitude")
                mags.append(mag)
            hit = [mags, self._ptrs(mags), gt]
        self._targets[key] = hit
        while len(self._targets) > self._max_cached:
            self._targets.popitem(last=False)
        return hit

    def target_block(self, gt: torch.Tensor) -> torch.Tensor:
        """The flat |FFT| table block of `gt` (plan path): what apply() reads."""
        assert self._plan is not None
        return self.target_magnitudes(gt)[0]

    def apply(self, image: torch.Tensor, target_block: torch.Tensor, dL_inout: torch.Tensor, loss_inout: Optional[torch.Tensor] = None):
        """__call__ with the target tables handed in (plan path, lambda_low = 0): a fixed launch sequence over fixed addresses,
        i.e. what a captured iteration replays; the caller copies the keyframe's block into its staging buffer beforehand."""
        assert self._plan is not None and self.lambda_low == 0.0 and target_block.numel() == self._target_floats
        st = self._lib.segs_freq_loss(self._plan, C.c_void_p(image.data_ptr()), C.c_void_p(target_block.data_ptr()),
                                      C.c_void_p(dL_inout.data_ptr()), C.c_void_p(self.value.data_ptr()),
                                      None if loss_inout is None else C.c_void_p(loss_inout.data_ptr()), self._stream())
        _capi.check(st, "segs_freq_loss")
        return self.value[0]

    def __call__(self, image: torch.Tensor, gt: torch.Tensor, dL_inout: torch.Tensor, loss_inout: Optional[torch.Tensor] = None):
        assert image.is_cuda and image.is_contiguous() and gt.is_contiguous() and dL_inout.is_contiguous()
        assert image.shape == (3, self.H, self.W) == gt.shape == dL_inout.shape and image.dtype == torch.float32
        tgt, tptrs, _ = self.target_magnitudes(gt)
        loss_p = None if loss_inout is None else C.c_void_p(loss_inout.data_ptr())
        if self._plan is not None:
            st = self._lib.segs_freq_loss(self._plan, C.c_void_p(image.data_ptr()), C.c_void_p(tgt.data_ptr()),
                                          C.c_void_p(dL_inout.data_ptr()), C.c_void_p(self.value.data_ptr()), loss_p, self._stream())
            _capi.check(st, "segs_freq_loss")
        else:
            n = len(self.sizes)
            for lv, spec in zip(self._pyramid(image), self._spec):
                torch.fft.rfft2(lv, out=spec)
            st = self._lib.segs_freq_spectrum_loss(3, n, self._h, self._w, self._spec_ptrs, tptrs, self._weight,
                                                   C.c_void_p(self.value.data_ptr()), loss_p, C.c_void_p(self._temp.data_ptr()),
                                                   self._stream())
            _capi.check(st, "segs_freq_spectrum_loss")
            for spec, g, (h, w) in zip(self._spec, self._grad, self.sizes):
                torch.fft.irfft2(spec, s=(h, w), norm="forward", out=g)
            st = self._lib.segs_freq_pyramid_backward_add(C.c_void_p(dL_inout.data_ptr()), 3, self.H, self.W, n, self._h, self._w,
                                                          self._grad_ptrs, self._stream())
            _capi.check(st, "segs_freq_pyramid_backward_add")
        if self.lambda_low != 0.0:
            from . import loss_utils
            with torch.no_grad():
                low = self.lambda_low * loss_utils.low_freq_loss(image, gt)
                self.value += low
                if loss_inout is not None:
                    loss_inout += low
        return self.value[0]
