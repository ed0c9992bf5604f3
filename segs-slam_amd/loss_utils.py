"""Mirror of include/loss_utils.h:26-124 (L1, PSNR, 11x11 Gaussian-window SSIM) with the same names and arithmetic.

These are the reference's own LibTorch op chains (third-party arithmetic); they stay on
PyTorch-ROCm here (MIOpen depthwise conv).  A fused L1+SSIM HIP kernel is SURVEY section 8(f) n2 ("next").
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:  # loss_utils.h:29-32
    return torch.abs(network_output - gt).mean()


def psnr(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:  # loss_utils.h:39-43
    mse = torch.pow(img1 - img2, 2).mean()
    return 10.0 * torch.log10(1.0 / mse)


def gaussian(window_size: int, sigma: float, device) -> torch.Tensor:  # loss_utils.h:51-64 (INTEGER x = i - 5)
    vals = [math.exp(-float((x - window_size // 2) ** 2) / (2.0 * sigma * sigma)) for x in range(window_size)]
    g = torch.tensor(vals, dtype=torch.float32, device=device)
    return g / g.sum()


def create_window(window_size: int, channel: int, device) -> torch.Tensor:  # loss_utils.h:66-75
    w1 = gaussian(window_size, 1.5, device).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _ssim(img1, img2, window, window_size: int, channel: int, size_average: bool = True):  # loss_utils.h:77-109
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 * 0.01, 0.03 * 0.03
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


_window_cache = {}


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True):  # loss_utils.h:111-124
    """(3,H,W) inputs are treated as unbatched, like the reference.  The window is cached per (device, channels)
    instead of being rebuilt and uploaded on every call (loss_utils.h:56-64 does a small H2D copy per iteration)."""
    channel = img1.size(-3)
    key = (str(img1.device), int(channel), window_size, img1.dtype)
    if key not in _window_cache:
        _window_cache[key] = create_window(window_size, channel, img1.device).to(img1.dtype)
    return _ssim(img1, img2, _window_cache[key], window_size, channel, size_average)


# ---- frequency-domain regularisers of the mapper loss (src/gaussian_mapper.cpp:930-948) -------------------------------
# Tensor-op mirrors of include/loss_utils.h:126-237 (torch.fft runs on the device through hipFFT; autograd gives
# dL/dimage).  Not fused HIP kernels: the FFTs are library calls in the reference too.  Pinned by the reference's own
# functions through tests/golden/loss_reference.npz (high_frequency_loss, low_freq_loss).
def _centre_box(H: int, W: int, cutoff_ratio: float):
    """The two slices of loss_utils.h:139-140 / :180-181.  NB the reference applies them with
    `mask.index_put_({Slice(crow-r, crow+r), Slice(ccol-r, ccol+r)}, v)` to the (3, H, W) spectrum, i.e. to dimensions 0
    (channel) and 1 (rows), not to rows and columns: for any real image size the channel slice is empty, so the
    "high-pass" mask stays all ones and the "low-pass" mask all zeros.  Mirrored as is (checked against the reference's
    compiled code, tests/golden/loss_reference.npz): the drop-in must give the reference's numbers, not the intended ones."""
    crow, ccol = H // 2, W // 2
    r = int(cutoff_ratio * min(H, W) / 2)
    return slice(crow - r, crow + r), slice(ccol - r, ccol + r)


def high_pass_filter(img: torch.Tensor, cutoff_ratio: float) -> torch.Tensor:  # loss_utils.h:126-145
    f = torch.fft.fftshift(torch.fft.fft2(img))          # fftshift over ALL dims, channel included, as the reference does
    mask = torch.ones_like(f)
    rs, cs = _centre_box(img.shape[1], img.shape[2], cutoff_ratio)
    mask[rs, cs] = 0          # dimensions (0, 1), see _centre_box
    return f * mask


def high_frequency_loss(img1: torch.Tensor, img2: torch.Tensor, cutoff_ratio: float = 0.4) -> torch.Tensor:  # :147-165
    return torch.mean(torch.abs(torch.abs(high_pass_filter(img1, cutoff_ratio)) - torch.abs(high_pass_filter(img2, cutoff_ratio))))


def low_pass_filter(img: torch.Tensor, cutoff_ratio: float) -> torch.Tensor:  # :167-186
    f = torch.fft.fftshift(torch.fft.fft2(img))
    mask = torch.zeros_like(f)
    rs, cs = _centre_box(img.shape[1], img.shape[2], cutoff_ratio)
    mask[rs, cs] = 1          # dimensions (0, 1), see _centre_box
    return f * mask


def low_freq_loss(img1: torch.Tensor, img2: torch.Tensor, cutoff_ratio: float = 0.2) -> torch.Tensor:  # :188-213
    norm = float(img1.shape[0] * img1.shape[1] * img1.shape[2])
    a, b = low_pass_filter(img1, cutoff_ratio), low_pass_filter(img2, cutoff_ratio)
    loss_la = torch.sum(torch.abs(torch.abs(a) - torch.abs(b))) / norm
    loss_lp = torch.sum(torch.abs(torch.angle(a) - torch.angle(b))) / norm
    return loss_la + loss_lp


def multi_scale_loss(gen_img: torch.Tensor, target_img: torch.Tensor, scales) -> torch.Tensor:  # :216-237
    loss = torch.zeros((), device=gen_img.device)
    for scale in scales:
        kw = dict(scale_factor=(float(scale), float(scale)), mode="bilinear", align_corners=False, recompute_scale_factor=True)
        g = F.interpolate(gen_img.unsqueeze(0), **kw).squeeze(0)
        t = F.interpolate(target_img.unsqueeze(0), **kw).squeeze(0)
        loss = loss + scale * high_frequency_loss(g, t)
    return loss
