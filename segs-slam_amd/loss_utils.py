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
