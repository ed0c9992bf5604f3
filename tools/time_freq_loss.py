#!/usr/bin/env python3
"""The frequency regulariser alone (segs_freq_loss through frequency_loss.FusedFrequencyLoss) at one image size: ms per call by HIP
events.  usage (GPU box): python tools/time_freq_loss.py [H W [calls]]   (SEGS_FREQ_HIPFFT=1: the vendor library's transforms)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from segs_slam_amd.frequency_loss import FusedFrequencyLoss  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (680, 1200)
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
gt = torch.rand(3, H, W, generator=g).to(dev)
img = (gt + 0.1 * torch.randn(3, H, W, generator=g).to(dev)).clamp(0, 1).contiguous()
fl = FusedFrequencyLoss(H, W, dev, lambda_high=0.01)
dL = torch.zeros_like(img)
word = torch.zeros(1, device=dev)
for _ in range(10):
    fl(img, gt, dL, word)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(calls):
    fl(img, gt, dL, word)
b.record()
torch.cuda.synchronize()
print(f"{W}x{H} folded={fl.folded} {'hipFFT' if os.environ.get('SEGS_FREQ_HIPFFT') == '1' else 'own transforms'}: {a.elapsed_time(b) / calls * 1e3:.1f} us per call")
