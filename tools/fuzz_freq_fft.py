#!/usr/bin/env python3
"""Randomised check of the frequency regulariser's own transforms (csrc/real_fft.h) over image sizes: every folded size H, W
(multiples of 4) whose factors are in {2, 3, 5, 17} drawn from a range, against a float64 evaluation of the reference's formula by
the rule of tests/test_frequency_loss_gpu._check_against_float64 (value 1e-5, gradient 1e-4 + 1e-5 of the largest entry, near ties
arbitrated).  usage (GPU box): python tools/fuzz_freq_fft.py [n_cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from segs_slam_amd.frequency_loss import FusedFrequencyLoss  # noqa: E402
from tests.test_frequency_loss_gpu import _check_against_float64, _images  # noqa: E402


def smooth(n):
    for p in (2, 3, 5, 17):
        while n % p == 0:
            n //= p
    return n == 1


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 97000)
    hs = [h for h in range(16, 1100, 4) if smooth(h)]
    ws = [w for w in range(16, 2000, 4) if smooth(w // 2) and smooth(w)]
    dev = torch.device("cuda:0")
    fails = 0
    for case in range(n_cases):
        H, W = int(rng.choice(hs)), int(rng.choice(ws))
        if H * W > 1300 * 700:
            H = int(rng.choice([h for h in hs if h * W <= 1300 * 700]))
        img, gt = _images(H, W, dev, 1000 + case)
        fl = FusedFrequencyLoss(H, W, dev, lambda_high=0.01)
        assert fl.folded
        dL = torch.zeros_like(img)
        val = float(fl(img, gt, dL))
        torch.cuda.synchronize()
        try:
            ties, taken = _check_against_float64(img, gt, 0.01, (1.0, 0.5, 0.25), val, dL)
            print(f"case {case:3d} {W}x{H}: ok ({ties} near ties, {taken} took the device's sign)", flush=True)
        except AssertionError as e:
            fails += 1
            print(f"case {case:3d} {W}x{H}: FAILED {str(e)[:200]}", flush=True)
        del fl
    print(f"failures: {fails}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
