#!/usr/bin/env python3
"""Generates tests/golden/loss_reference.npz from the REFERENCE's own loss code.

Runs only where /root/reference exists: `make -C oracle ref` compiles oracle/ref/loss_driver.cpp, which #includes
/root/reference/include/loss_utils.h from where it lies and evaluates
    loss = (1 - lambda) * l1_loss(img, gt) + lambda * (1 - ssim(img, gt))      (src/gaussian_trainer.cpp:89-90)
and its autograd gradient on the CPU through LibTorch.  The fixture holds inputs and outputs only (data, no source)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-s"])
import torch  # noqa: E402,F401  (loads libtorch before the driver)
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libloss_ref.so"))
lib.ref_l1_ssim.restype = C.c_int
lib.ref_l1_ssim.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
lib.ref_freq_losses.restype = C.c_int
lib.ref_freq_losses.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

lib.ref_multi_scale_loss.restype = C.c_int
lib.ref_multi_scale_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]

out = {}
cases = [(16, 16, 0.2, 1), (48, 64, 0.2, 2), (37, 53, 0.2, 3), (60, 90, 0.35, 4), (68, 120, 0.2, 5)]
for n, (H, W, lam, seed) in enumerate(cases):
    rng = np.random.default_rng(seed)
    gt = rng.random((3, H, W), dtype=np.float32)
    # a rendered image that resembles the target (smooth perturbation) plus a few exact matches (|d| = 0 sub-gradient)
    img = np.clip(gt + 0.2 * rng.standard_normal((3, H, W)).astype(np.float32), 0.0, 1.0).astype(np.float32)
    img[:, ::7, ::5] = gt[:, ::7, ::5]
    res = np.zeros(3, np.float32)
    dL = np.zeros((3, H, W), np.float32)
    rc = lib.ref_l1_ssim(img.ctypes.data, gt.ctypes.data, H, W, lam, res.ctypes.data, dL.ctypes.data)
    assert rc == 0
    out[f"case{n}_img"], out[f"case{n}_gt"], out[f"case{n}_lambda"] = img, gt, np.float32(lam)
    out[f"case{n}_loss_l1_ssim"], out[f"case{n}_dL_dimg"] = res, dL
    # frequency-domain losses of the mapper (high_frequency_loss cutoff 0.4, low_freq_loss cutoff 0.2) and their gradients
    fr = np.zeros(2, np.float32)
    dH, dLo = np.zeros((3, H, W), np.float32), np.zeros((3, H, W), np.float32)
    assert lib.ref_freq_losses(img.ctypes.data, gt.ctypes.data, H, W, fr.ctypes.data, dH.ctypes.data, dLo.ctypes.data) == 0
    out[f"case{n}_freq_high_low"], out[f"case{n}_dL_high"], out[f"case{n}_dL_low"] = fr, dH, dLo
    # multi_scale_loss over the Replica cfg's three scales (Mapper.scale_num: 3 -> 1, 1/2, 1/4), piece by piece (see the driver)
    scales = np.array([1.0, 0.5, 0.25], np.float32)
    ms = np.zeros(4, np.float32)
    dM = np.zeros((3, H, W), np.float32)
    assert lib.ref_multi_scale_loss(img.ctypes.data, gt.ctypes.data, H, W, scales.ctypes.data, 3, ms.ctypes.data, dM.ctypes.data) == 0
    out[f"case{n}_multi_scale"], out[f"case{n}_dL_multi_scale"] = ms, dM
    print(H, W, lam, res, fr, ms)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss_reference.npz"), **out)
