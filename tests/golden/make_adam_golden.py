#!/usr/bin/env python3
"""Generates tests/golden/adam_libtorch.npz from LibTorch's C++ torch::optim::Adam configured as the reference does
(src/gaussian_model.cpp:632-640) -- oracle/ref/adam_driver.cpp.  Inputs and outputs only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-s"])
import torch  # noqa: E402,F401
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libadam_ref.so"))
lib.ref_adam_steps.restype = C.c_int
lib.ref_adam_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]

rng = np.random.default_rng(11)
n, steps, lr = 4099, 5, 0.005
p0 = rng.standard_normal(n).astype(np.float32)
grads = (rng.standard_normal((steps, n)) * np.logspace(-6, 0, n)[None, :]).astype(np.float32)   # 6 decades of magnitudes
grads[:, ::17] = 0.0                                                                              # rows that never move
p = p0.copy()
m = np.zeros(n, np.float32)
v = np.zeros(n, np.float32)
assert lib.ref_adam_steps(p.ctypes.data, grads.ctypes.data, n, steps, lr, m.ctypes.data, v.ctypes.data) == 0
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "adam_libtorch.npz"), p0=p0, grads=grads, lr=np.float64(lr),
                    p=p, exp_avg=m, exp_avg_sq=v, torch_version=np.array(torch.__version__))
print("max |dp|", np.abs(p - p0).max(), torch.__version__)
