#!/usr/bin/env python3
"""Generates tests/golden/reference_params.json and tests/golden/reference_sh.npz from the REFERENCE's own code.

Runs only where /root/reference exists: `make -C oracle ref` compiles oracle/ref/params_driver.cpp, which #includes
include/gaussian_parameters.h, include/general_utils.h, include/sh_utils.h and compiles src/gaussian_parameters.cpp from where
they lie.  The fixtures hold values only: the parameter structs' defaults, inverse_sigmoid on a grid, and sh_utils::eval_sh /
RGB2SH / SH2RGB on seeded inputs (view directions of a seeded scene)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-s"])
import torch  # noqa: E402,F401  (loads libtorch before the driver)
from segs_slam_amd import scenes  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libparams_ref.so"))
buf = C.create_string_buffer(16384)
n = lib.ref_default_params_json(buf, len(buf))
assert n > 0
params = json.loads(buf.value.decode())
# src/gaussian_parameters.cpp:221 initialises the member from itself -- `opacity_lr_(opacity_lr_)` -- so the constructed value
# is whatever the memory held (e.g. -2.3e-29 in one run); the declared default argument is 0.05f
# (include/gaussian_parameters.h).  Harmless in SEGS-SLAM (_opacity never receives a gradient and is skipped by Adam).
params["optimization"]["opacity_lr"] = None
params["_notes"] = {"optimization.opacity_lr": "uninitialised in the reference (self-initialisation, src/gaussian_parameters.cpp:221); "
                                               "declared default argument 0.05f"}
with open(os.path.join(ROOT, "tests", "golden", "reference_params.json"), "w") as f:
    json.dump(params, f, indent=1, sort_keys=True)

out = {}
x = np.concatenate([np.linspace(0.001, 0.999, 64), [0.1, 0.5, 0.005]]).astype(np.float32)
y = np.zeros_like(x)
lib.ref_inverse_sigmoid(x.ctypes.data_as(C.c_void_p), len(x), y.ctypes.data_as(C.c_void_p))
out["inverse_sigmoid_x"], out["inverse_sigmoid_y"] = x, y

sc = scenes.make_scene(200, 64, 48, 55.0, 55.0, seed=977)
P, M = sc.P, 16
sh = ((scenes.uniform01(P * M * 3, 30, 977).reshape(P, M, 3) - 0.5) * 1.2).astype(np.float32)
sh[:, 0] += 0.4
campos = np.ascontiguousarray(sc.camera.camera_center, np.float32)
d = (sc.means3D - campos).astype(np.float32)
dirs = (d / np.sqrt((d * d).sum(1, keepdims=True, dtype=np.float32))).astype(np.float32)      # forward.cu:27-29: dir / length(dir)
sh_cm = np.ascontiguousarray(sh.transpose(0, 2, 1))                                          # (P, 3, 16): coefficient last, as eval_sh indexes
out["sh_means3D"], out["sh_campos"], out["sh_coeffs"], out["sh_dirs"] = sc.means3D.astype(np.float32), campos, sh, dirs
for deg in range(4):
    res = np.zeros((P, 3), np.float32)
    assert lib.ref_eval_sh(deg, sh_cm.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p), P, M, res.ctypes.data_as(C.c_void_p)) == 0
    out[f"eval_sh_deg{deg}"] = res
rgb = np.linspace(0.0, 1.0, 33).astype(np.float32)
a, b = np.zeros_like(rgb), np.zeros_like(rgb)
lib.ref_rgb2sh(rgb.ctypes.data_as(C.c_void_p), len(rgb), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
out["rgb"], out["rgb2sh"], out["sh2rgb_of_rgb2sh"] = rgb, a, b
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reference_sh.npz"), **out)
print(json.dumps(params)[:300], "...")
print({k: v.shape for k, v in out.items()})
