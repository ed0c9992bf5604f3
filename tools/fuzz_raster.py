#!/usr/bin/env python3
"""Randomised parity sweep of the rasterizer against the CPU oracle (run on the GPU box; not part of the test suite):
tools/fuzz_raster.py [n_cases] [seed0].  Scenes vary in P, image size (also non-multiples of 16), focal length, background,
scale multiplier, opacity range; every case goes through tests/test_raster_gpu.run_parity (bit-exact integers, tolerances on
floats) and then through the resident entry points (tight binning, dead-instance drop, 9-bit depth sort), whose image
must equal the reference-shaped path's bit for bit and whose gradients must agree within the atomics' tolerance."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from segs_slam_amd import scenes  # noqa: E402
import test_raster_gpu as t  # noqa: E402
import torch  # noqa: E402
from segs_slam_amd.raster_engine import RasterEngine  # noqa: E402


def resident_matches_sync(sc):
    cam = sc.camera
    a = [t._t(x) for x in (sc.bg, sc.means3D, sc.colors, sc.opacity, sc.scales, sc.rotations, cam.world_view_transform,
                           cam.full_proj_transform, cam.camera_center)]
    dL = t._t(sc.dL_dout_color)
    outs = []
    for resident in (False, True):
        eng = RasterEngine(sc.P, cam.width, cam.height, t.DEV, resident=resident)
        for _ in range(2):
            img = eng.forward(*a, cam.tanfovx, cam.tanfovy).clone()
            eng.backward(dL)
        assert eng.check(raise_on_overflow=False)
        torch.cuda.synchronize()
        assert eng._last_resident == resident
        outs.append((img, eng.radii.clone(), {k: v.cpu().numpy().copy() for k, v in eng.grads.items()}))
    assert torch.equal(outs[0][0], outs[1][0]), "resident image differs"
    assert torch.equal(outs[0][1], outs[1][1]), "resident radii differ"
    for k in outs[0][2]:
        t.assert_grad_close(k, outs[1][2][k], outs[0][2][k])


def float64_arbiter(sc):
    """A case that misses the gradient bar against the float32 CPU oracle goes to the independent float64 autograd formulation
    (oracle/torch_ref.py; feasible for small images): the device passes if every entry of the five parameter
    gradients is within 1e-4 relative + 2e-5 of the tensor's largest entry of the float64 value, and it is inside the pure
    1e-4 relative bound on at least as many entries (-0.5 %) as the oracle is.  (Scenes of a few hundred Gaussians that each
    cover thousands of pixels: float32 sums of both signs, where the oracle is no closer to the truth than the kernels.)"""
    from oracle import torch_ref
    o, _ = t.gs_oracle.run_scene(sc, backward=False)
    if int((o.get("radii") > 0).sum()) * sc.camera.width * sc.camera.height > 60_000_000:   # the float64 loop is per Gaussian x pixel (seconds up to here, many minutes at 60 000 Gaussians)
        return None
    unstable = o.unstable_pixels(3e-3)
    dL = sc.dL_dout_color.copy()
    dL[:, unstable] = 0.0
    args, fwd = t.gpu_forward(sc)
    g = t.gpu_state(sc, fwd)
    got = t.gpu_backward(sc, args, fwd, dL)
    ref = o.backward(dL)
    cam = sc.camera
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)  # noqa: E731
    m, s, r, op, col = t64(sc.means3D), t64(sc.scales), t64(sc.rotations), t64(sc.opacity), t64(sc.colors)
    img, _p = torch_ref.render(m, s, r, op, col, torch.tensor(sc.bg, dtype=torch.float64), torch.tensor(cam.world_view_transform),
                               torch.tensor(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, cam.height, cam.width,
                               torch.tensor(g["radii"]), torch.tensor(g["means2D"]), sc.scale_modifier)
    (img * torch.tensor(dL, dtype=torch.float64)).sum().backward()
    truth = dict(dL_dmean3D=m.grad.numpy(), dL_dscale=s.grad.numpy(), dL_drot=r.grad.numpy(), dL_dopacity=op.grad.numpy(),
                 dL_dcolor=col.grad.numpy())
    for k, want in truth.items():
        top = np.abs(want).max()
        if top == 0:
            continue
        nz = want != 0
        frac = {}
        for who, have in (("device", got[k]), ("oracle", ref[k])):
            err = np.abs(have.reshape(want.shape).astype(np.float64) - want)
            frac[who] = float((err[nz] <= 1e-4 * np.abs(want[nz])).mean())
            if who == "device" and not np.all(err <= 1e-4 * np.abs(want) + 2e-5 * top):
                return f"{k}: device {float(err.max() / top):.2e} of max off the float64 value"
        if frac["device"] < frac["oracle"] - 0.005:
            return f"{k}: device within 1e-4 on {frac['device']:.4f}, oracle on {frac['oracle']:.4f}"
    return "ok"


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fails = 0
for i in range(n_cases):
    sc = t.fuzz_scene(seed0 + i)
    P, W, H = sc.P, sc.camera.width, sc.camera.height
    try:
        # the oracle comparison needs a scene whose compositing decisions are not threshold-adjacent on > 1 % of the pixels
        # (dim, huge Gaussians put most alphas next to 1/255); such scenes still go through the resident-vs-sync check
        o, _ = t.gs_oracle.run_scene(sc, backward=False)
        comparable = o.unstable_pixels(1e-5).mean() < 0.01
        status = "ok" if comparable else "ok (resident vs sync only)"
        if comparable:
            o2, _ = t.run_parity(sc, backward=True)      # rule (3) of assert_grad_close: float64 arbiter for the per-Gaussian stage
            arb = {k: v for k, v in o2.grad_verdicts.items() if v != "pure"}
            if arb:
                status = "ok (" + "; ".join(f"{k}: {v}" for k, v in arb.items()) + ")"
        if P > 0:
            resident_matches_sync(sc)
    except AssertionError as e:
        verdict = None
        if "dL_d" in str(e):   # a tensor outside the stage (or a small scene): the independent float64 autograd formulation
            verdict = float64_arbiter(sc)
        if verdict == "ok":
            status = "ok (float64 autograd arbiter; against the float32 oracle: " + str(e)[:80] + ")"
        else:
            status = "FAIL " + str(e)[:200] + (" | arbiter: " + verdict if verdict else "")
            fails += 1
    print(f"case {i:3d} seed {seed0 + i} P={P:6d} {W}x{H} scale_mul={float(sc.scales.mean()):.4f}: {status}", flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
