"""Committed raster fixtures (tests/golden/raster_scenes.npz, made by tests/golden/make_raster_golden.py).

CPU leg: the oracle, rebuilt here, must reproduce the fixture (integers exact; floats to 1e-6 relative -- same source,
possibly another compiler).  GPU leg: the HIP path against the fixture alone, with the parity bar of test_raster_gpu.py.
The fixture pins the oracle against drift; for the rasterizer the reference itself holds no vectors, so
parity with the reference remains "unpinned" there (see the generator's docstring)."""
import hashlib
import os

import numpy as np
import pytest

from segs_slam_amd import scenes
from tests.golden.make_raster_golden import FLOAT_KEYS, FULL_CASES, GRAD_KEYS, INT_KEYS, small_scene, tile_sums

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "raster_scenes.npz"))


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _close(a, b, rel, floor):
    return np.all(np.abs(a - b) <= rel * np.abs(b) + floor)


@pytest.mark.parametrize("case", FULL_CASES, ids=lambda c: f"P{c[0]}")
def test_oracle_reproduces_fixture(case):
    from oracle import gs_oracle
    P = case[0]
    sc = small_scene(*case)
    o, _ = gs_oracle.run_scene(sc, backward=False)
    tag = f"p{P}_"
    assert o.R == int(GOLD[tag + "R"]) and o.sort_bits == int(GOLD[tag + "sort_bits"])
    for k in INT_KEYS:
        assert np.array_equal(o.get(k), GOLD[tag + k]), k
    assert np.array_equal(o.unstable_pixels(1e-5), GOLD[tag + "unstable"])
    for k in FLOAT_KEYS:
        assert _close(o.get(k), GOLD[tag + k], 1e-6, 1e-9), k
    dL = sc.dL_dout_color.copy()
    dL[:, GOLD[tag + "unstable"]] = 0.0
    grads = o.backward(dL)
    for k in GRAD_KEYS:
        assert _close(grads[k], GOLD[tag + k], 1e-5, 1e-7 * np.abs(GOLD[tag + k]).max(initial=1e-30)), k


def test_oracle_reproduces_config1_digests():
    from oracle import gs_oracle
    sc = scenes.make_config_scene("c1")
    o, _ = gs_oracle.run_scene(sc, backward=False)
    assert o.R == int(GOLD["c1_R"])
    for k in INT_KEYS[:-1]:
        assert _digest(o.get(k)) == str(GOLD["c1_sha256_" + k]), k
    unstable = np.unpackbits(GOLD["c1_unstable_packed"])[: sc.camera.width * sc.camera.height].reshape(sc.camera.height, -1).astype(bool)
    assert np.array_equal(o.unstable_pixels(1e-5), unstable)
    assert _digest(o.get("n_contrib")[~unstable]) == str(GOLD["c1_sha256_n_contrib_stable"])
    assert _close(tile_sums(o.get("out_color") * ~unstable[None]), GOLD["c1_out_color_tile_sums"], 1e-6, 1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FULL_CASES, ids=lambda c: f"P{c[0]}")
def test_hip_matches_fixture(case):
    from tests.test_raster_gpu import assert_grad_close, gpu_backward, gpu_forward, gpu_state
    P = case[0]
    sc = small_scene(*case)
    tag = f"p{P}_"
    args, fwd = gpu_forward(sc)
    assert fwd[0] == int(GOLD[tag + "R"])
    if P == 0:
        assert float(fwd[1].abs().max()) == 0.0       # reference short-circuit: zero image (rasterize_points.cu:81)
        return
    g = gpu_state(sc, fwd)
    for k in INT_KEYS[:-1]:
        assert np.array_equal(g[k], GOLD[tag + k]), k
    for k in ("means2D", "depths", "conic_opacity"):
        assert np.array_equal(g[k].view(np.uint32), GOLD[tag + k].view(np.uint32)), k
    ok = ~GOLD[tag + "unstable"]
    assert np.array_equal(g["n_contrib"][ok], GOLD[tag + "n_contrib"][ok])
    assert _close(g["final_T"][ok], GOLD[tag + "final_T"][ok], 1e-4, 2e-6)
    assert _close(g["out_color"][:, ok], GOLD[tag + "out_color"][:, ok], 1e-4, 2e-6)
    dL = sc.dL_dout_color.copy()
    dL[:, ~ok] = 0.0
    got = gpu_backward(sc, args, fwd, dL)
    for k in GRAD_KEYS:
        assert_grad_close(k, got[k].reshape(GOLD[tag + k].shape), GOLD[tag + k])


@pytest.mark.gpu
def test_hip_matches_config1_digests():
    from tests.test_raster_gpu import gpu_backward, gpu_forward, gpu_state
    sc = scenes.make_config_scene("c1")
    H, W = sc.camera.height, sc.camera.width
    args, fwd = gpu_forward(sc)
    assert fwd[0] == int(GOLD["c1_R"])
    g = gpu_state(sc, fwd)
    for k in INT_KEYS[:-1]:
        assert _digest(g[k]) == str(GOLD["c1_sha256_" + k]), k
    unstable = np.unpackbits(GOLD["c1_unstable_packed"])[: W * H].reshape(H, W).astype(bool)
    assert _digest(g["n_contrib"][~unstable]) == str(GOLD["c1_sha256_n_contrib_stable"])
    # per-tile sums of up to 256 pixels each within 1e-4 relative: bound the sum of the per-pixel tolerances
    ts = tile_sums(g["out_color"] * ~unstable[None])
    ref = GOLD["c1_out_color_tile_sums"]
    assert np.all(np.abs(ts - ref) <= 1e-4 * np.abs(ref) + 256 * 2e-6)
    dL = sc.dL_dout_color.copy()
    dL[:, unstable] = 0.0
    got = gpu_backward(sc, args, fwd, dL)
    for k in GRAD_KEYS:
        a = got[k].reshape(sc.P, -1)
        s = GOLD["c1_sample_" + k]
        tol = 1e-4 * np.abs(s) + 1e-5 * (np.abs(s).max() + 1e-30)
        assert np.all(np.abs(a[::97] - s) <= tol), k
        # column sums: errors of P rows add up; bound by 1e-4 of the sum of magnitudes
        assert np.all(np.abs(a.astype(np.float64).sum(0) - GOLD["c1_colsum_" + k]) <= 1e-4 * GOLD["c1_colabs_" + k] + 1e-12), k
