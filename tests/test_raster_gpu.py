"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI via the
reference-shaped wrappers, against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact on every integer / index product of binning (radii,
tiles_touched, point_offsets, sort keys, point_list, ranges) and on the per-Gaussian floats that feed
them; <= 1e-4 relative on rendered RGB and gradients.

Tolerances, stated once:
  * per-Gaussian forward floats (means2D, depth, conic): exact (same expression order, no contraction).
  * image / final_T: |gpu - ref| <= 1e-4 * |ref| + 2e-6 on pixels whose compositing decisions are not
    within 1e-5 (relative) of a threshold (power>0, alpha<1/255, T<1e-4) -- v_exp_f32 and glibc expf differ
    in the last bits, so a decision that close to its threshold may legitimately flip; such pixels are
    reported by the oracle (gso_unstable_pixels) and must be < 1 % of the image.
  * gradients: dL_dout_color is zeroed on those pixels for BOTH sides, then
    |gpu - ref| <= 1e-4 * |ref| + 1e-5 * max|ref| per tensor (float atomics sum in arbitrary order; the
    oracle sums in double) AND at least 99 % of the non-zero entries inside the pure 1e-4 relative bound.
    Why the floor is 1e-5 of the largest entry and not SURVEY 8c's absolute 1e-7 / a 1e-6 fraction: measured at BASELINE's
    sizes (profiles/r02_grad_error.txt, tools/grad_error_report.py) 99.5-99.7 % of the non-zero entries of every gradient
    tensor are inside the pure relative bound, and the largest error of any entry is 0.5e-6 ... 5.8e-6 of max|ref| (dL_dcov3D
    and dL_dscale, which go through K12's cancelling terms, are the worst): that is the float32 accumulation error of a sum
    of hundreds to thousands of pixel terms of both signs -- the reference's own atomics have it too -- so a 1e-6 floor
    fails single entries by a factor 2-6 in a run-dependent way while 1e-5 leaves a factor 2 over the worst observed.
    (The synthetic scenes' dL/dimage is U(-1,1)/(3HW), so an absolute floor would mean something else at every image size.)
"""
import numpy as np
import pytest
import torch

from oracle import gs_oracle
from segs_slam_amd import scenes
from tests.test_oracle import check_binning_invariants

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


def gpu_forward(sc):
    from segs_slam_amd import rasterize_points as rp
    cam = sc.camera
    e = torch.empty(0, device=DEV)
    args = dict(bg=_t(sc.bg), means3D=_t(sc.means3D), colors=_t(sc.colors), opacity=_t(sc.opacity), scales=_t(sc.scales),
                rotations=_t(sc.rotations), view=_t(cam.world_view_transform), proj=_t(cam.full_proj_transform),
                campos=_t(cam.camera_center))
    R, color, radii, geom, binning, img = rp.RasterizeGaussiansCUDA(
        args["bg"], args["means3D"], args["colors"], args["opacity"], args["scales"], args["rotations"], sc.scale_modifier,
        e, args["view"], args["proj"], cam.tanfovx, cam.tanfovy, cam.height, cam.width, e, 0, args["campos"], False)
    return args, (R, color, radii, geom, binning, img)


def gpu_backward(sc, args, fwd, dL):
    from segs_slam_amd import rasterize_points as rp
    cam = sc.camera
    e = torch.empty(0, device=DEV)
    R, color, radii, geom, binning, img = fwd
    out = rp.RasterizeGaussiansBackwardCUDA(args["bg"], args["means3D"], radii, args["colors"], args["scales"],
                                            args["rotations"], sc.scale_modifier, e, args["view"], args["proj"],
                                            cam.tanfovx, cam.tanfovy, _t(dL), e, 0, args["campos"], geom, R, binning, img)
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
    return {k: v.cpu().numpy() for k, v in zip(names, out)}


def gpu_state(sc, fwd):
    from segs_slam_amd import rasterize_points as rp
    R, color, radii, geom, binning, img = fwd
    st = rp.debug_state(sc.P, sc.camera.width, sc.camera.height, R, radii, geom, binning, img)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in st.items()}
    out["tiles_touched"] = out["tiles_touched"].view(np.uint32)
    out["point_offsets"] = out["point_offsets"].view(np.uint32)
    out["keys"] = out["keys"].view(np.uint64)
    out["point_list"] = out["point_list"].view(np.uint32)
    out["ranges"] = out["ranges"].view(np.uint32)
    out["n_contrib"] = out["n_contrib"].view(np.uint32)
    out["radii"] = radii.cpu().numpy()
    out["out_color"] = color.cpu().numpy()
    return out


def assert_forward_parity(sc, o, g, R):
    assert R == o.R
    for k in ("radii", "tiles_touched", "point_offsets", "keys", "point_list", "ranges"):
        assert np.array_equal(g[k], o.get(k)), k
    for k in ("means2D", "depths", "conic_opacity"):
        assert np.array_equal(g[k].view(np.uint32), o.get(k).view(np.uint32)), k
    unstable = o.unstable_pixels(1e-5)
    assert unstable.mean() < 0.01
    ok = ~unstable
    assert np.array_equal(g["n_contrib"][ok], o.get("n_contrib")[ok])
    for name, a, b in (("final_T", g["final_T"], o.get("final_T")),):
        assert np.all(np.abs(a - b)[ok] <= 1e-4 * np.abs(b)[ok] + 2e-6), name
    a, b = g["out_color"], o.get("out_color")
    err = np.abs(a - b)[:, ok]
    assert np.all(err <= 1e-4 * np.abs(b)[:, ok] + 2e-6), float(err.max())
    return unstable


GRAD_REL, GRAD_FLOOR, GRAD_PURE_FRACTION = 1e-4, 1e-5, 0.99
STAGE_TENSORS = ("dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot")     # outputs of the per-Gaussian backward stage (K12 + K13 + cov3D)


def pure_fraction(err, ref):
    nz = ref != 0
    return float((err[nz] <= GRAD_REL * np.abs(ref[nz])).mean()) if nz.any() else 1.0


PLAIN_BAR_TENSORS = ("dL_dcolor", "dL_dopacity", "dL_dmean2D", "colors", "opacity")   # (the engine's bucket names the first two by field)
PLAIN_BAR_SIGNIFICANT = 0.02


def assert_grad_close(name, a, b, arbiter=None):
    """The gradient bar (a = device, b = float32 CPU oracle):
      (1) every entry within 1e-4 relative + 1e-5 of the tensor's largest entry (the floor covers entries that are the float32
          sum of large terms of both signs -- SURVEY 8c asks for 1e-7 absolute, which float atomics in arbitrary order cannot
          give on such entries; profiles/r02_grad_error.txt);
      (2) at least 99 % of the non-zero entries inside the PURE 1e-4 relative bound, no floor;
      (3) where (2) fails on an output of the per-Gaussian backward stage -- scenes whose few visible Gaussians each cover
          thousands of pixels: float32 sums of both signs in the ORACLE too -- `arbiter(name)` evaluates that stage in float64
          from its defining equations on the oracle's own inputs (oracle/preprocess_backward_f64.py) and returns
          (device stage output, oracle stage output, float64 value): the device passes iff it is inside (1) against the
          float64 value and inside the pure bound on at least as many entries as the oracle is (-0.5 %).  The oracle is only as
          good a reference as float32 lets it be; the rule says so instead of a hand-kept list of accepted seeds.
      (4) for dL_dcolor / dL_dopacity / dL_dmean2D: the plain 1e-4 relative bar, no floor, on every entry of at least 2 % of
          the tensor's largest (see below)."""
    err = np.abs(a - b)
    tol = GRAD_REL * np.abs(b) + GRAD_FLOOR * (np.abs(b).max() + 1e-30)
    bad = err > tol
    assert not bad.any(), (name, int(bad.sum()), float(err.max()), float(np.abs(b).max()))
    if name in PLAIN_BAR_TENSORS:
        # (4) north_star's bar as stated -- 1e-4 relative, NO floor -- on every entry that is not cancellation noise: the
        # outputs the tile backward accumulates directly, wherever the entry reaches 2 % of the tensor's largest (the floor of
        # rule (1) is 1e-5 of the largest, i.e. it only ever decides entries 20 times smaller than these).  A regression of the
        # tile kernels cannot hide behind the floor here.
        big = np.abs(b) >= PLAIN_BAR_SIGNIFICANT * np.abs(b).max()
        over = big & (err > GRAD_REL * np.abs(b))
        assert not over.any(), (name, "plain 1e-4 relative bar", int(over.sum()), int(big.sum()), float((err[big] / np.abs(b[big])).max()))
    nz = b != 0
    if nz.sum() >= 1000:     # most entries must pass WITHOUT the floor (it only covers the cancellation-dominated ones)
        pure = pure_fraction(err, b)
        if pure >= GRAD_PURE_FRACTION:
            return "pure"
        assert arbiter is not None and name in STAGE_TENSORS, (name, pure)
        dev64, ora64, truth = arbiter(name)
        e_dev, e_ora = np.abs(dev64 - truth), np.abs(ora64 - truth)
        top = np.abs(truth).max()
        assert np.all(e_dev <= GRAD_REL * np.abs(truth) + GRAD_FLOOR * top), (name, "device vs float64", float(e_dev.max() / top))
        f_dev, f_ora = pure_fraction(e_dev, truth), pure_fraction(e_ora, truth)
        assert f_dev >= f_ora - 0.005, (name, "device worse than the oracle against float64", f_dev, f_ora, pure)
        return f"arbiter: device {f_dev:.4f} / oracle {f_ora:.4f} of the entries within 1e-4 of float64 (device vs oracle {pure:.4f})"
    return "pure"


def stage_arbiter(sc, o, ref):
    """name -> (device, oracle, float64) outputs of the per-Gaussian backward stage, all three fed the ORACLE's dL_dmean2D /
    dL_dconic (so that only this stage's arithmetic is compared).  Evaluated lazily, once."""
    import ctypes as C
    from segs_slam_amd import _capi
    from oracle.preprocess_backward_f64 import stage_f64
    cache = {}

    def get(name):
        if not cache:
            cam, P = sc.camera, sc.P
            m3, sca, rot = _t(sc.means3D), _t(sc.scales), _t(sc.rotations)
            view, proj = _t(cam.world_view_transform), _t(cam.full_proj_transform)
            radii = _t(o.get("radii"), torch.int32)
            d2, dc = _t(ref["dL_dmean2D"]), _t(ref["dL_dconic"])
            outs = [torch.empty((P, n), device=DEV) for n in (3, 6, 3, 4)]
            p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
            _capi.check(_capi.lib().segs_debug_preprocess_backward(P, cam.width, cam.height, p(m3), p(radii), p(sca), 1.0, p(rot), None,
                                                                   p(view), p(proj), cam.tanfovx, cam.tanfovy, p(d2), p(dc),
                                                                   *[p(x) for x in outs],
                                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "segs_debug_preprocess_backward")
            torch.cuda.synchronize()
            truth = stage_f64(sc.means3D, sc.scales, sc.rotations, cam.world_view_transform, cam.full_proj_transform, cam.width,
                              cam.height, cam.tanfovx, cam.tanfovy, ref["dL_dmean2D"], ref["dL_dconic"], o.get("radii"))
            for x, k in zip(outs, STAGE_TENSORS):
                cache[k] = (x.cpu().numpy().astype(np.float64), ref[k].astype(np.float64).reshape(truth[k].shape), truth[k])
        return cache[name]
    return get


def run_parity(sc, backward=True):
    o, _ = gs_oracle.run_scene(sc, backward=False)
    args, fwd = gpu_forward(sc)
    g = gpu_state(sc, fwd)
    unstable = assert_forward_parity(sc, o, g, fwd[0])
    if sc.P:
        check_binning_invariants(sc.P, sc.camera.width, sc.camera.height, g["radii"], g["means2D"], g["depths"],
                                 g["tiles_touched"], g["point_offsets"], g["keys"], g["point_list"], g["ranges"], fwd[0],
                                 o.sort_bits)
    if backward:
        dL = sc.dL_dout_color.copy()
        dL[:, unstable] = 0.0
        ref = o.backward(dL)
        got = gpu_backward(sc, args, fwd, dL)
        arb = stage_arbiter(sc, o, ref)
        how = {}
        for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot"):
            how[k] = assert_grad_close(k, got[k].reshape(ref[k].shape), ref[k], arbiter=arb)
        o.grad_verdicts = how
    return o, g


@pytest.mark.parametrize("P,W,H,bg", [(1000, 64, 64, (0.1, 0.2, 0.3)), (17, 33, 17, (0, 0, 0)), (5000, 200, 120, (1, 1, 1)),
                                      (1, 16, 16, (0, 0, 0))])
def test_small_scenes(P, W, H, bg):
    sc = scenes.make_scene(P, W, H, 0.9 * W, 0.9 * W, seed=4000 + P, bg=bg)
    sc.scales *= 3.0
    sc.dL_dout_color[:] = (scenes.uniform01(sc.dL_dout_color.size, 55, P).reshape(sc.dL_dout_color.shape) * 2 - 1)
    run_parity(sc)


def test_matrix_pipe_gaussian_role_matches_oracle():
    """render_bwd_mfma_kernel (SEGS_RASTER_MFMA_MOMENTS): the tile backward's nine sums per Gaussian as v_mfma_f32_16x16x4_f32
    products with moments about the quadrant centre -- the measured A/B partner of the default kernel -- held
    to the same gradient bar, unchanged, on a deep small scene and on BASELINE config 1."""
    from segs_slam_amd import _capi
    lib = _capi.lib()
    old = lib.segs_raster_set_flags(64)
    try:
        sc = scenes.make_scene(5000, 200, 120, 180.0, 180.0, seed=9005, bg=(0.1, 0.0, 0.2))
        sc.scales *= 3.0
        sc.dL_dout_color[:] = (scenes.uniform01(sc.dL_dout_color.size, 56, 5000).reshape(sc.dL_dout_color.shape) * 2 - 1)
        run_parity(sc)
        run_parity(scenes.make_config_scene("c1"))
    finally:
        lib.segs_raster_set_flags(old)


def test_config1_50k_640x480():
    """BASELINE.json configs[0]: 50k Gaussians, 640x480 (forward + backward here)."""
    sc = scenes.make_config_scene("c1")
    o, g = run_parity(sc)
    assert 100_000 < o.R < 250_000


def test_degenerate_inputs():
    from segs_slam_amd import rasterize_points as rp
    e = torch.empty(0, device=DEV)
    # P = 0: the tensor-level entry short-circuits like the reference (zero image, R = 0)
    sc = scenes.make_scene(0, 32, 32, 30.0, 30.0, bg=(0.5, 0.25, 0.125))
    args, fwd = gpu_forward(sc)
    assert fwd[0] == 0 and float(fwd[1].abs().max()) == 0.0
    grads = gpu_backward(sc, args, fwd, sc.dL_dout_color)
    assert all(v.size == 0 or np.all(v == 0) for v in grads.values())
    # all culled -> R = 0, image = background
    sc = scenes.make_scene(64, 40, 24, 30.0, 30.0, seed=3, bg=(0.5, 0.25, 0.125))
    sc.means3D[:, 2] = 0.1
    o, g = run_parity(sc)
    assert o.R == 0 and np.allclose(g["out_color"][0], 0.5)
    # alpha < 1/255 everywhere
    sc = scenes.make_scene(200, 48, 48, 40.0, 40.0, seed=4, bg=(0.3, 0.3, 0.3))
    sc.opacity[:] = 0.003
    o, g = run_parity(sc)
    assert np.all(g["n_contrib"] == 0)
    # one Gaussian spanning the whole non-multiple-of-16 image
    sc = scenes.make_scene(1, 50, 35, 40.0, 40.0, seed=6)
    sc.means3D[:] = [0, 0, 1.0]
    sc.scales[:] = 2.0
    o, g = run_parity(sc)
    assert o.R == 12
    # bad rank -> RuntimeError like AT_ERROR (rasterize_points.cu:57-59)
    with pytest.raises(RuntimeError):
        rp.RasterizeGaussiansCUDA(e, torch.zeros(4, 2, device=DEV), e, e, e, e, 1.0, e, e, e, 1.0, 1.0, 8, 8, e, 0, e, False)


def _stage_outputs(sc, o, ref, cov3D_precomp=None):
    """(device outputs, float64 truth) of the per-Gaussian backward stage fed the oracle's dL_dmean2D / dL_dconic."""
    import ctypes as C
    from oracle.preprocess_backward_f64 import stage_f64
    from segs_slam_amd import _capi
    cam = sc.camera
    P = sc.P
    m3, sca, rot = _t(sc.means3D), _t(sc.scales), _t(sc.rotations)
    view, proj = _t(cam.world_view_transform), _t(cam.full_proj_transform)
    radii = _t(o.get("radii"), torch.int32)
    d2, dc = _t(ref["dL_dmean2D"]), _t(ref["dL_dconic"])
    outs = [torch.empty((P, n), device=DEV) for n in (3, 6, 3, 4)]
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    pre = _t(cov3D_precomp) if cov3D_precomp is not None else None
    st = _capi.lib().segs_debug_preprocess_backward(P, cam.width, cam.height, p(m3), p(radii), None if pre is not None else p(sca), 1.0,
                                                    None if pre is not None else p(rot), p(pre) if pre is not None else None,
                                                    p(view), p(proj), cam.tanfovx, cam.tanfovy, p(d2), p(dc),
                                                    p(outs[0]), p(outs[1]), None if pre is not None else p(outs[2]),
                                                    None if pre is not None else p(outs[3]), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _capi.check(st, "segs_debug_preprocess_backward")
    torch.cuda.synchronize()
    truth = stage_f64(sc.means3D, sc.scales, sc.rotations, cam.world_view_transform, cam.full_proj_transform, cam.width, cam.height,
                      cam.tanfovx, cam.tanfovy, ref["dL_dmean2D"], ref["dL_dconic"], o.get("radii"))
    return outs, truth


def test_preprocess_backward_alone_matches_float64_evaluation():
    """K12 + K13 + cov3D backward alone, fed the oracle's dL_dmean2D / dL_dconic.  The device code is derived from the forward
    map (preprocess.hip: dL/dC = -k adj(C) G adj(C), factorised through U = M2 L) instead of following backward.cu's
    expression tree, so device and oracle are two float32 evaluations of the same function; the arbiter is a float64
    evaluation of the defining equations (oracle/preprocess_backward_f64.py).  Both must sit within float32 rounding of it:
    1e-5 relative + 4e-6 of the tensor's largest entry (measured: device 1e-7 ... 1e-6, oracle 2e-7 ... 2.4e-6 of max|truth|)."""
    sc = scenes.make_scene(4000, 128, 96, 100.0, 100.0, seed=77, bg=(0.2, 0.2, 0.2))
    sc.scales *= 3.0
    o, ref = gs_oracle.run_scene(sc)
    outs, truth = _stage_outputs(sc, o, ref)
    ref2 = o.backward(sc.dL_dout_color, ref["dL_dmean2D"], ref["dL_dconic"])
    for t, k in zip(outs, STAGE_TENSORS):
        want = truth[k]
        assert np.abs(want).max() > 0
        for who, got in (("device", t.cpu().numpy().astype(np.float64)), ("oracle", ref2[k].astype(np.float64))):
            err = np.abs(got - want)
            assert np.all(err <= 1e-5 * np.abs(want) + 4e-6 * np.abs(want).max()), (who, k, float(err.max()), float(np.abs(want).max()))


def test_preprocess_backward_clamped_jacobian_and_precomputed_covariance_branches():
    """Two branches of the stage that ordinary scenes reach only by chance (fuzz sweeps), pinned deterministically:
      * the clamped Jacobian (backward.cu:175-176, forward.cu:84-89: t.x / t.z limited to 1.3 tan(fov/2), the clamped
        coordinate passes no gradient): a scene widened to 1.5 tan(fov/2) with Gaussians large enough to stay visible there;
      * cov3D_precomp (backward.cu:287-290 is skipped, dL_dcov3D is the output): the same scene handed over as covariances."""
    sc = scenes.make_scene(4000, 128, 96, 100.0, 100.0, seed=78, bg=(0.0, 0.0, 0.0))
    sc.means3D[:, :2] *= 1.25          # make_scene draws |x| <= 1.2 z tan(fov/2): now up to 1.5
    sc.scales *= 10.0
    o, ref = gs_oracle.run_scene(sc)
    cam = sc.camera
    V = cam.world_view_transform.reshape(4, 4).astype(np.float64)
    t = sc.means3D.astype(np.float64) @ V[:3, :3] + V[3, :3]
    vis = o.get("radii") > 0
    clamped = vis & ((np.abs(t[:, 0] / t[:, 2]) > 1.3 * cam.tanfovx) | (np.abs(t[:, 1] / t[:, 2]) > 1.3 * cam.tanfovy))
    assert clamped.sum() > 100, int(clamped.sum())
    outs, truth = _stage_outputs(sc, o, ref)
    ref2 = o.backward(sc.dL_dout_color, ref["dL_dmean2D"], ref["dL_dconic"])
    for tns, k in zip(outs, STAGE_TENSORS):
        want = truth[k]
        sub = want[clamped]
        assert np.abs(sub).max() > 0
        for who, got in (("device", tns.cpu().numpy().astype(np.float64)), ("oracle", ref2[k].astype(np.float64))):
            err = np.abs(got - want)
            assert np.all(err <= 1e-5 * np.abs(want) + 4e-6 * np.abs(want).max()), (who, k, float(err.max()), float(np.abs(want).max()))
    # the same Gaussians as precomputed covariances (R S S^T R^T, the quaternion used as given, forward.cu:118-152)
    r, x, y, z = sc.rotations.astype(np.float64).T
    R = np.stack([np.stack([1 - 2 * (y*y + z*z), 2 * (x*y - r*z), 2 * (x*z + r*y)], 1),
                  np.stack([2 * (x*y + r*z), 1 - 2 * (x*x + z*z), 2 * (y*z - r*x)], 1),
                  np.stack([2 * (x*z - r*y), 2 * (y*z + r*x), 1 - 2 * (x*x + y*y)], 1)], 1)
    L = R * sc.scales.astype(np.float64)[:, None, :]
    S = L @ L.transpose(0, 2, 1)
    cov6 = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    outs_pre, _ = _stage_outputs(sc, o, ref, cov3D_precomp=cov6)
    for tns, k in zip(outs_pre[:2], STAGE_TENSORS[:2]):
        want = truth[k]
        err = np.abs(tns.cpu().numpy().astype(np.float64) - want)
        # (the covariances were rounded to float32 once more on the way in: 1e-5 relative + 1e-5 of max)
        assert np.all(err <= 1e-5 * np.abs(want) + 1e-5 * np.abs(want).max()), (k, float(err.max()), float(np.abs(want).max()))


# sizes on both sides of the count kernel's chunk policy (1 / 2 / 4 tiles per chunk at <= 256 / <= 2048 / more sort tiles) and
# grids that are not multiples of the 8 XCDs the scatter and count kernels deal their tiles to
@pytest.mark.parametrize("n,end_bit", [(1, 44), (63, 40), (4096, 44), (4097, 45), (100_000, 44), (1_000_003, 48), (5000, 7),
                                       (524_288, 44), (524_289, 33), (4_194_305, 36)])
def test_sort_pairs(n, end_bit):
    """Stable LSD radix sort on key bits [0,end_bit) == numpy stable argsort of the masked keys."""
    import ctypes as C
    from segs_slam_amd import _capi
    rng = np.random.default_rng(n)
    # skewed keys: few distinct tile ids / exponent bytes, many ties
    keys = (rng.integers(0, 3000, n, dtype=np.uint64) << np.uint64(32)) | \
           (rng.integers(0x3E000000, 0x40C00000, n, dtype=np.uint64) & np.uint64(0xFFFFF000))
    vals = np.arange(n, dtype=np.uint32)
    mask = np.uint64((1 << end_bit) - 1)
    order = np.argsort(keys & mask, kind="stable")
    k_in, v_in = _t(keys.view(np.int64), torch.int64), _t(vals.view(np.int32), torch.int32)
    k_out, v_out = torch.empty_like(k_in), torch.empty_like(v_in)
    l = _capi.lib()
    temp = torch.empty(l.segs_binning_bytes(n), dtype=torch.uint8, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _capi.check(l.segs_sort_pairs(p(k_in), p(v_in), p(k_out), p(v_out), n, end_bit, p(temp),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "segs_sort_pairs")
    torch.cuda.synchronize()
    assert np.array_equal(v_out.cpu().numpy().view(np.uint32), vals[order])
    assert np.array_equal(k_out.cpu().numpy().view(np.uint64), keys[order])


def test_visible_filter_and_mark_visible():
    from segs_slam_amd import rasterize_points as rp
    sc = scenes.make_scene(30_000, 320, 240, 250.0, 260.0, seed=31)
    cam = sc.camera
    e = torch.empty(0, device=DEV)
    radii = rp.RasterizeGaussiansfilterCUDA(_t(sc.means3D), _t(sc.scales), _t(sc.rotations), 1.0, e,
                                            _t(cam.world_view_transform), _t(cam.full_proj_transform), cam.tanfovx,
                                            cam.tanfovy, cam.height, cam.width, False, False)
    ref = gs_oracle.visible_filter(sc.means3D, sc.scales, sc.rotations, 1.0, cam.world_view_transform,
                                   cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.height, cam.width)
    assert np.array_equal(radii.cpu().numpy(), ref)
    pres = rp.markVisible(_t(sc.means3D), _t(cam.world_view_transform), _t(cam.full_proj_transform))
    assert np.array_equal(pres.cpu().numpy(), gs_oracle.mark_visible(sc.means3D, cam.world_view_transform, cam.full_proj_transform))
    # strided scales view (exp(_scaling)[:, :3] of an (A,6) tensor, src/gaussian_renderer.cpp:166-170) is accepted
    wide = torch.cat([_t(sc.scales), _t(sc.scales) * 2], dim=1)
    radii2 = rp.RasterizeGaussiansfilterCUDA(_t(sc.means3D), wide[:, :3], _t(sc.rotations), 1.0, e,
                                             _t(cam.world_view_transform), _t(cam.full_proj_transform), cam.tanfovx,
                                             cam.tanfovy, cam.height, cam.width, False, False)
    assert torch.equal(radii, radii2)


def test_autograd_module_matches_oracle():
    """GaussianRasterizer module (include/gaussian_rasterizer.h) end to end through autograd."""
    from segs_slam_amd.gaussian_rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    sc = scenes.make_scene(3000, 96, 80, 80.0, 80.0, seed=91, bg=(0.1, 0.0, 0.2))
    sc.scales *= 3.0
    cam = sc.camera
    o, _ = gs_oracle.run_scene(sc, backward=False)
    unstable = o.unstable_pixels(1e-5)
    dL = (scenes.uniform01(sc.dL_dout_color.size, 56, 91).reshape(sc.dL_dout_color.shape) * 2 - 1).astype(np.float32)
    dL[:, unstable] = 0
    ref = o.backward(dL)
    bg = _t(sc.bg)
    view, proj, campos = _t(cam.world_view_transform), _t(cam.full_proj_transform), _t(cam.camera_center)
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, bg, 1.0, view, proj, 0, campos, False)
    rast = GaussianRasterizer(rs)
    leaf = lambda a: _t(a).requires_grad_(True)  # noqa: E731
    m3, op, sca, rot = leaf(sc.means3D), leaf(sc.opacity), leaf(sc.scales), leaf(sc.rotations)
    # colours arrive as a strided view of a wider tensor (src/gaussian_renderer.cpp:319-324)
    wide = torch.cat([_t(sc.colors), torch.zeros(sc.P, 19, device=DEV)], dim=1).requires_grad_(True)
    col = wide[:, :3]
    means2D = torch.zeros_like(m3, requires_grad=True)
    img, radii = rast(m3, means2D, op, False, True, True, True, False, colors_precomp=col, scales=sca, rotations=rot)
    (img * _t(dL)).sum().backward()
    assert np.array_equal(radii.cpu().numpy(), o.get("radii"))
    assert_grad_close("means3D", m3.grad.cpu().numpy(), ref["dL_dmean3D"])
    assert_grad_close("means2D", means2D.grad.cpu().numpy(), ref["dL_dmean2D"])
    assert_grad_close("opacity", op.grad.cpu().numpy(), ref["dL_dopacity"])
    assert_grad_close("scales", sca.grad.cpu().numpy(), ref["dL_dscale"])
    assert_grad_close("rotations", rot.grad.cpu().numpy(), ref["dL_drot"])
    assert_grad_close("colors", wide.grad[:, :3].cpu().numpy(), ref["dL_dcolor"])
    with pytest.raises(RuntimeError):
        rast(m3, means2D, op, True, True, True, True, False, colors_precomp=col, scales=sca, rotations=rot)


def test_resident_engine_matches_sync_path():
    """The no-host-sync entry points (segs_rasterize_*_resident) give bit-identical images and radii, and the same
    per-Gaussian backward products, as the reference-shaped synchronising call -- with the reference's full instance
    lists (keep_dead_instances) and with the default tight binning that leaves out instances no pixel can see."""
    from segs_slam_amd.raster_engine import RasterEngine
    sc = scenes.make_scene(30_000, 320, 240, 260.0, 260.0, seed=17, bg=(0.1, 0.2, 0.3))
    sc.scales *= 2.0
    cam = sc.camera
    a = dict(bg=_t(sc.bg), m=_t(sc.means3D), c=_t(sc.colors), o=_t(sc.opacity), s=_t(sc.scales), r=_t(sc.rotations),
             v=_t(cam.world_view_transform), p=_t(cam.full_proj_transform), cp=_t(cam.camera_center))
    dL = _t(sc.dL_dout_color)
    outs = []
    for resident, keep in ((False, False), (True, True), (True, False)):
        eng = RasterEngine(sc.P, cam.width, cam.height, DEV, resident=resident, keep_dead_instances=keep, want_cov3D_grad=True)
        for it in range(3):  # resident: first call calibrates through the sync path, the rest are no-sync
            img = eng.forward(a["bg"], a["m"], a["c"], a["o"], a["s"], a["r"], a["v"], a["p"], a["cp"], cam.tanfovx, cam.tanfovy).clone()
            eng.backward(dL)
        eng.check()
        torch.cuda.synchronize()
        assert (not resident) or (eng.capacity > eng.R > 0 and eng._last_resident)
        if resident:   # a second backward on the same forward state: the self-cleaning accumulators start from zero again
            first = {k: v.cpu().numpy().copy() for k, v in eng.grads.items()}
            eng.backward(dL)
            for k, v in eng.grads.items():
                assert_grad_close(k, v.cpu().numpy(), first[k])
        outs.append((img.cpu().numpy(), eng.radii.cpu().numpy(), eng.R, eng.dL_dcov3D.cpu().numpy().copy(),
                     {k: v.cpu().numpy().copy() for k, v in eng.grads.items()}))
    (i0, r0, R0, c0, g0), (i1, r1, R1, c1, g1), (i2, r2, R2, c2, g2) = outs
    assert R0 == R1 and 0 < R2 < R0, (R0, R1, R2)
    assert np.array_equal(r0, r1) and np.array_equal(r0, r2)
    assert np.array_equal(i0, i1) and np.array_equal(i0, i2)
    # gradients: float atomics are order-dependent -> tolerance, not bits
    for k in g0:
        assert_grad_close(k, g1[k], g0[k])
        assert_grad_close(k, g2[k], g0[k])


def test_resident_depth_key_range_overflow_falls_back_to_the_exact_path():
    """The resident depth sort covers view depths below 0.2 * 2^16 = 13 107 (27 key bits above the near plane).  A binned
    Gaussian beyond that flags the step like a capacity overflow; the next forward goes through the synchronising path,
    whose sort uses the exact host-known depth range."""
    from segs_slam_amd.raster_engine import RasterEngine
    sc = scenes.make_scene(2000, 160, 96, 140.0, 140.0, seed=23)
    sc.means3D[:5] = [[0.0, 0.0, 20000.0], [300.0, 0.0, 21000.0], [-300.0, 100.0, 15000.0], [0.0, -200.0, 30000.0], [10.0, 10.0, 14000.0]]
    sc.scales[:5] = 400.0
    sc.opacity[:5] = 0.9
    cam = sc.camera
    a = [_t(x) for x in (sc.bg, sc.means3D, sc.colors, sc.opacity, sc.scales, sc.rotations, cam.world_view_transform,
                         cam.full_proj_transform, cam.camera_center)]
    ref = RasterEngine(sc.P, cam.width, cam.height, DEV, resident=False)
    want = ref.forward(*a, cam.tanfovx, cam.tanfovy).clone()
    assert int((ref.radii[:5] > 0).sum()) >= 3          # the far Gaussians are really binned
    eng = RasterEngine(sc.P, cam.width, cam.height, DEV, resident=True)
    img = eng.forward(*a, cam.tanfovx, cam.tanfovy).clone()          # calibrating call: synchronising path
    assert torch.equal(img, want) and eng.check(raise_on_overflow=False)
    eng.forward(*a, cam.tanfovx, cam.tanfovy)                        # resident call: out-of-range depth -> flagged
    assert eng._last_resident and eng.check(raise_on_overflow=False) is False
    img = eng.forward(*a, cam.tanfovx, cam.tanfovy).clone()          # redone through the exact path
    assert not eng._last_resident and torch.equal(img, want)
    # a scene inside the range is not flagged
    sc.means3D[:5, 2] = 5000.0
    a[1] = _t(sc.means3D)
    eng2 = RasterEngine(sc.P, cam.width, cam.height, DEV, resident=True)
    for _ in range(3):
        eng2.forward(*a, cam.tanfovx, cam.tanfovy)
    assert eng2._last_resident and eng2.check(raise_on_overflow=False)


def test_resident_engine_survives_a_shrinking_active_row_count():
    """A map that shrinks inside pre-sized buffers (adjust_anchor pruning rows): the resident geometry buffer is carved up
    for the rows it was ALLOCATED for, so the self-cleaned accumulator rows do not move when P_active changes.  With the
    layout keyed by P_active (the bug) a shrink by more than ~5 % laid the accumulators over stale depth-range words and
    Bin records of the previous layout and the gradients of hundreds of Gaussians were garbage."""
    from segs_slam_amd.raster_engine import RasterEngine
    sc = scenes.make_scene(40_000, 320, 240, 260.0, 260.0, seed=29)
    sc.scales *= 2.0
    cam = sc.camera
    a = [_t(x) for x in (sc.bg, sc.means3D, sc.colors, sc.opacity, sc.scales, sc.rotations, cam.world_view_transform,
                         cam.full_proj_transform, cam.camera_center)]
    dL = _t(sc.dL_dout_color)
    eng = RasterEngine(sc.P, cam.width, cam.height, DEV, resident=True)
    for _ in range(3):                       # full size: calibrate, then resident passes leave their scratch behind
        eng.forward(*a, cam.tanfovx, cam.tanfovy)
        eng.backward(dL)
    assert eng.check() and eng._last_resident
    for frac in (0.93, 0.8, 0.55, 0.97):     # shrink past the 4.4 % and 13 % marks of the old layout, then grow again
        n = int(sc.P * frac)
        eng.set_active(n)
        for _ in range(2):
            img = eng.forward(*a, cam.tanfovx, cam.tanfovy).clone()
            eng.backward(dL)
        assert eng.check() and eng._last_resident
        got = {k: v[:n].cpu().numpy().copy() for k, v in eng.grads.items()}
        ref = RasterEngine(n, cam.width, cam.height, DEV, resident=False)      # reference-shaped path on the first n rows
        want_img = ref.forward(a[0], *[x[:n].contiguous() for x in a[1:6]], *a[6:], cam.tanfovx, cam.tanfovy)
        ref.backward(dL)
        assert torch.equal(img, want_img), frac
        for k, v in ref.grads.items():
            assert_grad_close(f"{k}@{frac}", got[k], v.cpu().numpy())


def test_device_gradients_match_independent_float64_autograd():
    """The device against the INDEPENDENT float64 autograd formulation (oracle/torch_ref.py: written from the behavioural spec,
    not from backward.cu, and not sharing a line with the C++ oracle), at a size where tiles hold hundreds of blended
    Gaussians: 3 000 Gaussians on 160x96.  tests/test_oracle.py checks the C++ oracle against it on 120 Gaussians; here the
    HIP path itself is the one compared, so an error common to the oracle and the kernels (both follow the same derivation of
    K11-K13) could not hide.  Integer decisions (tile rectangles) come from the device's own radii / means2D; dL/dimage is
    zeroed on the pixels whose compositing decisions sit within 3e-3 of a threshold (1.3 % of them here): float32 and float64
    evaluations of the conic differ by up to ~1e-3 relative in the exponent, and with a 1e-4 margin a few flipped decisions
    put errors of 4 % of the largest entry into dL/dopacity -- identically for the device and for the C++ oracle.
    Bar (float32 evaluation against float64 truth): every entry within 1e-4 relative + 2e-5 of the tensor's largest entry
    (measured: 1.5e-6 ... 8.4e-6 of max, the same for the C++ oracle), and 95 % of the non-zero entries within the pure 1e-4
    relative bound (measured 96.8 ... 98.1 %: the float32 conic's own rounding, again the same for the oracle)."""
    from oracle import torch_ref
    sc = scenes.make_scene(3000, 160, 96, 130.0, 130.0, seed=909, bg=(0.1, 0.3, 0.2))
    sc.scales *= 2.0
    sc.dL_dout_color[:] = (scenes.uniform01(sc.dL_dout_color.size, 91, 9).reshape(sc.dL_dout_color.shape) * 2 - 1)
    o, _ = gs_oracle.run_scene(sc, backward=False)
    unstable = o.unstable_pixels(3e-3)
    assert unstable.mean() < 0.05
    dL = sc.dL_dout_color.copy()
    dL[:, unstable] = 0.0
    args, fwd = gpu_forward(sc)
    g = gpu_state(sc, fwd)
    got = gpu_backward(sc, args, fwd, dL)
    cam = sc.camera
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)  # noqa: E731
    m, s, r, op, col = t64(sc.means3D), t64(sc.scales), t64(sc.rotations), t64(sc.opacity), t64(sc.colors)
    img, p_proj = torch_ref.render(m, s, r, op, col, torch.tensor(sc.bg, dtype=torch.float64), torch.tensor(cam.world_view_transform),
                                   torch.tensor(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, cam.height, cam.width,
                                   torch.tensor(g["radii"]), torch.tensor(g["means2D"]), sc.scale_modifier)
    (img * torch.tensor(dL, dtype=torch.float64)).sum().backward()
    assert float(np.abs(g["out_color"] - img.detach().numpy())[:, ~unstable].max()) < 2e-5
    truth = dict(dL_dmean3D=m.grad.numpy(), dL_dscale=s.grad.numpy(), dL_drot=r.grad.numpy(), dL_dopacity=op.grad.numpy(),
                 dL_dcolor=col.grad.numpy(), dL_dmean2D=p_proj.grad.numpy()[:, :2])
    for k, want in truth.items():
        have = got[k].astype(np.float64)
        have = have[:, :2] if k == "dL_dmean2D" else have.reshape(want.shape)
        err = np.abs(have - want)
        top = np.abs(want).max()
        assert top > 0
        assert np.all(err <= 1e-4 * np.abs(want) + 2e-5 * top), (k, float(err.max() / top))
        nz = want != 0
        pure = float((err[nz] <= 1e-4 * np.abs(want[nz])).mean())
        assert pure >= 0.95, (k, pure)


def fuzz_scene(seed):
    """The scene tools/fuzz_raster.py builds for one seed."""
    rng = np.random.default_rng(seed)
    P = int(rng.choice([1, 3, 50, 700, 4000, 20000, 60000]))
    W, H = int(rng.integers(17, 400)), int(rng.integers(17, 300))
    f = float(rng.uniform(0.4, 1.5)) * max(W, H)
    bg = tuple(float(x) for x in rng.choice([0.0, 0.5, 1.0], size=3))
    sc = scenes.make_scene(P, W, H, f, f, seed=seed, bg=bg)
    sc.scales *= float(rng.choice([0.3, 1.0, 3.0, 10.0]))
    if rng.random() < 0.3:
        sc.opacity[:] = (sc.opacity * float(rng.choice([0.02, 0.2]))).astype(np.float32)
    return sc


@pytest.mark.parametrize("seed", [21117, 41105, 51072])
def test_extreme_overdraw_cases_pass_under_the_float64_arbiter_rule(seed):
    """The three cases of round 2's fuzz sweeps (profiles/r02_fuzz_summary.txt) that fell under the 99 % pure-relative bar
    against the float32 oracle -- tens of thousands of Gaussians at 10x scale on a ~150x60 image, a few hundred visible, each
    covering thousands of pixels -- as tests under rule (3) of assert_grad_close instead of accepted failures of a tool."""
    sc = fuzz_scene(seed)
    o, _ = run_parity(sc, backward=True)
    assert all(v == "pure" or v.startswith("arbiter") for v in o.grad_verdicts.values()), o.grad_verdicts


def test_tiles_touched_by_gather_path_gives_the_same_binning():
    """The build carries tiles_touched into depth order in the spare bits of the depth sort's values; above 2^26 Gaussians (fewer
    than six spare bits) the last sort pass gathers it instead.  SEGS_RASTER_GATHER_TILES_TOUCHED forces that path: binning and
    image must not change.  Includes Gaussians whose tile count saturates the packed field (a 640x480 image has 1200 tiles, a
    4000-Gaussian scene leaves 20 spare bits, so saturation is forced here by scale: every visible Gaussian covers the image)."""
    from segs_slam_amd import _capi
    sc = scenes.make_scene(4000, 208, 144, 150.0, 150.0, seed=515, bg=(0.1, 0.1, 0.1))
    sc.scales *= 3.0
    outs = []
    for flags in (0, 4, 8):   # packed (default), gathered by the last pass, packed into two bits (counts above 2 saturate and are fetched)
        old = _capi.lib().segs_raster_set_flags(flags)
        try:
            args, fwd = gpu_forward(sc)
            outs.append(gpu_state(sc, fwd))
        finally:
            _capi.lib().segs_raster_set_flags(old)
    assert (outs[0]["tiles_touched"] > 2).mean() > 0.2
    for other in outs[1:]:
        for k in ("keys", "point_list", "ranges", "n_contrib", "out_color", "tiles_touched"):
            assert np.array_equal(outs[0][k], other[k]), k


def test_range_table_from_the_last_sort_pass_equals_the_range_kernel():
    """With two tile-id passes the last scatter pass fills the range table and the resident status words itself (two integer
    atomics per (sort tile, tile id) pair, radix_scatter_kernel) and does not store the sorted keys nobody reads.
    SEGS_RASTER_UNFUSED_BINNING runs identify_tile_ranges_kernel instead: every list, the range table, n_contrib and the image
    must be identical -- through the reference-shaped call (full lists, sorted keys kept) and through the resident engine
    (dead instances dropped by the first pass, status words), including a map that does not fill the engine's rows."""
    from segs_slam_amd import _capi
    from segs_slam_amd.raster_engine import RasterEngine
    sc = scenes.make_scene(150_000, 1200, 680, 600.0, 600.0, seed=909, bg=(0.0, 0.1, 0.0))
    sc.scales *= 2.5
    cam = sc.camera
    outs = []
    for flags in (0, 16):
        old = _capi.lib().segs_raster_set_flags(flags)
        try:
            args, fwd = gpu_forward(sc)
            outs.append(gpu_state(sc, fwd))
            outs[-1]["R"] = fwd[0]
        finally:
            _capi.lib().segs_raster_set_flags(old)
    assert outs[0]["R"] > 257 * 2048, outs[0]["R"]      # two 8-bit tile-id passes, count chunks of two sort tiles
    for k in ("keys", "point_list", "ranges", "n_contrib", "out_color", "tiles_touched"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    a = dict(bg=_t(sc.bg), m=_t(sc.means3D), c=_t(sc.colors), o=_t(sc.opacity), s=_t(sc.scales), r=_t(sc.rotations),
             v=_t(cam.world_view_transform), p=_t(cam.full_proj_transform), cp=_t(cam.camera_center))
    dL = _t(sc.dL_dout_color)
    res = []
    for extra in (0, 16):
        for active in (sc.P, sc.P - 37_111):
            eng = RasterEngine(sc.P, cam.width, cam.height, DEV, resident=True)
            eng.flags |= extra
            eng.set_active(active)
            sl = slice(0, active)
            for _ in range(3):
                img = eng.forward(a["bg"], a["m"][sl].contiguous(), a["c"][sl].contiguous(), a["o"][sl].contiguous(),
                                  a["s"][sl].contiguous(), a["r"][sl].contiguous(), a["v"], a["p"], a["cp"], cam.tanfovx, cam.tanfovy)
                eng.backward(dL)
            assert eng.check() and eng._last_resident
            torch.cuda.synchronize()
            st = _capi_state(eng, cam)
            res.append((extra, active, img.cpu().numpy().copy(), eng.R, eng.R_live, st,
                        {k: v.cpu().numpy().copy() for k, v in eng.grads.items()}))
    for i in (0, 1):
        f, u = res[i], res[2 + i]
        assert f[1] == u[1] and f[3] == u[3] and f[4] == u[4] and 0 < f[4] < f[3]
        assert np.array_equal(f[2], u[2])
        for k in ("ranges", "n_contrib", "values"):
            assert np.array_equal(f[5][k], u[5][k]), k
        for k in f[6]:
            assert_grad_close(k, f[6][k][:f[1]], u[6][k][:f[1]])
    assert np.array_equal(res[0][2], outs[0]["out_color"])


def _capi_state(eng, cam):
    """ranges / n_contrib / sorted instance values of a resident engine's last forward."""
    import ctypes as C
    from segs_slam_amd import _capi
    lib = _capi.lib()
    tiles = ((cam.width + 15) // 16) * ((cam.height + 15) // 16)
    ranges = torch.zeros((tiles, 2), dtype=torch.int32, device=DEV)
    ncontrib = torch.zeros((cam.height, cam.width), dtype=torch.int32, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    _capi.check(lib.segs_debug_unpack_image(C.c_void_p(eng._img_r.data_ptr()), cam.width, cam.height, C.c_void_p(ranges.data_ptr()),
                                            None, C.c_void_p(ncontrib.data_ptr()), st), "segs_debug_unpack_image")
    vals = torch.zeros(eng.capacity, dtype=torch.int32, device=DEV)
    _capi.check(lib.segs_debug_instance_values(C.c_void_p(eng._bin_r.data_ptr()), eng.capacity, C.c_void_p(vals.data_ptr()), st),
                "segs_debug_instance_values")
    torch.cuda.synchronize()
    return dict(ranges=ranges.cpu().numpy(), n_contrib=ncontrib.cpu().numpy(), values=vals.cpu().numpy()[:eng.R_live])
