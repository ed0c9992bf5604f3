"""Spherical-harmonics colour branch (forward.cu:20-71, backward.cu:20-139): off the live SEGS-SLAM path, kept for
API parity.  Oracle vs independent autograd on CPU; HIP vs oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle import gs_oracle, torch_ref
from segs_slam_amd import scenes


def _scene(P, deg, seed):
    sc = scenes.make_scene(P, 64, 48, 55.0, 55.0, seed=seed, bg=(0.05, 0.1, 0.15))
    sc.scales *= 3.5
    M = 16
    sh = ((scenes.uniform01(P * M * 3, 30, seed).reshape(P, M, 3) - 0.5) * 1.2).astype(np.float32)
    sh[:, 0] += 0.4
    sh[::4, 0, :2] = -2.5  # every 4th Gaussian: red/green go negative before the clamp (clamped flags, zero gradient)
    dL = (scenes.uniform01(sc.dL_dout_color.size, 31, seed).reshape(sc.dL_dout_color.shape) * 2 - 1).astype(np.float32)
    return sc, sh, dL


def _oracle(sc, sh, deg):
    cam = sc.camera
    o = gs_oracle.Oracle()
    o.forward(sc.bg, sc.means3D, None, sc.opacity, sc.scales, 1.0, sc.rotations, cam.world_view_transform,
              cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.height, cam.width, sh=sh, degree=deg,
              campos=cam.camera_center)
    return o


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_oracle_matches_autograd(deg):
    sc, sh, dL = _scene(90, deg, 240 + 7 * deg)
    cam = sc.camera
    o = _oracle(sc, sh, deg)
    unstable = o.unstable_pixels(1e-4)  # float32-vs-float64 threshold decisions may differ there: excluded on both sides
    assert unstable.mean() < 0.01
    dL[:, unstable] = 0
    g = o.backward(dL)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)  # noqa: E731
    m, s, r, op, tsh = t(sc.means3D), t(sc.scales), t(sc.rotations), t(sc.opacity), t(sh)
    col = torch_ref.sh_to_rgb(m, torch.tensor(cam.camera_center, dtype=torch.float64), tsh, deg)
    assert np.abs(col.detach().numpy() - o.get("rgb"))[o.get("radii") > 0].max() < 1e-6
    assert (col == 0).any()  # the clamp is exercised
    img, _ = torch_ref.render(m, s, r, op, col, torch.tensor(sc.bg, dtype=torch.float64), torch.tensor(cam.world_view_transform),
                              torch.tensor(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, cam.height, cam.width,
                              torch.tensor(o.get("radii")), torch.tensor(o.get("means2D")))
    (img * torch.tensor(dL, dtype=torch.float64)).sum().backward()
    assert np.abs(img.detach().numpy() - o.get("out_color"))[:, ~unstable].max() < 2e-5
    for key, ref in (("dL_dsh", tsh.grad.numpy()), ("dL_dmean3D", m.grad.numpy())):
        assert np.abs(g[key] - ref).max() / (np.abs(ref).max() + 1e-30) < 2e-4, key
    nact = (deg + 1) ** 2
    assert np.all(g["dL_dsh"][:, nact:] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("deg", [0, 3])
def test_sh_gpu_matches_oracle(deg):
    from segs_slam_amd import rasterize_points as rp
    DEV = "cuda:0"
    sc, sh, dL = _scene(4000, deg, 50 + deg)
    cam = sc.camera
    o = _oracle(sc, sh, deg)
    unstable = o.unstable_pixels(1e-5)
    dL[:, unstable] = 0
    ref = o.backward(dL)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    e = torch.empty(0, device=DEV)
    bg, m3, op, sca, rot, tsh = t(sc.bg), t(sc.means3D), t(sc.opacity), t(sc.scales), t(sc.rotations), t(sh)
    view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
    R, color, radii, geom, binning, img = rp.RasterizeGaussiansCUDA(bg, m3, e, op, sca, rot, 1.0, e, view, proj, cam.tanfovx,
                                                                    cam.tanfovy, cam.height, cam.width, tsh, deg, campos, False)
    assert R == o.R and np.array_equal(radii.cpu().numpy(), o.get("radii"))
    st = rp.debug_state(sc.P, cam.width, cam.height, R, radii, geom, binning, img)
    vis = o.get("radii") > 0
    # SH -> RGB: the device evaluates a basis table and a dot product (sh_color.h), the oracle forward.cu's running sum
    np.testing.assert_allclose(st["rgb"].cpu().numpy()[vis], o.get("rgb")[vis], rtol=2e-6, atol=2e-6)
    ok = ~unstable
    a, b = color.cpu().numpy(), o.get("out_color")
    assert np.all(np.abs(a - b)[:, ok] <= 1e-4 * np.abs(b)[:, ok] + 2e-6)
    grads = rp.RasterizeGaussiansBackwardCUDA(bg, m3, radii, e, sca, rot, 1.0, e, view, proj, cam.tanfovx, cam.tanfovy, t(dL), tsh,
                                              deg, campos, geom, R, binning, img)
    for got, key in ((grads[5], "dL_dsh"), (grads[3], "dL_dmean3D"), (grads[6], "dL_dscale"), (grads[2], "dL_dopacity")):
        g, r = got.cpu().numpy(), ref[key]
        assert np.all(np.abs(g - r) <= 1e-4 * np.abs(r) + 1e-5 * np.abs(r).max()), key   # floor: see tests/test_raster_gpu.py


@pytest.mark.gpu
def test_project2_image_matches_oracle():
    """RasterizeGaussiansprojectCUDA (src/rasterize_points.cu:282-360 -> Rasterizer::project2_image,
    rasterizer_impl.cu:494-585 -> projectCUDA, forward.cu:573-673): per-Gaussian pixel position, radius and colour.  The
    reference leaves rows of rejected Gaussians uninitialised, so only radii > 0 rows are compared (bit-exact)."""
    from segs_slam_amd import rasterize_points as rp
    DEV = "cuda:0"
    sc, sh, _ = _scene(5000, 2, 91)
    cam = sc.camera
    o = _oracle(sc, sh, 2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    e = torch.empty(0, device=DEV)
    bg, m3, op, sca, rot, tsh = t(sc.bg), t(sc.means3D), t(sc.opacity), t(sc.scales), t(sc.rotations), t(sh)
    view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
    pts, radii, col = rp.RasterizeGaussiansprojectCUDA(bg, m3, e, op, sca, rot, 1.0, e, view, proj, cam.tanfovx, cam.tanfovy,
                                                       cam.height, cam.width, tsh, 2, campos, False)
    vis = o.get("radii") > 0
    assert vis.sum() > 1000 and np.array_equal(radii.cpu().numpy(), o.get("radii"))
    assert np.array_equal(pts.cpu().numpy()[vis].view(np.uint32), o.get("means2D")[vis].view(np.uint32))
    np.testing.assert_allclose(col.cpu().numpy()[vis], o.get("rgb")[vis], rtol=2e-6, atol=2e-6)
    # precomputed colours: positions and radii are unchanged
    pts2, radii2, _ = rp.RasterizeGaussiansprojectCUDA(bg, m3, t(sc.colors), op, sca, rot, 1.0, e, view, proj, cam.tanfovx,
                                                       cam.tanfovy, cam.height, cam.width, e, 0, campos, False)
    assert torch.equal(radii2, radii) and torch.equal(pts2[t(vis)], pts[t(vis)])
    with pytest.raises(RuntimeError):
        rp.RasterizeGaussiansprojectCUDA(bg, torch.zeros(4, 2, device=DEV), e, e, e, e, 1.0, e, view, proj, 1.0, 1.0, 8, 8, e, 0, campos, False)
