"""simple-knn / operate_points / stereo_vision: oracle sanity on CPU, bit-exact parity on the GPU."""
import numpy as np
import pytest
import torch

from oracle import gs_oracle

DEV = "cuda:0"


def _pts(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, 3)) * np.array([2.0, 1.0, 0.5])).astype(np.float32)


def test_knn_oracle_against_kdtree():
    from scipy.spatial import cKDTree
    p = _pts(3000, 1)
    got = gs_oracle.knn_mean_dist2(p)
    d, _ = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=4)
    ref = (d[:, 1:] ** 2).mean(axis=1)
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-9)
    # duplicates: a duplicated point has nearest distance 0
    q = np.concatenate([p[:10], p[:10]])
    assert np.all(gs_oracle.knn_mean_dist2(q)[:10] <= gs_oracle.knn_mean_dist2(p[:10]) + 1e-6)


def test_quaternion_transform_oracle_is_rotation_composition():
    rng = np.random.default_rng(2)
    q = rng.standard_normal((50, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    ang = 0.7
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=np.float32)
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] = Rz
    M[:3, 3] = [0.1, -0.2, 0.3]
    Mt = np.ascontiguousarray(M.T)  # transposed layout, auxiliary.h:59-67
    p = _pts(50, 3)
    out_p, out_r = gs_oracle.scale_and_transform_points(p, q, Mt, np.ones(50, np.uint8), 2.0)
    assert np.allclose(out_p, (2.0 * p) @ Rz.T + M[:3, 3], atol=1e-5)
    # (w, x) of the composed rotation are right; slot 2 holds z and slot 3 stays 0 (the reference's insert_rot_to_rots quirk)
    def quat_to_R(w, x, y, z):
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    for i in range(50):
        Rn = Rz @ quat_to_R(*q[i])
        w = 0.5 * np.sqrt(max(1e-12, 1 + np.trace(Rn)))
        if np.trace(Rn) > 0:
            assert abs(out_r[i, 0] - w) < 1e-4
            assert abs(out_r[i, 1] - (Rn[2, 1] - Rn[1, 2]) / (4 * w)) < 1e-4
            assert abs(out_r[i, 2] - (Rn[1, 0] - Rn[0, 1]) / (4 * w)) < 1e-4
        assert out_r[i, 3] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 1000, 20000])
def test_dist2_gpu_bit_exact(n):
    from segs_slam_amd import points as sp
    p = _pts(n, n)
    got = sp.distCUDA2(torch.from_numpy(p).to(DEV)).cpu().numpy()
    ref = gs_oracle.knn_mean_dist2(p)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_operate_points_gpu_bit_exact():
    from segs_slam_amd import points as sp
    from segs_slam_amd import scenes
    sc = scenes.make_scene(5000, 320, 240, 250.0, 250.0, seed=8)
    cam = sc.camera
    rng = np.random.default_rng(4)
    A = rng.standard_normal((3, 3))
    Q, _ = np.linalg.qr(A)
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] = Q.astype(np.float32)
    M[:3, 3] = [0.05, 0.02, -0.03]
    Mt = np.ascontiguousarray(M.T)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    out = sp.transformPoints(t(sc.means3D), t(Mt)).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), gs_oracle.transform_points(sc.means3D, Mt).view(np.uint32))
    # scaleAndTransformThenMarkVisiblePoints
    ntm = rng.random(sc.P) < 0.7
    unst = rng.random(sc.P) < 0.8
    pts, rots, m1 = t(sc.means3D), t(sc.rotations), torch.from_numpy(ntm).to(DEV)
    n = sp.scaleAndTransformThenMarkVisiblePoints(pts, rots, m1, torch.from_numpy(unst).to(DEV), t(Mt),
                                                  t(cam.world_view_transform), t(cam.full_proj_transform), 3, 1.5)
    present = gs_oracle.mark_visible(sc.means3D, cam.world_view_transform, cam.full_proj_transform)
    final = ntm & unst & present
    assert n == 3 + int(final.sum())
    rp, rr = gs_oracle.scale_and_transform_points(sc.means3D, sc.rotations, Mt, final.astype(np.uint8), 1.5)
    exp_p, exp_r = sc.means3D.copy(), sc.rotations.copy()
    exp_p[final], exp_r[final] = rp[final], rr[final]
    assert np.array_equal(pts.cpu().numpy().view(np.uint32), exp_p.view(np.uint32))
    assert np.array_equal(rots.cpu().numpy().view(np.uint32), exp_r.view(np.uint32))
    assert np.array_equal(m1.cpu().numpy(), ntm & ~final)


@pytest.mark.gpu
def test_stereo_vision_gpu_bit_exact():
    from segs_slam_amd import points as sp
    rng = np.random.default_rng(9)
    W, H = 64, 48
    depth = (rng.random(W * H) * 5 + 0.2).astype(np.float32)
    mask = rng.random(W * H) < 0.6
    intr = [60.5, 61.25, 31.7, 23.9]
    got = sp.reprojectDepthPinhole(torch.from_numpy(depth).to(DEV), torch.from_numpy(mask).to(DEV), intr, W).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), gs_oracle.reproject_depths_pinhole(depth, mask, intr, W).view(np.uint32))
    N = 700
    px = np.stack([rng.integers(0, W, N), rng.integers(0, H, N)], axis=1).astype(np.float32)
    has3d = rng.random(N) < 0.4
    p3 = np.concatenate([rng.standard_normal((N, 2)), rng.random((N, 1)) * 4 + 0.3], axis=1).astype(np.float32)
    colors = rng.random(W * H + 3).astype(np.float32)
    rp, rc = sp.monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints(
        torch.from_numpy(px).to(DEV), torch.from_numpy(has3d).to(DEV), torch.from_numpy(p3).to(DEV),
        torch.from_numpy(colors).to(DEV), 3.0, intr, W)
    op, oc = gs_oracle.search_neighborhood_depth(px, has3d, p3, colors, 3.0, intr, W)
    valid = op[:, 2] > 0
    assert np.array_equal(rp.cpu().numpy().view(np.uint32), op[valid].view(np.uint32))
    assert np.array_equal(rc.cpu().numpy().view(np.uint32), oc[valid].view(np.uint32))
    assert 0 < valid.sum() < N
