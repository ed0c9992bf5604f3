"""Independent differentiable formulation of the rasterizer (PyTorch, float64, CPU).  TEST ONLY.

Written from the behavioural spec (projection, EWA covariance, alpha
compositing), NOT from the reference's backward.cu -- its autograd gradients are the independent
check of the oracle's analytic backward (K11-K13).  Usable only for small P / small images:
it loops over Gaussians in Python and holds a (pixels x P) alpha matrix.

Non-differentiable decisions (which tiles a Gaussian is binned to) are taken from integer data
(`radii`, `means2D`) supplied by the caller, so that only the differentiable math is under test.
"""
from __future__ import annotations

import torch


SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def sh_to_rgb(means3D, campos, sh, deg):
    """Real spherical harmonics up to degree 3 evaluated in the view direction, +0.5, clamped at 0 (written from the
    standard SH basis; the clamp's zero gradient is torch.clamp's)."""
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp(res + 0.5, min=0.0)


def render(means3D, scales, rotations, opacity, colors, bg, viewmatrix, projmatrix, tanfovx, tanfovy, H, W,
           radii, means2D_ref, scale_modifier=1.0):
    """All tensor args float64 (requires_grad where wanted); radii int tensor (P,), means2D_ref (P,2).
    Returns (image (3,H,W), p_proj (P,3) -- retain_grad()'ed so that .grad[:, :2] is dL/dmean2D in the
    reference's NDC-scaled convention, cf. backward.cu:545-546)."""
    dt = torch.float64
    V = viewmatrix.to(dt)   # transposed layout: x' = V[0,0]x + V[1,0]y + V[2,0]z + V[3,0]
    PV = projmatrix.to(dt)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], dim=1)
    p_view = hom @ V[:, :3]
    p_hom = hom @ PV
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :3] * p_w[:, None]
    p_proj.retain_grad()

    # Sigma = R S^2 R^T with R from the un-normalised quaternion (r,x,y,z)
    r, x, y, z = rotations.unbind(1)
    Rm = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(P, 3, 3)
    S = scale_modifier * scales
    Sigma = Rm @ torch.diag_embed(S * S) @ Rm.transpose(1, 2)

    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    # Clamp to 1.3x the frustum.  The reference treats the clamped t.x / t.y as constants in its
    # backward (x_grad_mul / y_grad_mul = 0 and no extra t.z term, backward.cu:175-176,246-248), which is
    # what detach() expresses here; unclamped values pass through unchanged.
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    tx = torch.where((txtz < -limx) | (txtz > limx), (torch.clamp(txtz, -limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (torch.clamp(tytz, -limy, limy) * tz).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], dim=1).reshape(P, 2, 3)
    Wr = V[:3, :3].T  # rotation part of the (untransposed) view matrix
    T = J @ Wr
    cov = T @ Sigma @ T.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conA, conB, conC = c / det, -b / det, a / det

    pixx = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    pixy = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5

    # tile membership from integer data
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rad = radii.to(torch.float32)
    m2 = means2D_ref.to(torch.float32)
    rminx = torch.clamp(((m2[:, 0] - rad) / 16).to(torch.int32), 0, gx)
    rminy = torch.clamp(((m2[:, 1] - rad) / 16).to(torch.int32), 0, gy)
    rmaxx = torch.clamp(((m2[:, 0] + rad + 15) / 16).to(torch.int32), 0, gx)
    rmaxy = torch.clamp(((m2[:, 1] + rad + 15) / 16).to(torch.int32), 0, gy)
    visible = radii > 0

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    px = xs.reshape(-1).to(dt)
    py = ys.reshape(-1).to(dt)
    tpx = (xs.reshape(-1) // 16).to(torch.int32)
    tpy = (ys.reshape(-1) // 16).to(torch.int32)

    depth32 = p_view[:, 2].detach().to(torch.float32)
    order = sorted([i for i in range(P) if bool(visible[i])], key=lambda i: (float(depth32[i]), i))

    npix = H * W
    Tcur = torch.ones(npix, dtype=dt)
    C = torch.zeros(npix, 3, dtype=dt)
    alive = torch.ones(npix, dtype=torch.bool)
    for i in order:
        inrect = (tpx >= rminx[i]) & (tpx < rmaxx[i]) & (tpy >= rminy[i]) & (tpy < rmaxy[i])
        dx = pixx[i] - px
        dy = pixy[i] - py
        power = -0.5 * (conA[i] * dx * dx + conC[i] * dy * dy) - conB[i] * dx * dy
        alpha = torch.clamp(opacity[i, 0] * torch.exp(power), max=0.99)
        use = alive & inrect & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = Tcur * (1 - alpha)
        stop = use & (test_T < 1e-4)
        alive = alive & ~stop
        use = use & ~stop
        w = torch.where(use, alpha * Tcur, torch.zeros_like(alpha))
        C = C + w[:, None] * colors[i][None, :]
        Tcur = torch.where(use, test_T, Tcur)
    img = C + Tcur[:, None] * bg.to(dt)[None, :]
    return img.T.reshape(3, H, W), p_proj
