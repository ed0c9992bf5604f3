"""Float64 evaluation of the per-Gaussian backward stage (K12 + K13 + cov3D backward) from its defining equations.  TEST ONLY.

Written from the forward map: t = Wv p + tv with the 1.3 tan(fov/2) clamp, J, M2 = J Wv,
L = R(q) diag(mod * scale), C = (M2 L)(M2 L)^T + 0.3 I, conic = C^-1, and the chain rule in matrix form --
dL/dC = -k adj(C) G adj(C) with the reference's guarded k = 1 / (det^2 + 1e-7) (backward.cu:205), a clamped coordinate
passing no gradient (:175-176), dL/dscale taken with respect to the modified scale and the quaternion used as given
(:316-320, :340).  It is the arbiter between two float32 evaluations: the HIP kernel's (same matrix form) and the CPU oracle's
(backward.cu's expression order) -- both must agree with it to float32 rounding.

Only tests/ may import this module.
"""
import numpy as np


def stage_f64(m3, sc, rot, view, proj, W, H, tx, ty, g2, gc, radii, mod=1.0):
    f = np.float64
    m3, sc, rot, view, proj, g2, gc = (a.astype(f) for a in (m3, sc, rot, view, proj, g2, gc))
    hx_, hy_ = W / (2 * tx), H / (2 * ty)
    V = view.reshape(4, 4)      # V[j, i] = view[4 j + i]
    Wv = V[:3, :3].T            # Wv[i][j] = view[4j+i]
    t = m3 @ V[:3, :3] + V[3, :3]
    limx, limy = 1.3 * tx, 1.3 * ty
    rx, ry = t[:, 0] / t[:, 2], t[:, 1] / t[:, 2]
    fx, fy = ~((rx < -limx) | (rx > limx)), ~((ry < -limy) | (ry > limy))
    txc, tyc, tz = np.clip(rx, -limx, limx) * t[:, 2], np.clip(ry, -limy, limy) * t[:, 2], t[:, 2]
    P = m3.shape[0]
    J = np.zeros((P, 2, 3)); J[:, 0, 0] = hx_ / tz; J[:, 0, 2] = -hx_ * txc / tz**2; J[:, 1, 1] = hy_ / tz; J[:, 1, 2] = -hy_ * tyc / tz**2
    M2 = J @ Wv
    r, x, y, z = rot.T
    R = np.stack([np.stack([1 - 2 * (y*y + z*z), 2 * (x*y - r*z), 2 * (x*z + r*y)], 1),
                  np.stack([2 * (x*y + r*z), 1 - 2 * (x*x + z*z), 2 * (y*z - r*x)], 1),
                  np.stack([2 * (x*z - r*y), 2 * (y*z + r*x), 1 - 2 * (x*x + y*y)], 1)], 1)
    s = mod * sc
    L = R * s[:, None, :]
    U = M2 @ L
    Cm = U @ U.transpose(0, 2, 1); a = Cm[:, 0, 0] + 0.3; b = Cm[:, 0, 1]; c = Cm[:, 1, 1] + 0.3
    det = a * c - b * b; k = 1 / (det**2 + 1e-7)
    G = np.zeros((P, 2, 2)); G[:, 0, 0] = gc[:, 0, 0]; G[:, 0, 1] = G[:, 1, 0] = gc[:, 0, 1]; G[:, 1, 1] = gc[:, 1, 1]
    adj = np.zeros((P, 2, 2)); adj[:, 0, 0] = c; adj[:, 0, 1] = adj[:, 1, 0] = -b; adj[:, 1, 1] = a
    Dc = -k[:, None, None] * (adj @ G @ adj)
    S3 = M2.transpose(0, 2, 1) @ Dc @ M2
    cov = np.stack([S3[:, 0, 0], 2 * S3[:, 0, 1], 2 * S3[:, 0, 2], S3[:, 1, 1], 2 * S3[:, 1, 2], S3[:, 2, 2]], 1)
    dM2 = 2 * Dc @ (U @ L.transpose(0, 2, 1))
    dJ = dM2 @ Wv.T
    dtx = np.where(fx, -hx_ / tz**2 * dJ[:, 0, 2], 0.0); dty = np.where(fy, -hy_ / tz**2 * dJ[:, 1, 2], 0.0)
    dtz = -hx_ / tz**2 * dJ[:, 0, 0] - hy_ / tz**2 * dJ[:, 1, 1] + 2 * hx_ * txc / tz**3 * dJ[:, 0, 2] + 2 * hy_ * tyc / tz**3 * dJ[:, 1, 2]
    dt = np.stack([dtx, dty, dtz], 1)
    PV = proj.reshape(4, 4)
    hom = m3 @ PV[:3, :] + PV[3, :]
    w = 1 / (hom[:, 3] + 1e-7)
    dmean = dt @ Wv
    for j in range(3):
        dmean[:, j] += (PV[j, 0] * w - PV[j, 3] * hom[:, 0] * w * w) * g2[:, 0] + (PV[j, 1] * w - PV[j, 3] * hom[:, 1] * w * w) * g2[:, 1]
    dL = 2 * M2.transpose(0, 2, 1) @ (Dc @ U)
    dscale = (R * dL).sum(1)
    GR = dL * s[:, None, :]
    g = GR
    dq = np.stack([2 * (z * (g[:, 1, 0] - g[:, 0, 1]) + y * (g[:, 0, 2] - g[:, 2, 0]) + x * (g[:, 2, 1] - g[:, 1, 2])),
                   2 * (y * (g[:, 0, 1] + g[:, 1, 0]) + z * (g[:, 0, 2] + g[:, 2, 0]) + r * (g[:, 2, 1] - g[:, 1, 2])) - 4 * x * (g[:, 1, 1] + g[:, 2, 2]),
                   2 * (x * (g[:, 0, 1] + g[:, 1, 0]) + r * (g[:, 0, 2] - g[:, 2, 0]) + z * (g[:, 1, 2] + g[:, 2, 1])) - 4 * y * (g[:, 0, 0] + g[:, 2, 2]),
                   2 * (r * (g[:, 1, 0] - g[:, 0, 1]) + x * (g[:, 0, 2] + g[:, 2, 0]) + y * (g[:, 1, 2] + g[:, 2, 1])) - 4 * z * (g[:, 0, 0] + g[:, 1, 1])], 1)
    vis = radii > 0
    out = {"dL_dmean3D": dmean, "dL_dcov3D": cov, "dL_dscale": dscale, "dL_drot": dq}
    for v in out.values():
        v[~vis] = 0
    return out

