"""Seeded synthetic scenes and camera tensors for the rasterizer hot path.

Datasets are not available offline, so every BASELINE.json config is a seeded synthetic
restatement.  The RNG is a counter-based splitmix64 so the same scene can be
regenerated bit-identically anywhere (numpy only, no torch RNG).

Camera tensors follow the reference's construction (src/gaussian_keyframe.cpp:151-184 for the
transposed layout, :229-249 getWorld2View2, :251-279 getProjectionMatrix; FoV from
include/graphics_utils.h:48-51): the 4x4 tensors handed to the kernels hold the TRANSPOSED
matrices, full_proj = view^T-layout @ proj^T-layout, campos = inverse(view)[3, :3].
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(counter: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 output for state = seed + (counter+1)*golden (vectorised, wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (counter.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(n: int, stream: int, seed: int) -> np.ndarray:
    """n float32 values in [0,1) from stream `stream` (24 random mantissa bits)."""
    ctr = np.arange(n, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40))
    z = splitmix64(ctr, seed)
    return ((z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def fov2focal(fov: float, pixels: int) -> float:  # include/graphics_utils.h:43-46
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal: float, pixels: int) -> float:  # include/graphics_utils.h:48-51
    return 2.0 * math.atan(pixels / (2.0 * focal))


def world2view2(R: np.ndarray, t: np.ndarray, trans=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """src/gaussian_keyframe.cpp:229-249 -- [R t; 0 1] after the camera-centre shift/scale."""
    Rt = np.zeros((4, 4), dtype=np.float32)
    Rt[:3, :3] = R
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    c = (C2W[:3, 3] + np.asarray(trans, dtype=np.float32)) * np.float32(scale)
    C2W[:3, 3] = c
    return np.linalg.inv(C2W).astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """src/gaussian_keyframe.cpp:251-279 (untransposed P; principal point ignored)."""
    f32 = np.float32
    tan_y = f32(math.tan(fovy / 2))
    tan_x = f32(math.tan(fovx / 2))
    top = tan_y * f32(znear)
    bottom = -top
    right = tan_x * f32(znear)
    left = -right
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = f32(zfar) / (f32(zfar) - f32(znear))
    P[2, 3] = -(f32(zfar) * f32(znear)) / (f32(zfar) - f32(znear))
    return P


@dataclass
class Camera:
    """What GaussianKeyframe::computeTransformTensors produces (src/gaussian_keyframe.cpp:151-184)."""
    width: int
    height: int
    fovx: float
    fovy: float
    world_view_transform: np.ndarray  # (4,4) float32, transposed layout
    projection_matrix: np.ndarray     # (4,4) float32, transposed layout
    full_proj_transform: np.ndarray   # (4,4)
    camera_center: np.ndarray         # (3,)

    @property
    def tanfovx(self) -> float:
        return math.tan(self.fovx * 0.5)

    @property
    def tanfovy(self) -> float:
        return math.tan(self.fovy * 0.5)


def make_camera(width: int, height: int, fx: float, fy: float, R: np.ndarray, t: np.ndarray,
                znear: float = 0.01, zfar: float = 100.0) -> Camera:
    fovx = focal2fov(fx, width)
    fovy = focal2fov(fy, height)
    wvt = np.ascontiguousarray(world2view2(R, t).T)
    proj = np.ascontiguousarray(projection_matrix(znear, zfar, fovx, fovy).T)
    full = (wvt @ proj).astype(np.float32)
    center = np.linalg.inv(wvt)[3, :3].astype(np.float32)
    return Camera(width, height, fovx, fovy, wvt, proj, np.ascontiguousarray(full), np.ascontiguousarray(center))


@dataclass
class Scene:
    name: str
    camera: Camera
    means3D: np.ndarray    # (P,3)
    scales: np.ndarray     # (P,3)
    rotations: np.ndarray  # (P,4) wxyz, normalised by the generator (the caller passes F::normalize'd rotations)
    opacity: np.ndarray    # (P,1)
    colors: np.ndarray     # (P,3)
    bg: np.ndarray         # (3,)
    dL_dout_color: np.ndarray  # (3,H,W)
    scale_modifier: float = 1.0
    meta: dict = field(default_factory=dict)

    @property
    def P(self) -> int:
        return int(self.means3D.shape[0])


# name -> (P, W, H, fx, fy); sizes per SURVEY.md section 8d / BASELINE.json configs.
CONFIGS = {
    "c1": (50_000, 640, 480, 525.0, 525.0),            # config 1: 50k, 640x480
    "c2": (500_000, 1200, 680, 600.0, 600.0),          # config 2: Replica office0 camera
    "c2_1080p": (500_000, 1920, 1080, 960.0, 960.0),   # config 2 at the metric's 1080p
    "c4": (200_000, 640, 480, 535.4, 539.2),           # config 4: TUM fr3 camera, one keyframe per GPU
    "c5": (3_000_000, 1200, 680, 600.0, 600.0),        # config 5: ~3M Gaussians
    "1080p_1m": (1_000_000, 1920, 1080, 960.0, 960.0),
    "1080p_2m": (2_000_000, 1920, 1080, 960.0, 960.0),
    "1080p_3m": (3_000_000, 1920, 1080, 960.0, 960.0),
}


def _small_rotation(seed: int, max_deg: float = 5.0) -> np.ndarray:
    u = uniform01(4, 900, seed).astype(np.float64)
    axis = np.array([u[0] - 0.5, u[1] - 0.5, u[2] - 0.5])
    axis /= np.linalg.norm(axis) + 1e-12
    ang = math.radians(max_deg) * u[3]
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
    return R.astype(np.float32)


def keyframe_camera(width: int, height: int, fx: float, fy: float, seed: int = 0x5E65, keyframe: int = 0) -> Camera:
    """The camera of keyframe `keyframe` of a seeded scene: a rotation of at most 5 degrees and (keyframe > 0) a shift of at most
    0.1 m per axis around the scene's camera at the origin -- the poses of the synthetic keyframe orbit."""
    R = _small_rotation(seed + 7919 * keyframe)
    t = np.zeros(3, dtype=np.float32)
    if keyframe:
        t = ((uniform01(3, 901, seed + 7919 * keyframe) - 0.5) * 0.2).astype(np.float32)
    return make_camera(width, height, fx, fy, R, t)


def make_config_camera(name: str, keyframe: int = 0, seed_offset: int = 0) -> Camera:
    """Camera of make_config_scene(name, keyframe=keyframe) without generating its Gaussians."""
    P, W, H, fx, fy = CONFIGS[name]
    return keyframe_camera(W, H, fx, fy, 0x5E65 + list(CONFIGS).index(name) + seed_offset, keyframe)


def make_scene(P: int, width: int, height: int, fx: float, fy: float, seed: int = 0x5E65,
               bg=(0.0, 0.0, 0.0), name: str = "custom", keyframe: int = 0) -> Scene:
    """Seeded scene of SURVEY.md section 8d.  `keyframe` perturbs only the camera pose (same Gaussians),
    which is what keyframe-parallel training needs (section 8e)."""
    cam = keyframe_camera(width, height, fx, fy, seed, keyframe)
    tanx, tany = cam.tanfovx, cam.tanfovy

    u = lambda stream: uniform01(P, stream, seed)  # noqa: E731
    z = 1.0 + 5.0 * u(1)
    near = u(2) < 0.03
    z = np.where(near, 0.01 + 0.19 * u(3), z).astype(np.float32)
    x = ((u(4) * 2 - 1) * 1.2 * z * np.float32(tanx)).astype(np.float32)
    y = ((u(5) * 2 - 1) * 1.2 * z * np.float32(tany)).astype(np.float32)
    means = np.stack([x, y, z], axis=1).astype(np.float32)

    k = (5e5 / max(P, 5e5)) ** (1.0 / 3.0)
    lo, hi = math.log(0.003 * k), math.log(0.025 * k)
    scales = np.stack([np.exp(lo + (hi - lo) * u(6 + a).astype(np.float64)) for a in range(3)], axis=1).astype(np.float32)

    # normalised N(0,1)^4 via Box-Muller
    u1 = np.maximum(uniform01(2 * P, 10, seed).astype(np.float64), 1e-12)
    u2 = uniform01(2 * P, 11, seed).astype(np.float64)
    rad = np.sqrt(-2.0 * np.log(u1))
    g = np.concatenate([rad * np.cos(2 * math.pi * u2), rad * np.sin(2 * math.pi * u2)])[: 4 * P].reshape(P, 4) if P else np.zeros((0, 4))
    g = g + (np.abs(g).sum(axis=1, keepdims=True) == 0)  # avoid the zero quaternion
    rot = (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32) if P else np.zeros((0, 4), np.float32)

    opacity = (0.02 + 0.98 * u(12)).astype(np.float32).reshape(P, 1)
    colors = np.stack([u(13), u(14), u(15)], axis=1).astype(np.float32)
    n_pix = 3 * height * width
    dL = ((uniform01(n_pix, 20, seed) * 2 - 1) / np.float32(n_pix)).astype(np.float32).reshape(3, height, width)
    return Scene(name, cam, np.ascontiguousarray(means), np.ascontiguousarray(scales), np.ascontiguousarray(rot),
                 opacity, np.ascontiguousarray(colors), np.asarray(bg, dtype=np.float32), dL,
                 meta={"seed": seed, "keyframe": keyframe})


def make_config_scene(name: str, seed_offset: int = 0, keyframe: int = 0, **overrides) -> Scene:
    P, W, H, fx, fy = CONFIGS[name]
    P = overrides.pop("P", P)
    idx = list(CONFIGS).index(name)
    return make_scene(P, W, H, fx, fy, seed=0x5E65 + idx + seed_offset, name=name, keyframe=keyframe, **overrides)
