"""Scaffold-GS anchors -> neural Gaussians on the GPU, and the anchor-level trainer step around it.

Host mirror of GaussianRenderer::generate_neural_gaussians / render / prefilter_voxel (src/gaussian_renderer.cpp:214-334,
21-129, 131-199) and of the anchor-level loop body of GaussianTrainer::trainingOnce (src/gaussian_trainer.cpp:47-117) /
GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:823-1032) over the C ABI of include/segs_neural.h.
No CPU fallback: everything here needs the HIP library.

All trainable state of the model (the reference's Adam groups 0-2, 4, 6-8, 10, 11: anchor, offset, anchor_feat,
scaling, the MLPs) lives in ONE flat fp32 bucket with a same-shaped gradient bucket: the operand of the fused Adam and
of the keyframe-parallel all-reduce.  `_opacity` and `_rotation` (groups 3, 5) never receive a gradient
 and are kept outside the bucket.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _capi
from .gaussian_trainer import DeviceStepCount, FusedL1SSIM, expon_lr
from .raster_engine import RasterEngine


@dataclass
class ModelDims:
    """Model.* keys of cfg/gaussian_mapper/RGB-D/Replica/office0.yaml:13-26 that shape the MLPs."""
    feat_dim: int = 32
    n_offsets: int = 10
    appearance_dim: int = 32
    use_feat_bank: bool = True
    add_opacity_dist: bool = False
    add_cov_dist: bool = False
    add_color_dist: bool = False

    def c_struct(self) -> _capi.NeuralDims:
        return _capi.NeuralDims(self.feat_dim, self.n_offsets, self.appearance_dim, int(self.use_feat_bank),
                                int(self.add_opacity_dist), int(self.add_cov_dist), int(self.add_color_dist))

    def mlp_tensor_names(self) -> List[str]:
        names = []
        for m in ("mlp_opacity", "mlp_cov", "mlp_color"):
            names += [f"{m}.0.weight", f"{m}.0.bias", f"{m}.2.weight", f"{m}.2.bias"]
        if self.appearance_dim > 0:
            names += ["mlp_apperance.0.weight", "mlp_apperance.0.bias"]
        if self.use_feat_bank:
            names += [f"mlp_feature_bank.{k}" for k in ("0.weight", "0.bias", "2.weight", "2.bias")]
        return names

    def mlp_tensor_shape(self, name: str) -> Tuple[int, ...]:
        fd, no = self.feat_dim, self.n_offsets
        m, layer, kind = name.split(".")
        if m == "mlp_apperance":
            return (self.appearance_dim, 7) if kind == "weight" else (self.appearance_dim,)
        if m == "mlp_feature_bank":
            out, inn = (fd, 4) if layer == "0" else (3, fd)
        else:
            extra = {"mlp_opacity": int(self.add_opacity_dist), "mlp_cov": int(self.add_cov_dist),
                     "mlp_color": int(self.add_color_dist) + self.appearance_dim}[m]
            nout = {"mlp_opacity": no, "mlp_cov": 7 * no, "mlp_color": 3 * no}[m]
            out, inn = (fd, fd + 3 + extra) if layer == "0" else (nout, fd)
        return (out, inn) if kind == "weight" else (out,)


@dataclass
class ScaffoldOptimizationParams:
    """Optimization.* of cfg/gaussian_mapper/RGB-D/Replica/office0.yaml:76-137 used by the step."""
    lambda_dssim: float = 0.2
    position_lr_init: float = 0.0
    position_lr_final: float = 0.0
    position_lr_max_steps: int = 30000
    offset_lr_init: float = 0.08
    offset_lr_final: float = 0.0001
    offset_lr_max_steps: int = 30000
    feature_lr: float = 0.0010
    scaling_lr: float = 0.005
    mlp_opacity_lr_init: float = 0.002
    mlp_opacity_lr_final: float = 0.00002
    mlp_opacity_lr_max_steps: int = 30000
    mlp_cov_lr_init: float = 0.004
    mlp_cov_lr_final: float = 0.004
    mlp_cov_lr_max_steps: int = 30000
    mlp_color_lr_init: float = 0.008
    mlp_color_lr_final: float = 0.00005
    mlp_color_lr_max_steps: int = 30000
    mlp_featurebank_lr_init: float = 0.01
    mlp_featurebank_lr_final: float = 0.00001
    mlp_featurebank_lr_max_steps: int = 30000
    appearance_lr_init: float = 0.05
    appearance_lr_final: float = 0.0005
    appearance_lr_max_steps: int = 30000
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-15


_p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


class ScaffoldModel:
    """Anchors + MLPs in one flat parameter bucket (plus same-shaped gradient and Adam-moment buckets) on `device`.

    The anchor segments are laid out for `capacity` >= A anchors so that densification (densify.AnchorDensifier) appends
    and prunes rows in place; views returned by param()/grad() cover the first A rows."""

    WIDTHS = (("anchor", 3), ("offset", None), ("anchor_feat", None), ("scaling", 6))

    def __init__(self, A: int, dims: ModelDims, device, capacity: Optional[int] = None):
        self.A, self.dims, self.device = int(A), dims, torch.device(device)
        self.capacity = max(int(capacity or A), self.A, 1)
        self._lib = _capi.lib()
        self._cdims = dims.c_struct()
        offs = (C.c_int64 * 18)()
        cnts = (C.c_int64 * 18)()
        nt, total = C.c_int(0), C.c_int64(0)
        _capi.check(self._lib.segs_neural_param_layout(C.byref(self._cdims), offs, cnts, C.byref(nt), C.byref(total)),
                    "segs_neural_param_layout")
        self._mlp_names = dims.mlp_tensor_names()
        assert nt.value == len(self._mlp_names)
        self.mlp_total = int(total.value)
        self._mlp_rel = {n: (int(offs[i]), int(cnts[i])) for i, n in enumerate(self._mlp_names)}
        self.widths = {"anchor": 3, "offset": 3 * dims.n_offsets, "anchor_feat": dims.feat_dim, "scaling": 6}
        self._allocate(self.capacity)

    def _allocate(self, capacity: int):
        self.capacity = int(capacity)
        self.seg_offset: Dict[str, int] = {}
        pos = 0
        for name in ("anchor", "offset", "anchor_feat", "scaling"):
            self.seg_offset[name] = pos
            pos += self.capacity * self.widths[name]
        self.mlp_offset = pos
        self.n_params = pos + self.mlp_total
        f = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.n_params, **f)
        self.grads = torch.zeros(self.n_params + 4, **f)[:self.n_params]   # (+4: keyframe_parallel.BucketExchange's spare tail)
        self.exp_avg = torch.zeros(self.n_params, **f)       # Adam moments, same layout (torch::optim::Adam state)
        self.exp_avg_sq = torch.zeros(self.n_params, **f)
        self.mlp_layout = {n: (pos + o, c) for n, (o, c) in self._mlp_rel.items()}
        self.rotation = torch.zeros((self.capacity, 4), **f)   # _rotation: identity quaternion, never trained
        self.rotation[:, 0] = 1.0
        self.opacity = torch.zeros((self.capacity, 1), **f)    # _opacity: unused by the forward

    def reserve(self, capacity: int):
        """Grow the buckets to `capacity` anchors, keeping the first A rows of every segment and the MLP block."""
        if capacity <= self.capacity:
            return
        old = {k: getattr(self, k) for k in ("params", "grads", "exp_avg", "exp_avg_sq")}
        old_off, old_mlp, old_rot, old_op = dict(self.seg_offset), self.mlp_offset, self.rotation, self.opacity
        self._allocate(capacity)
        for k, src in old.items():
            dst = getattr(self, k)
            for name, w in self.widths.items():
                n = self.A * w
                dst[self.seg_offset[name]:self.seg_offset[name] + n] = src[old_off[name]:old_off[name] + n]
            dst[self.mlp_offset:] = src[old_mlp:]
        self.rotation[:self.A] = old_rot[:self.A]
        self.opacity[:self.A] = old_op[:self.A]

    @property
    def segments(self) -> Dict[str, Tuple[int, int]]:
        """(offset, live element count) of the anchor segments."""
        return {n: (self.seg_offset[n], self.A * w) for n, w in self.widths.items()}

    # -- views
    def _view(self, bucket, name, rows=None):
        if name in self.widths:
            rows = self.A if rows is None else rows
            o = self.seg_offset[name]
            shape = {"anchor": (rows, 3), "offset": (rows, self.dims.n_offsets, 3),
                     "anchor_feat": (rows, self.dims.feat_dim), "scaling": (rows, 6)}[name]
            return bucket[o:o + rows * self.widths[name]].view(shape)
        o, n = self.mlp_layout[name]
        return bucket[o:o + n].view(self.dims.mlp_tensor_shape(name))

    def param(self, name, rows=None):
        return self._view(self.params, name, rows)

    def grad(self, name, rows=None):
        return self._view(self.grads, name, rows)

    @property
    def mlp_params(self):
        return self.params[self.mlp_offset:]

    @property
    def mlp_grads(self):
        return self.grads[self.mlp_offset:]

    def load(self, anchor, offset, anchor_feat, scaling_log, mlp: Dict[str, torch.Tensor]):
        for name, t in (("anchor", anchor), ("offset", offset), ("anchor_feat", anchor_feat), ("scaling", scaling_log)):
            self.param(name).copy_(t.to(self.device, torch.float32))
        for name in self.dims.mlp_tensor_names():
            self.param(name).copy_(mlp[name].to(self.device, torch.float32))

    def adam_groups(self, lrs: Dict[str, float]) -> List[Tuple[int, int, float]]:
        """(offset, count, lr) per Adam group in the reference's order (src/gaussian_model.cpp:632-690)."""
        seg = self.segments
        out = [(*seg["anchor"], lrs["anchor"]), (*seg["offset"], lrs["offset"]),
               (*seg["anchor_feat"], lrs["anchor_feat"]), (*seg["scaling"], lrs["scaling"])]
        for m, key in (("mlp_opacity", "mlp_opacity"), ("mlp_cov", "mlp_cov"), ("mlp_color", "mlp_color"),
                       ("mlp_apperance", "appearance"), ("mlp_feature_bank", "mlp_featurebank")):
            names = [n for n in self.mlp_layout if n.startswith(m + ".")]
            if names:
                o = self.mlp_layout[names[0]][0]
                out.append((o, sum(self.mlp_layout[n][1] for n in names), lrs[key]))
        return out


class NeuralGaussians:
    """segs_neural_forward / segs_neural_backward over resident candidate-domain buffers (A * n_offsets rows)."""

    def __init__(self, model: ScaffoldModel):
        self.model = model
        self._lib = model._lib
        self._cap = 0
        self._ensure()
        self._last = None

    @property
    def P(self):
        """live candidate slots = A * n_offsets (the leading rows of the capacity-sized buffers)"""
        return self.model.A * self.model.dims.n_offsets

    def _ensure(self):
        """(re)allocate the candidate-domain buffers when the model's capacity grew"""
        model = self.model
        if model.capacity <= self._cap:
            return
        self._cap = model.capacity
        f = dict(dtype=torch.float32, device=model.device)
        Pc = self._cap * model.dims.n_offsets
        self.P_capacity = Pc
        self.means3D = torch.zeros((Pc, 3), **f)
        self.colors = torch.zeros((Pc, 3), **f)
        self.opacity = torch.zeros((Pc, 1), **f)
        self.scales = torch.zeros((Pc, 3), **f)
        self.rotations = torch.zeros((Pc, 4), **f)
        self.neural_opacity = torch.zeros((Pc, 1), **f)
        self.temp = torch.empty(self._lib.segs_neural_temp_bytes(C.byref(model._cdims), self._cap), dtype=torch.uint8,
                                device=model.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.model.device).cuda_stream)

    def forward(self, camera_center: torch.Tensor, pose7: torch.Tensor, visible_radii: Optional[torch.Tensor]):
        m = self.model
        self._ensure()
        st = self._lib.segs_neural_forward(
            C.byref(m._cdims), m.A, _p(m.param("anchor")), _p(m.param("offset")), _p(m.param("anchor_feat")),
            _p(m.param("scaling")), _p(visible_radii), _p(m.mlp_params), _p(camera_center), _p(pose7), _p(self.means3D),
            _p(self.colors), _p(self.opacity), _p(self.scales), _p(self.rotations), _p(self.neural_opacity), _p(self.temp),
            self._stream())
        _capi.check(st, "segs_neural_forward")
        self._last = (camera_center, pose7)
        return self.means3D, self.colors, self.opacity, self.scales, self.rotations

    def forward_projected(self, kf: "Keyframe", visible_radii: Optional[torch.Tensor], engine, scale_modifier: float = 1.0,
                          anchor_rotations: Optional[torch.Tensor] = None):
        """segs_neural_forward_projected: the forward that also runs the rasterizer's per-Gaussian stage on the candidates it
        generates and leaves the records / radii / depth keys in `engine`'s resident buffers.  self.colors and
        self.opacity are NOT written (they exist only inside the records).  With `anchor_rotations` (normalised) the prefilter is
        folded in too: visible_radii is then filled by this call instead of being read."""
        m = self.model
        self._ensure()
        tg = engine.projection_targets()
        st = self._lib.segs_neural_forward_projected(
            C.byref(m._cdims), m.A, _p(m.param("anchor")), _p(m.param("offset")), _p(m.param("anchor_feat")),
            _p(m.param("scaling")), _p(visible_radii), _p(anchor_rotations), _p(m.mlp_params), _p(kf.campos), _p(kf.pose7),
            _p(self.means3D), _p(self.scales), _p(self.rotations), _p(self.neural_opacity), C.byref(tg), _p(kf.view), _p(kf.proj), engine.W, engine.H,
            float(kf.tanfovx), float(kf.tanfovy), float(scale_modifier), _p(self.temp), self._stream())
        _capi.check(st, "segs_neural_forward_projected")
        self._last = (kf.campos, kf.pose7)
        return self.means3D, self.scales, self.rotations

    def mask(self):
        """The reference's `mask` (neural_opacity > 0, src/gaussian_renderer.cpp:279) in the candidate domain."""
        return self.neural_opacity.view(-1)[:self.P] > 0

    def backward(self, dL_dmeans3D, dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, scaling_reg_weight: float = 0.0):
        """Accumulates into model.grads.  scaling_reg_weight adds the mapper's 0.01 * mean(prod(scaling)) term
        (src/gaussian_mapper.cpp:926-928); its value lands in self.scaling_reg."""
        if not hasattr(self, "scaling_reg"):
            self.scaling_reg = torch.zeros(1, dtype=torch.float32, device=self.model.device)
        m = self.model
        camera_center, pose7 = self._last
        st = self._lib.segs_neural_backward(
            C.byref(m._cdims), m.A, _p(m.param("anchor")), _p(m.param("offset")), _p(m.param("anchor_feat")),
            _p(m.param("scaling")), _p(m.mlp_params), _p(camera_center), _p(pose7), _p(dL_dmeans3D), _p(dL_dcolors),
            _p(dL_dopacity), _p(dL_dscales), _p(dL_drotations), _p(m.grad("anchor")), _p(m.grad("offset")),
            _p(m.grad("anchor_feat")), _p(m.grad("scaling")), _p(m.mlp_grads), float(scaling_reg_weight),
            _p(self.scaling_reg) if scaling_reg_weight else None, _p(self.temp), self._stream())
        _capi.check(st, "segs_neural_backward")


@dataclass
class Keyframe:
    """What the step needs of a GaussianKeyframe (src/gaussian_keyframe.cpp:151-184): device tensors + scalars."""
    view: torch.Tensor
    proj: torch.Tensor
    campos: torch.Tensor
    pose7: torch.Tensor      # (t_xyz, q_wxyz), gaussian_renderer.cpp:258-261
    tanfovx: float
    tanfovy: float

    def packed(self) -> torch.Tensor:
        """view | proj | campos | pose7 as one 42-float device tensor (made once): one copy refreshes a captured iteration's
        staging keyframe."""
        pk = getattr(self, "_packed", None)
        if pk is None:
            pk = self._packed = torch.cat([self.view.reshape(-1), self.proj.reshape(-1), self.campos.reshape(-1), self.pose7.reshape(-1)]).contiguous()
        return pk

    @classmethod
    def from_pose(cls, q_wxyz, t_xyz, width: int, height: int, fx: float, fy: float, device, znear: float = 0.01,
                  zfar: float = 100.0) -> "Keyframe":
        """GaussianKeyframe::setPose (src/gaussian_keyframe.cpp:21-42: the quaternion is normalised) followed by
        computeTransformTensors (:151-184: R = q.toRotationMatrix(), view / projection / full projection in the transposed
        layout, camera centre) -- the tensors one keyframe hands to the renderer, plus the 7-vector the appearance MLP reads."""
        import numpy as np
        from . import scenes
        q = np.asarray(q_wxyz, dtype=np.float64)
        q = q / np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float32)
        t = np.asarray(t_xyz, dtype=np.float32)
        cam = scenes.make_camera(width, height, fx, fy, R, t, znear, zfar)
        dev = torch.device(device)
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
        pose7 = np.concatenate([t, q.astype(np.float32)]).astype(np.float32)
        return cls(tt(cam.world_view_transform), tt(cam.full_proj_transform), tt(cam.camera_center), tt(pose7), cam.tanfovx,
                   cam.tanfovy)


class ScaffoldTrainerStep:
    """prefilter_voxel -> generate_neural_gaussians -> rasterize -> L1/SSIM -> backward -> [all-reduce] -> fused Adam,
    all on the device without a host synchronisation in steady state."""

    def __init__(self, model: ScaffoldModel, width: int, height: int, opt: Optional[ScaffoldOptimizationParams] = None,
                 spatial_lr_scale: float = 1.0, process_group=None, scaling_reg_weight: float = 0.0):
        # scaling_reg_weight = 0.01 gives the mapper's loss (src/gaussian_mapper.cpp:926-928), 0 the trainer's (:89-90 of
        # src/gaussian_trainer.cpp); the mapper's FFT regularisers (:930-945): enable_frequency_regularization()
        self.scaling_reg_weight = float(scaling_reg_weight)
        self.freq_reg = None             # see enable_frequency_regularization()
        self.model, self.opt = model, opt or ScaffoldOptimizationParams()
        self.W, self.H = int(width), int(height)
        dev = model.device
        self._lib = model._lib
        self.neural = NeuralGaussians(model)
        self.engine = RasterEngine(self.neural.P_capacity, width, height, dev, resident=True, skip_nonpositive_opacity=True)
        self.loss_fn = FusedL1SSIM(height, width, dev, self.opt.lambda_dssim)
        # Gaussian-pyramid training (src/gaussian_mapper.cpp:837-858, 872-875, 913-915): a keyframe is trained at the size of
        # its current pyramid level, i.e. of the target image it hands over.  One rasterizer engine + loss object per size,
        # made on first use; `engine` / `loss_fn` / `W` / `H` always name the level of the iteration in flight.
        self._levels = {(self.W, self.H): (self.engine, self.loss_fn)}
        self.bg = torch.zeros(3, dtype=torch.float32, device=dev)      # Model.white_background: set_background(True)
        self.visible_radii = torch.zeros(model.capacity, dtype=torch.int32, device=dev)
        self.spatial_lr_scale = float(spatial_lr_scale)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.iteration = 0
        # torch::optim::Adam keeps one step count per parameter; they only diverge at densification iterations, where
        # the re-created anchor tensors have no gradient and are skipped by the optimizer (src/gaussian_model.cpp:1677).
        # Both counts live on the device (a guarded step that is dropped there must not advance them); the anchor groups
        # share the MLPs' count until the first densification.
        self._mlp_count = DeviceStepCount(dev)
        self._anchor_count = None
        # N > 1: True = reduce-scatter -> Adam on this rank's shard -> all-gather (keyframe_parallel.BucketExchange); False = dense
        # all-reduce and a full Adam on every rank; "auto" = by bucket size (BucketExchange.AUTO_SHARD_BYTES)
        self.sharded_optimizer = True
        self.single_rank_collectives = False   # keyframe_parallel.BucketExchange: run the collectives with one rank too
        self.densifier = None            # densify.AnchorDensifier, see enable_densification()
        self.densify_generator = None
        self.keyframe_selector = None    # keyframe_window.SlidingWindowKeyframes: the mapper's walk instead of round-robin
        # the mapper (not the trainer) multiplies rendering and target by mask_rgb = (gt != 0).any(-1): gt is (3,H,W), so this
        # is a per-(channel, row) mask of shape (3,H,1) that blanks rows whose target is entirely zero
        # (src/gaussian_mapper.cpp:917-922).  Off by default (trainer semantics); mapper_config.make_mapper_step turns it on.
        self.row_mask = False
        self._row_mask_cache = {}
        # Whole-iteration hipGraph (enable_graph): the iterations between two adjust_anchor calls issue a fixed launch sequence
        # over fixed addresses once the per-iteration values (keyframe matrices, target image, learning rates, the frequency
        # regulariser's target tables) sit in staging buffers refreshed before each replay.
        # let the neural forward run the rasterizer's per-Gaussian stage on the candidates while it has them in
        # registers (segs_neural_forward_projected) once the resident buffers are calibrated.  Same image, radii, gradients bit
        # for bit; neural.colors / neural.opacity are then not materialised (set False before render() to have them).
        self.fuse_projection = True
        # the device drops an iteration whose forward overflowed the resident capacity; with one rank the host runs it again
        # before the next one (training_once), so no optimizer step of the reference's sequence is lost
        self.redo_dropped_steps = True
        self.redone_steps = 0
        self._last_iteration = None
        self.use_graph = False
        self._graphs = {}
        self._graph_stage = {}
        self.graph_replays = 0

    def set_background(self, white: bool):
        """bg_color of GaussianMapper's constructor (src/gaussian_mapper.cpp:61-67)."""
        self.bg.fill_(1.0 if white else 0.0)

    def enable_frequency_regularization(self, lambda_high: float = 0.01, lambda_low: float = 0.0, scales=(1.0, 0.5, 0.25),
                                        start: int = 5000, until: int = 25500, multi_resolution: bool = True,
                                        fused: bool = True):
        """The mapper's FFT regularisers (src/gaussian_mapper.cpp:930-945, Mapper.* keys of the Replica cfg :140-146).
        `fused` (default): frequency_loss.FusedFrequencyLoss -- cached |FFT(gt)|, one forward and one inverse real FFT per scale
        (hipFFT library calls, as in the reference) and three small kernels around them (csrc/freq_loss.hip); otherwise the
        torch.fft + autograd mirror of the reference's op chain (loss_utils.py), kept as the A/B and parity partner."""
        self.freq_reg = dict(lambda_high=lambda_high, lambda_low=lambda_low, scales=tuple(scales), start=start, until=until,
                             multi=multi_resolution, fused=bool(fused))
        self._freq_fused = {}
        # a captured iteration holds the old plan's scratch and target-table addresses, which die with the old object
        self._graphs.clear()
        self._graph_stage.clear()

    def _freq_active(self):
        """(low term on, high term on) at this iteration: the two conditions of src/gaussian_mapper.cpp:932-944."""
        fr, it = self.freq_reg, self.iteration
        return (it < fr["until"] and fr["lambda_low"] != 0.0, fr["start"] < it < fr["until"] and fr["lambda_high"] != 0.0)

    def _freq_fused_add(self, image: torch.Tensor, gt: torch.Tensor, dL: torch.Tensor, loss_word: torch.Tensor) -> bool:
        """Adds the regulariser's gradient to dL and its value to loss_word, both in place; False when it is off."""
        from .frequency_loss import FusedFrequencyLoss
        fr = self.freq_reg
        low_on, high_on = self._freq_active()
        if not (low_on or high_on):
            return False
        key = (self.W, self.H, low_on, high_on)
        fl = self._freq_fused.get(key)
        if fl is None:
            fl = self._freq_fused[key] = FusedFrequencyLoss(
                self.H, self.W, self.model.device, lambda_high=fr["lambda_high"] if high_on else 0.0, scales=fr["scales"],
                multi_resolution=fr["multi"], lambda_low=fr["lambda_low"] if low_on else 0.0)
        fl(image, gt, dL, loss_word)
        return True

    def _freq_grad(self, image: torch.Tensor, gt: torch.Tensor):
        from . import loss_utils
        fr, it = self.freq_reg, self.iteration
        terms = []
        img = image.detach().requires_grad_(True)
        if it < fr["until"] and fr["lambda_low"] != 0.0:
            terms.append(fr["lambda_low"] * loss_utils.low_freq_loss(img, gt))
        if fr["start"] < it < fr["until"] and fr["lambda_high"] != 0.0:
            hf = loss_utils.multi_scale_loss(img, gt, fr["scales"]) if fr["multi"] else loss_utils.high_frequency_loss(img, gt)
            terms.append(fr["lambda_high"] * hf)
        if not terms:
            return None, None
        total = sum(terms)
        (g,) = torch.autograd.grad(total, img)
        return total.detach(), g

    def enable_densification(self, densifier, seed: int = 0):
        """training_statis / adjust_anchor on the schedule of trainForOneIteration (src/gaussian_mapper.cpp:961-968); the
        random keep masks come from a generator seeded identically on every rank."""
        self.densifier = densifier
        self.densify_generator = torch.Generator(device="cpu").manual_seed(seed)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.model.device).cuda_stream)

    def _adam(self, groups, count: "DeviceStepCount", guard):
        """Fused Adam over `groups` (restricted to this rank's shard of the bucket when the optimizer is sharded), guarded by
        the summed overflow word, step count on the device."""
        groups = self._exchange().clip_segments(groups)
        call = count.eager_call()        # the launch below always happens (an empty shard still advances the count)
        if not groups:
            groups = [(0, 0, 0.0)]
        segs = (_capi.AdamSegment * len(groups))()
        for i, (o, n, lr) in enumerate(groups):
            segs[i].offset, segs[i].count, segs[i].lr = o, n, float(lr)
        m = self.model
        st = self._lib.segs_adam_step_device(_p(m.params), _p(m.grads), _p(m.exp_avg), _p(m.exp_avg_sq), segs, len(groups),
                                             self.opt.beta1, self.opt.beta2, self.opt.eps, _p(count.words), call,
                                             1.0 / self.world, 1, guard, self._stream())
        _capi.check(st, "segs_adam_step_device")

    def learning_rates(self, it: int) -> Dict[str, float]:
        """updateLearningRate (src/gaussian_model.cpp:874-915); anchor/offset scaled by spatial_lr_scale (:637,640)."""
        o = self.opt
        return {
            "anchor": expon_lr(it, o.position_lr_init * self.spatial_lr_scale, o.position_lr_final * self.spatial_lr_scale,
                               o.position_lr_max_steps),
            "offset": expon_lr(it, o.offset_lr_init * self.spatial_lr_scale, o.offset_lr_final * self.spatial_lr_scale,
                               o.offset_lr_max_steps),
            "anchor_feat": o.feature_lr, "scaling": o.scaling_lr,
            "mlp_opacity": expon_lr(it, o.mlp_opacity_lr_init, o.mlp_opacity_lr_final, o.mlp_opacity_lr_max_steps),
            "mlp_cov": expon_lr(it, o.mlp_cov_lr_init, o.mlp_cov_lr_final, o.mlp_cov_lr_max_steps),
            "mlp_color": expon_lr(it, o.mlp_color_lr_init, o.mlp_color_lr_final, o.mlp_color_lr_max_steps),
            "mlp_featurebank": expon_lr(it, o.mlp_featurebank_lr_init, o.mlp_featurebank_lr_final, o.mlp_featurebank_lr_max_steps),
            "appearance": expon_lr(it, o.appearance_lr_init, o.appearance_lr_final, o.appearance_lr_max_steps),
        }

    def _anchor_rotations(self) -> torch.Tensor:
        """normalize(_rotation) of the live anchors.  _rotation is never trained (src/gaussian_model.cpp:372): normalised again
        only when densification rewrote rows."""
        m = self.model
        key = (m.A, m.rotation._version, m.rotation.data_ptr())
        if getattr(self, "_rot_key", None) != key:
            self._rot_key, self._rot_normalized = key, torch.nn.functional.normalize(m.rotation[:m.A]).contiguous()
        return self._rot_normalized

    def prefilter_voxel(self, kf: Keyframe) -> torch.Tensor:
        """radii of the anchors drawn as Gaussians with exp(scaling[:, :3]) and normalize(rotation)
        (src/gaussian_renderer.cpp:131-199); the result stays on the device."""
        m = self.model
        rots = self._anchor_rotations()
        # get_scaling()[:, :3] = exp(_scaling[:, :3]) is formed inside the kernel (rows of 6 log-scales): no intermediate tensor
        st = self._lib.segs_visible_filter_log_scales(m.A, self.W, self.H, _p(m.param("anchor")), _p(m.param("scaling")), 6, _p(rots),
                                                      _p(kf.view), _p(kf.proj), float(kf.tanfovx), float(kf.tanfovy),
                                                      _p(self.visible_radii), self._stream())
        _capi.check(st, "segs_visible_filter_log_scales")
        return self.visible_radii

    def use_level(self, width: int, height: int):
        """Make (width, height) the size of the iteration in flight (a pyramid level of the keyframe: same field of view,
        fewer pixels)."""
        key = (int(width), int(height))
        if key == (self.W, self.H):
            return
        lv = self._levels.get(key)
        if lv is None:
            lv = self._levels[key] = (RasterEngine(self.neural.P_capacity, key[0], key[1], self.model.device, resident=True,
                                                   skip_nonpositive_opacity=True),
                                      FusedL1SSIM(key[1], key[0], self.model.device, self.opt.lambda_dssim))
        self._levels[(self.W, self.H)] = (self.engine, self.loss_fn)     # (a caller may have wrapped the current level's loss)
        self.engine, self.loss_fn = lv
        self.W, self.H = key

    def render(self, kf: Keyframe) -> torch.Tensor:
        ng = self.neural
        if self.model.capacity * self.model.dims.n_offsets > self.engine.P:   # the map outgrew the engines' buffers
            self.engine = RasterEngine(self.model.capacity * self.model.dims.n_offsets, self.W, self.H, self.model.device,
                                       resident=True, skip_nonpositive_opacity=True)
            self._levels = {(self.W, self.H): (self.engine, self.loss_fn)}   # the other levels' engines are re-made on use
            self.visible_radii = torch.zeros(self.model.capacity, dtype=torch.int32, device=self.model.device)
        self.engine.set_active(ng.P)
        if self.fuse_projection:
            self.engine.check(raise_on_overflow=False)   # an overflow of the previous step sends this one through the calibrating path
        if self.fuse_projection and self.engine.can_take_projected():
            # the neural forward projects the candidates itself; the rasterizer starts at the binning
            # (and works out prefilter_voxel's radii on the way: self.visible_radii is filled by the same call)
            ng.forward_projected(kf, self.visible_radii, self.engine, anchor_rotations=self._anchor_rotations())
            return self.engine.forward_projected(self.bg, ng.means3D, ng.scales, ng.rotations, kf.view, kf.proj, kf.campos,
                                                 kf.tanfovx, kf.tanfovy)
        ng.forward(kf.campos, kf.pose7, self.prefilter_voxel(kf))
        return self.engine.forward(self.bg, ng.means3D, ng.colors, ng.opacity, ng.scales, ng.rotations, kf.view, kf.proj,
                                   kf.campos, kf.tanfovx, kf.tanfovy)

    def _forward_backward(self, kf: Keyframe, gt: torch.Tensor, exchange=None, flag_on_host: bool = False):
        """`flag_on_host`: the caller reads the summed overflow word on the host before the gradient exchange (adjust_anchor
        iterations), so it needs its own collective instead of riding with the gradients."""
        self.use_level(gt.shape[-1], gt.shape[-2])
        if self.model.A == 0:
            # every anchor was pruned: the reference's rasterizer short-circuits P == 0 to a zero image
            # (src/rasterize_points.cu:81) and nothing receives a gradient
            if exchange is not None:
                exchange.reduce_flag_async(None, allow_piggyback=not flag_on_host)
            return self.loss_fn(torch.zeros(3, self.H, self.W, device=self.model.device), gt)[0]
        image = self.render(kf)
        if exchange is not None:
            # the overflow word is final once the forward's binning has run: its all-reduce hides behind loss and backward
            status = getattr(self.engine, "_status", None)
            exchange.reduce_flag_async(status[3:4] if (status is not None and self.engine._last_resident) else None,
                                       allow_piggyback=not flag_on_host)
        mask = None
        if self.row_mask:
            mask, gt = self._row_mask_of(gt)
            if mask is not None:
                image = image * mask
        loss, dL = self.loss_fn(image, gt)
        if self.freq_reg is not None:
            if self.freq_reg["fused"]:
                # loss is element 0 of the fused L1/SSIM object's result words, dL its own gradient buffer: both updated in place
                self._freq_fused_add(image, gt, dL, loss.view(1))
            else:
                floss, fg = self._freq_grad(image, gt)
                if fg is not None:
                    dL = dL + fg
                    loss = loss + floss
        if mask is not None:
            dL = dL * mask
        g = self.engine.backward(dL)
        self.neural.backward(g["means3D"], g["colors"], g["opacity"], g["scales"], g["rotations"], self.scaling_reg_weight)
        return loss

    def _row_mask_of(self, gt: torch.Tensor):
        """(mask or None, masked target).  Evaluated once per target tensor (one host read); targets without an all-zero
        row -- the normal case -- take the unmasked fast path."""
        key = (gt.data_ptr(), gt._version)
        hit = self._row_mask_cache.get(key)
        if hit is None:
            mask = (gt != 0).any(-1).to(torch.float32).unsqueeze(-1)
            hit = (None, gt) if bool(mask.all()) else (mask, (gt * mask).contiguous())
            if len(self._row_mask_cache) > 4096:
                self._row_mask_cache.clear()
            self._row_mask_cache[key] = hit
        return hit

    def keyframe_for(self, step: int, n_keyframes: int) -> int:
        return (step * self.world + self.rank) % n_keyframes

    def training_once(self, keyframes: List[Keyframe], gt_images: List[torch.Tensor]) -> torch.Tensor:
        """One mapper iteration.  Nothing in it waits for the device except the densification iterations (adjust_anchor
        sizes tensors on the host): a pass whose instance count outgrew the rasterizer's resident capacity on ANY rank is
        dropped on the device by every rank -- statistics and optimizer are guarded by the all-reduced overflow word, the Adam
        step counts live on the device and do not advance -- and the rank that overflowed re-sizes its scratch at its next
        forward.  The loss returned for such an iteration comes from an invalid image.

        With eager launches (`redo_dropped_steps`, on by default; not under `enable_graph`, whose replay loop polls the status
        word instead of waiting for it) a dropped iteration is not lost: the host learns of it when it resolves that step's
        overflow word -- one rank: the engine's own status word, i.e. before anything of the next iteration is queued; N > 1
        ranks: the SUMMED word, which every rank mirrored into pinned host memory right after the gradient exchange
        (BucketExchange.mirror_flag), so every rank takes the same decision without another collective -- and runs the same
        keyframe with the same iteration number again right there, on every rank (the forward of the rank that overflowed
        re-calibrates).  Parameters, moments, statistics and step counts are what the dropped pass found, so the optimizer takes
        every step the reference takes (src/gaussian_mapper.cpp:823-1032 never skips one), in the same order, and replicas stay
        bit-identical.

        CONTRACT: the keyframe's and the target's tensors handed to a call must stay unchanged until the next call (or
        finish()) has returned -- a redo trains on them again (refresh staging buffers only after that)."""
        self._redo_if_dropped()
        self.iteration += 1
        if self.keyframe_selector is not None:
            # useOneRandomSlidingWindowKeyframe (src/gaussian_mapper.cpp:827): one draw per rank, identical on every rank
            k = self.keyframe_selector.use_for_ranks(self.world)[self.rank]
        else:
            k = self.keyframe_for(self.iteration - 1, len(keyframes))
        self._last_iteration = (keyframes[k], gt_images[k], self.iteration)
        return self._iteration_body(keyframes[k], gt_images[k], self.iteration)

    def _redo_if_dropped(self):
        prev = self._last_iteration
        if prev is None or not self.redo_dropped_steps or self.use_graph or not self.engine.resident:
            return
        for _ in range(4):
            if self.world == 1:
                dropped = not self.engine.check(raise_on_overflow=False)
            else:
                dropped = bool(self._exchange().step_dropped())
                self.engine.check(raise_on_overflow=False)       # the rank that overflowed re-calibrates in its next forward
            if not dropped:
                break
            self.redone_steps += 1
            self._iteration_body(*prev)
            if self.world == 1:
                break                                            # (a re-calibrating forward cannot overflow)
        else:
            raise RuntimeError("an iteration kept being dropped by the device")
        self._last_iteration = None

    def finish(self):
        """Resolve the LAST iteration's overflow word and run that iteration again if the device dropped it (training_once only
        learns of a drop at the next call).  Call once after the last training_once of a run, before reporting."""
        self._redo_if_dropped()

    def lost_steps(self) -> int:
        """Iterations the device dropped and nobody ran again (0 with redo_dropped_steps after finish(); synchronises)."""
        return self.dropped_steps() - self.redone_steps

    def _iteration_body(self, kf: Keyframe, gt: torch.Tensor, it: int) -> torch.Tensor:
        """Iteration `it` on keyframe `kf` (training_once; also the re-run of an iteration the device dropped)."""
        lrs = self.learning_rates(it)
        ex = self._exchange()
        d = self.densifier
        in_stat_window = d is not None and self.model.A > 0 and d.p.start_stat < it < d.p.update_until  # gaussian_mapper.cpp:961-968
        adjust_now = in_stat_window and it > d.p.update_from and it % d.p.update_interval == 0
        if self.use_graph and not adjust_now and not ex.active:
            loss = self._training_once_graph(kf, gt, lrs, in_stat_window)
            if loss is not None:
                return loss
        loss = self._forward_backward(kf, gt, ex, flag_on_host=adjust_now)
        flag = ex.wait_flag()
        if adjust_now:
            # adjust_anchor reads tensor sizes on the host and must see a valid pass on every rank: resolve the summed
            # overflow word here (the one synchronisation of a densification iteration) and redo the pass while it is set
            for _ in range(3):
                if int(flag.item()) == 0:
                    break
                self.engine.check(raise_on_overflow=False)     # the rank that overflowed re-calibrates in its next forward
                self.model.grads.zero_()
                loss = self._forward_backward(kf, gt, ex, flag_on_host=True)
                flag = ex.wait_flag()
            else:
                raise RuntimeError("resident rasterizer kept overflowing its re-sized scratch")
        guard = C.c_void_p(flag.data_ptr())
        # A densification may re-size the bucket (reserve() moves the MLP block), so the shard partition the optimizer clips
        # to below is not the one a reduce-scatter would have summed for: every element gets the full sum on those steps.
        ex.reduce_gradients(self.model.grads, dense=adjust_now)
        if ex.active and self.redo_dropped_steps and not adjust_now:
            ex.mirror_flag()              # (an adjust_anchor iteration has resolved its word above)
        adjusted = False
        if in_stat_window:
            d.training_statis(self.neural, self.visible_radii, self.engine.radii, self.engine.dL_dmean2D, guard,
                              into_delta=self.world > 1)
            if adjust_now:
                if ex.sharded:      # moments are only current inside each rank's shard: make them whole before rows move
                    ex.gather(self.model.exp_avg)
                    ex.gather(self.model.exp_avg_sq)
                d.reduce_statistics(self.pg)
                d.adjust_anchor(generator=self.densify_generator, views_per_iteration=self.world)
                adjusted = True
                ex = self._exchange()      # the bucket may have been re-sized: new shard ranges (the moments are whole here)
        groups = self.model.adam_groups(lrs)
        anchor_groups, mlp_groups = groups[:4], groups[4:]
        if adjusted:
            # the six anchor tensors were re-created by adjust_anchor: no gradient, skipped by Adam this iteration
            # (src/gaussian_model.cpp:1677), so from here on their step count lags the MLPs'
            for name in self.model.widths:
                self.model.grad(name).zero_()
            if self._anchor_count is None:
                self._anchor_count = self._mlp_count.clone()
            self._adam(mlp_groups, self._mlp_count, guard)
        elif self._anchor_count is None:
            self._adam(groups, self._mlp_count, guard)
        else:
            self._adam(anchor_groups, self._anchor_count, guard)
            self._adam(mlp_groups, self._mlp_count, guard)
        if ex.sharded:
            ex.gather(self.model.params)
            self.model.grads.zero_()      # outside this rank's shard the bucket still holds its own contribution
        return loss

    # ---- whole-iteration hipGraph ------------------------------------------------------------------------------------
    def enable_graph(self, on: bool = True):
        """Replay the iterations whose launch sequence is fixed (single rank, no adjust_anchor this iteration, a calibrated
        resident rasterizer, a target without blanked rows) from a captured hipGraph.  Same kernels, same arguments, same
        order as the eager path: parameters stay bit-identical (tests/test_graph_step_gpu.py); what goes is the host's
        per-launch cost and the gaps between dependent small launches."""
        self.use_graph = bool(on)
        self._graphs.clear()

    def _graph_stage_for(self, fl):
        key = (self.W, self.H)
        st = self._graph_stage.get(key)
        if st is None:
            dev = self.model.device
            pk = torch.zeros(42, dtype=torch.float32, device=dev)
            st = dict(packed=pk, gt=torch.empty((3, self.H, self.W), dtype=torch.float32, device=dev),
                      lr=torch.zeros(16, dtype=torch.float64, device=dev), table=None)
            self._graph_stage[key] = st
        if fl is not None and (st["table"] is None or st["table"].numel() != fl._target_floats):
            st["table"] = torch.empty(fl._target_floats, dtype=torch.float32, device=self.model.device)
        return st

    def _training_once_graph(self, kf: Keyframe, gt: torch.Tensor, lrs, in_stat_window: bool):
        """One iteration from the captured graph; None when this iteration has to take the eager path."""
        self.use_level(gt.shape[-1], gt.shape[-2])
        eng, m = self.engine, self.model
        if m.A == 0 or not eng.resident or eng.capacity <= 0 or not eng.poll() or eng.capacity <= 0:
            return None                                  # not calibrated (or an overflow just came to light): eager, which re-sizes
        if m.capacity * m.dims.n_offsets > eng.P:
            return None
        if self.row_mask and self._row_mask_of(gt)[0] is not None:
            return None                                  # a target with blanked rows multiplies image and gradient by its mask: eager
        fl = None
        if self.freq_reg is not None:
            low_on, high_on = self._freq_active()
            if low_on or (high_on and not self.freq_reg["fused"]):
                return None
            if high_on:
                from .frequency_loss import FusedFrequencyLoss
                fr = self.freq_reg
                fkey = (self.W, self.H, False, True)
                fl = self._freq_fused.get(fkey)
                if fl is None:
                    fl = self._freq_fused[fkey] = FusedFrequencyLoss(self.H, self.W, m.device, lambda_high=fr["lambda_high"], scales=fr["scales"],
                                                                      multi_resolution=fr["multi"])
        st = self._graph_stage_for(fl)
        groups = m.adam_groups(lrs)
        split = self._anchor_count is not None
        key = (self.W, self.H, m.A, m.capacity, id(eng), eng.capacity, eng._bin_r.data_ptr(), float(kf.tanfovx), float(kf.tanfovy),
               bool(in_stat_window), fl is not None, split, len(groups), m.params.data_ptr(), bool(self.fuse_projection))
        # ---- refresh the staging buffers (ordinary stream work in front of the replay)
        st["packed"].copy_(kf.packed())
        st["gt"].copy_(gt)
        if fl is not None:
            st["table"].copy_(fl.target_block(gt))
        vals = (C.c_double * len(groups))(*[float(g[2]) for g in groups])
        _capi.check(self._lib.segs_set_doubles(_p(st["lr"]), vals, len(groups), self._stream()), "segs_set_doubles")
        counts = (self._anchor_count, self._mlp_count) if split else (self._mlp_count,)
        for c in counts:
            c.sync_device_calls()
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) > 16:
                self._graphs.clear()
            pk = st["packed"]
            skf = Keyframe(pk[0:16].view(4, 4), pk[16:32].view(4, 4), pk[32:35], pk[35:42], kf.tanfovx, kf.tanfovy)
            status = eng._status
            guard = C.c_void_p(status[3:4].data_ptr())
            d = self.densifier
            seg_all = [(o, n) for o, n, _ in groups]

            def adam(seg, lr_off, count):
                segs = (_capi.AdamSegment * len(seg))()
                for i, (o, n) in enumerate(seg):
                    segs[i].offset, segs[i].count, segs[i].lr = o, n, 0.0
                lr_ptr = C.c_void_p(st["lr"].data_ptr() + 8 * lr_off)
                rc = self._lib.segs_adam_step_graph(_p(m.params), _p(m.grads), _p(m.exp_avg), _p(m.exp_avg_sq), segs, len(seg), lr_ptr,
                                                    self.opt.beta1, self.opt.beta2, self.opt.eps, _p(count.words), 1.0, 1, guard, self._stream())
                _capi.check(rc, "segs_adam_step_graph")

            def body():
                image = self.render(skf)
                loss, dL = self.loss_fn(image, st["gt"])
                if fl is not None:
                    fl.apply(image, st["table"], dL, loss.view(1))
                gr = eng.backward(dL)
                self.neural.backward(gr["means3D"], gr["colors"], gr["opacity"], gr["scales"], gr["rotations"], self.scaling_reg_weight)
                if in_stat_window:
                    d.training_statis(self.neural, self.visible_radii, eng.radii, eng.dL_dmean2D, guard, into_delta=False)
                if split:
                    adam(seg_all[:4], 0, self._anchor_count)
                    adam(seg_all[4:], 4, self._mlp_count)
                else:
                    adam(seg_all, 0, self._mlp_count)

            eng.check(raise_on_overflow=False)           # nothing pending while the capture runs
            if eng.capacity <= 0:
                return None
            if fl is not None and not getattr(fl, "_ran_eagerly", False):
                # the FFT library sets a transform up on its first execution: not inside a capture
                fl.apply(eng.out_color, st["table"], torch.zeros_like(st["gt"]))
                fl._ran_eagerly = True
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            self._graphs[key] = g
        g.replay()
        for c in counts:
            c.calls += 1                                 # (the device-side call count advanced with the replay)
        eng.after_graph_replay()
        self.graph_replays += 1
        return self.loss_fn.out[0]

    def profile_phases(self, kf: Keyframe, gt: torch.Tensor, iters: int = 20) -> Dict[str, float]:
        """Mean milliseconds per phase of one iteration (HIP events on the current stream between the same calls
        training_once issues; measurement support for bench.py, single rank, no densification)."""
        names = ("prefilter_voxel", "neural_forward", "raster_forward", "loss", "freq_loss", "raster_backward", "neural_backward",
                 "adam")
        tot = {n: 0.0 for n in names}
        ng = self.neural
        for _ in range(iters):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
            self.iteration += 1
            lrs = self.learning_rates(self.iteration)
            self.engine.set_active(ng.P)
            if self.fuse_projection:
                self.engine.check(raise_on_overflow=False)
            fused = self.fuse_projection and self.engine.can_take_projected()
            ev[0].record()
            radii = None if fused else self.prefilter_voxel(kf)
            ev[1].record()
            if fused:
                # (prefilter and per-Gaussian projection then count as neural_forward: they run inside its kernels)
                ng.forward_projected(kf, self.visible_radii, self.engine, anchor_rotations=self._anchor_rotations())
                ev[2].record()
                image = self.engine.forward_projected(self.bg, ng.means3D, ng.scales, ng.rotations, kf.view, kf.proj, kf.campos,
                                                      kf.tanfovx, kf.tanfovy)
            else:
                ng.forward(kf.campos, kf.pose7, radii)
                ev[2].record()
                image = self.engine.forward(self.bg, ng.means3D, ng.colors, ng.opacity, ng.scales, ng.rotations, kf.view, kf.proj,
                                            kf.campos, kf.tanfovx, kf.tanfovy)
            ev[3].record()
            loss, dL = self.loss_fn(image, gt)
            ev[4].record()
            if self.freq_reg is not None:       # the mapper's frequency regulariser, when its iteration window is open
                if self.freq_reg["fused"]:
                    self._freq_fused_add(image, gt, dL, loss.view(1))
                else:
                    _, fg = self._freq_grad(image, gt)
                    dL = dL if fg is None else dL + fg
            ev[5].record()
            g = self.engine.backward(dL)
            ev[6].record()
            ng.backward(g["means3D"], g["colors"], g["opacity"], g["scales"], g["rotations"], self.scaling_reg_weight)
            ev[7].record()
            groups = self.model.adam_groups(lrs)
            if self._anchor_count is None:
                self._adam(groups, self._mlp_count, None)
            else:
                self._adam(groups[:4], self._anchor_count, None)
                self._adam(groups[4:], self._mlp_count, None)
            ev[8].record()
            torch.cuda.synchronize(self.model.device)
            for i, n in enumerate(names):
                tot[n] += ev[i].elapsed_time(ev[i + 1])
        return {n: v / iters for n, v in tot.items()}

    def _exchange_sharded(self) -> bool:
        return self._exchange().sharded

    def dropped_steps(self) -> int:
        """Iterations the device dropped so far (overflowed resident capacity on some rank; synchronises).  Their returned
        loss is invalid and their keyframe use was still counted; the reference never drops an iteration."""
        return self._mlp_count.dropped()

    def _exchange(self):
        """The step's BucketExchange over the model's flat bucket (rebuilt when densification re-sized the bucket)."""
        from .keyframe_parallel import BucketExchange
        ex = getattr(self, "_ex", None)
        total = self.model.params.numel()
        # Frozen anchor positions (Optimization.position_lr_init = position_lr_final = 0: the Replica and TUM configurations,
        # SURVEY 5.6) head the bucket and nobody ever applies their gradient: they stay out of the exchange -- 3 of the 71
        # floats per anchor, 4.2 % of the bytes on the links.  (The offset is rounded down to the exchange's 16-byte grain.)
        off = (self.model.seg_offset["offset"] // BucketExchange.ALIGN) * BucketExchange.ALIGN if self.anchors_frozen() else 0
        stale = ex is None or ex.offset != off or ex.offset + ex.n != total or \
            (ex._ext is not None and ex._ext.data_ptr() != self.model.grads[off:].data_ptr())
        if stale:
            ex = self._ex = BucketExchange(total - off, self.model.device, self.pg, sharded=self.sharded_optimizer,
                                           single_rank_collectives=self.single_rank_collectives, grads=self.model.grads, offset=off)
        return ex

    def anchors_frozen(self) -> bool:
        o = self.opt
        return o.position_lr_init == 0.0 and o.position_lr_final == 0.0


def init_mlps(dims: ModelDims, generator: torch.Generator) -> Dict[str, torch.Tensor]:
    """torch::nn::Linear's default initialisation (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases) of the
    Sequential MLPs built in GaussianModel's constructor (src/gaussian_model.cpp:60-150).  The reference draws from
    LibTorch's global generator; any seeded stream of the same distribution stands for it."""
    import math
    mlp = {}
    for name in dims.mlp_tensor_names():
        shape = dims.mlp_tensor_shape(name)
        fan_in = shape[1] if len(shape) == 2 else dims.mlp_tensor_shape(name.replace(".bias", ".weight"))[1]
        mlp[name] = (torch.rand(*shape, generator=generator) * 2 - 1) / math.sqrt(fan_in)
    return mlp


def anchors_from_points(points: torch.Tensor, voxel_size: float, place_size: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Shared head of createFromPcd / increasePcd (src/gaussian_model.cpp:343-361, 455-470): voxel centres
    unique_dim(round(points / voxel_size)) * voxel_size in lexicographic order, and log(sqrt(clamp_min(distCUDA2, 1e-7)))
    repeated over the 6 scaling columns.  `points` (N,3) float32 on the GPU.  `place_size`: the size the unique voxel indices
    are multiplied by when it is not the size they were rounded at (createCoarseAnchorFromPcd, :290-291)."""
    from .points import distCUDA2
    fused = (torch.unique(torch.round(points / voxel_size), dim=0, sorted=True) * (voxel_size if place_size is None else place_size)).to(torch.float32).contiguous()
    dist2 = torch.clamp_min(distCUDA2(fused), 0.0000001)
    scaling = torch.log(torch.sqrt(dist2)).unsqueeze(1).repeat(1, 6)
    return fused, scaling


def create_from_pcd(points: torch.Tensor, dims: ModelDims, voxel_size: float, device, capacity: Optional[int] = None,
                    mlp_seed: int = 0, coarse=None) -> ScaffoldModel:
    """GaussianModel::createFromPcd (src/gaussian_model.cpp:327-381): anchors at the occupied voxels of the point cloud,
    zero offsets and features, isotropic log-scales from simple-knn, identity rotations, opacity inverse_sigmoid(0.1),
    freshly initialised MLPs.  `coarse` (coarse_anchors.CoarseParams; Model.use_coarse_anchor = 1): the coarse anchor set is
    created from the same points and attached as `model.coarse` (:379-380)."""
    points = points.to(device, torch.float32)
    anchor, scaling = anchors_from_points(points, voxel_size)
    A = anchor.shape[0]
    model = ScaffoldModel(A, dims, device, capacity)
    zeros = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)  # noqa: E731
    model.load(anchor, zeros(A, dims.n_offsets, 3), zeros(A, dims.feat_dim), scaling,
               init_mlps(dims, torch.Generator().manual_seed(0x5E65 + mlp_seed)))
    x = 0.1 * torch.ones((A, 1), dtype=torch.float32, device=device)
    model.opacity[:A] = torch.log(x / (1 - x))          # general_utils::inverse_sigmoid
    if coarse is not None:
        from .coarse_anchors import CoarseAnchors
        model.coarse = CoarseAnchors.create_from_pcd(points, coarse, voxel_size, dims, device, mlp_seed)
    return model


def synthetic_model(A: int, dims: ModelDims, cam, device, seed: int = 0) -> ScaffoldModel:
    """Seeded synthetic anchors inside the frustum of `cam` (a scenes.Camera at the origin looking down +z) with
    torch::nn::Linear-style uniform MLP init: the mapper-loop workload of SURVEY 8d config 3 (no dataset in the image).
    Anchor depth ~ U(1, 6) m, lateral position within 1.1x the field of view, per-anchor Gaussian scales
    ~ exp(U(ln 0.006, ln 0.05)) (halved by the sigmoid in the forward), offsets ~ N(0, 0.5) voxels of 5 cm."""
    import math
    g = torch.Generator().manual_seed(0x5E65 + seed)
    r = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    z = 1.0 + 5.0 * r(A)
    x = (r(A) * 2 - 1) * 1.1 * cam.tanfovx * z
    y = (r(A) * 2 - 1) * 1.1 * cam.tanfovy * z
    anchor = torch.stack([x, y, z], dim=1)
    offset = 0.5 * torch.randn(A, dims.n_offsets, 3, generator=g)
    feat = 0.5 * torch.randn(A, dims.feat_dim, generator=g)
    scaling_log = torch.cat([torch.full((A, 3), math.log(0.05)),
                             math.log(0.006) + (math.log(0.05) - math.log(0.006)) * r(A, 3)], dim=1)
    model = ScaffoldModel(A, dims, device)
    mlp = {}
    for name in dims.mlp_tensor_names():
        shape = dims.mlp_tensor_shape(name)
        fan_in = shape[1] if len(shape) == 2 else dims.feat_dim
        mlp[name] = (r(*shape) * 2 - 1) / math.sqrt(fan_in)
    model.load(anchor, offset, feat, scaling_log, mlp)
    return model
