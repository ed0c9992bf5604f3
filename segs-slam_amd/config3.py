"""SURVEY 8d config 3 as a reproducible run: the mapper loop on a Scaffold model that GROWS -- from ~50 k anchors x 10
offsets towards ~2 M Gaussians -- through training_statis / adjust_anchor (src/gaussian_mapper.cpp:957-968,
src/gaussian_model.cpp:1459-1762) on a synthetic 64-keyframe orbit.

No dataset ships with the image, so the run is a teacher / student restatement of what the reference's mapper sees:
  * teacher = the explicit 500 k-Gaussian scene of config 2 (scenes.make_config_scene("c2")) rendered by the HIP rasterizer from
    64 seeded keyframe poses (<= 5 degrees, +-0.1 m around the scene's camera): the "RGB frames";
  * student = GaussianModel::createFromPcd (neural_gaussians.create_from_pcd) on a sparse sample of the teacher's visible
    centres (the role of the back-projected depth points): anchors at the occupied voxels, zero offsets and features, fresh
    MLPs -- a map that under-fits the frames, which is what makes anchor_growing add anchors;
  * the mapper's keyframe walk (keyframe_window.SlidingWindowKeyframes), L1 + SSIM + 0.01 scaling regulariser, the Replica
    densification hyper-parameters (densify.DensifyParams) on a compressed schedule (statistics from iteration 100, growing
    from 300, every 100: the reference's 500 / 1500 / 100 would spend the bounded run waiting).
Used by bench.py (the `config3` block of the bench line), tests/test_config3_gpu.py and tools/soak_scaffold.py."""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import densify, neural_gaussians as ng, scenes
from .keyframe_window import SlidingWindowKeyframes
from .raster_engine import RasterEngine


class Config3Run:
    def __init__(self, device, n_keyframes: int = 64, start_anchors: int = 50_000, teacher: str = "c2",
                 dims: Optional[ng.ModelDims] = None, params: Optional[densify.DensifyParams] = None, seed: int = 0):
        dev = self.device = torch.device(device)
        sc = scenes.make_config_scene(teacher)
        cam = self.cam = sc.camera
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        # ---- teacher frames
        eng = RasterEngine(sc.P, cam.width, cam.height, dev, resident=False)
        bg, m3, col, op, sca, rot = t(sc.bg), t(sc.means3D), t(sc.colors), t(sc.opacity), t(sc.scales), t(sc.rotations)
        self.keyframes: List[ng.Keyframe] = []
        self.targets: List[torch.Tensor] = []
        visible = torch.zeros(sc.P, dtype=torch.bool, device=dev)
        for k in range(n_keyframes):
            camk = scenes.make_config_camera(teacher, keyframe=k)
            view, proj, campos = t(camk.world_view_transform), t(camk.full_proj_transform), t(camk.camera_center)
            img = eng.forward(bg, m3, col, op, sca, rot, view, proj, campos, camk.tanfovx, camk.tanfovy)
            self.targets.append(img.clone())
            visible |= eng.radii > 0
            pose7 = torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev)
            pose7[:3] = campos
            self.keyframes.append(ng.Keyframe(view, proj, campos, pose7, camk.tanfovx, camk.tanfovy))
        del eng
        # ---- student: createFromPcd on a sparse sample of the visible centres
        ids = torch.nonzero(visible).view(-1)
        g = torch.Generator(device="cpu").manual_seed(0x5E65 + seed)
        pick = ids[torch.randperm(ids.numel(), generator=g)[:start_anchors].to(dev)]
        self.params = params or densify.DensifyParams(start_stat=100, update_from=300, update_interval=100, update_until=10 ** 9)
        self.model = ng.create_from_pcd(m3[pick], dims or ng.ModelDims(), self.params.voxel_size, dev, mlp_seed=seed)
        self.step = ng.ScaffoldTrainerStep(self.model, cam.width, cam.height, scaling_reg_weight=0.01)
        self.densifier = densify.AnchorDensifier(self.model, self.params)
        self.step.enable_densification(self.densifier, seed=seed)
        self.step.keyframe_selector = SlidingWindowKeyframes(seed=seed)
        for _ in self.keyframes:
            self.step.keyframe_selector.add_keyframe(8)          # Mapper.new_keyframe_times_of_use
        self.anchors_start = self.model.A
        self.history: List[Dict] = []

    def run(self, max_iters: int, target_anchors: int = 200_000, log: Optional[Callable[[str], None]] = None,
            after_adjust: Optional[Callable[["Config3Run", int], None]] = None) -> Dict:
        """Iterate until the map holds `target_anchors` (x 10 offsets ~ 2 M Gaussians) or `max_iters` are spent; wall clock
        over everything (adjust_anchor iterations and their host synchronisations included)."""
        st, m, p = self.step, self.model, self.params
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        it = 0
        last_A = m.A
        while it < max_iters and m.A < target_anchors:
            loss = st.training_once(self.keyframes, self.targets)
            it += 1
            # (the step's own iteration count decides when adjust_anchor runs: a caller may have stepped it before this loop)
            if st.iteration > p.update_from and st.iteration % p.update_interval == 0:
                if after_adjust is not None:
                    after_adjust(self, it)
                self.history.append({"iteration": it, "anchors": m.A, "s": time.perf_counter() - t0})
                if log:
                    log(f"it {it:5d} anchors {last_A:7d} -> {m.A:7d} capacity {m.capacity:7d} loss {float(loss):.4f}")
                last_A = m.A
        st.finish()        # a drop of the LAST iteration is only seen here (training_once resolves the previous call's word)
        torch.cuda.synchronize(self.device)
        wall = time.perf_counter() - t0
        return {"iterations": it, "seconds": wall, "iters_per_s": it / wall if wall > 0 else 0.0, "anchors_start": self.anchors_start,
                "anchors_end": m.A, "gaussians_end": m.A * m.dims.n_offsets, "dropped_steps": st.dropped_steps(), "redone_steps": st.redone_steps,
                "final_loss": float(loss) if it else None, "anchors_over_time": [(h["iteration"], h["anchors"]) for h in self.history]}
