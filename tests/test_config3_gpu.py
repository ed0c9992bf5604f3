"""SURVEY 8d config 3 at its own size: the mapper loop (statistics, anchor growing, pruning, capacity growth, Adam-state
migration) on the synthetic 64-keyframe orbit from ~50 k anchors until anchors x 10 ~ 2 M Gaussians (segs_slam_amd.config3).
No oracle finishes this size in seconds, so the checks are the size-independent properties of the loop:
  * the map really grows, and past every adjust_anchor: rows <= capacity, every parameter / moment of the live rows finite,
    the Adam moments of appended rows start at zero, statistics rows reset for new anchors;
  * after each adjust_anchor the resident (no-host-sync, tight-binning) render of a keyframe equals the reference-shaped
    synchronising render of the same neural Gaussians BIT FOR BIT (the engine survives re-sized buffers and a changed
    active-row count);
  * no iteration was dropped on the device, and the loss goes down over the run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_config3_growth_run_to_two_million_gaussians():
    from segs_slam_amd import config3, densify
    from segs_slam_amd.raster_engine import RasterEngine
    params = densify.DensifyParams(start_stat=100, update_from=300, update_interval=100, update_until=10 ** 9,
                                   densify_grad_threshold=2e-5)   # (2e-4, the Replica value, stops near 64 k anchors on this scene)
    run = config3.Config3Run(DEV, params=params)
    assert 45_000 <= run.model.A <= 50_000
    seen = []
    first_loss = float(run.step.training_once(run.keyframes, run.targets))

    def after_adjust(r, it):
        m, st = r.model, r.step
        A = m.A
        assert 0 < A <= m.capacity
        for name in ("anchor", "offset", "anchor_feat", "scaling"):
            assert bool(torch.isfinite(m.param(name)).all()), (it, name)
        assert bool(torch.isfinite(m.mlp_params).all())
        for bucket in (m.exp_avg, m.exp_avg_sq):
            assert bool(torch.isfinite(bucket).all())
        if seen and A > seen[-1][1]:
            # rows appended by this adjust_anchor: moments zero (src/gaussian_model.cpp:1663-1687), statistics zero
            lo, hi = seen[-1][1], A
            grown_only = r.history and False
            o = m.seg_offset["anchor_feat"]
            w = m.widths["anchor_feat"]
            tail = m.exp_avg[o + (hi - 1) * w:o + hi * w]
            assert bool((tail == 0).all()), it
            assert float(r.densifier.stat("anchor_demon")[hi - 1]) == 0.0
        # resident render == synchronising render of the same neural Gaussians
        kf = r.keyframes[it % len(r.keyframes)]
        st.fuse_projection = False   # the comparison below reads the candidate arrays the unfused forward writes
        for _ in range(2):      # (the first forward after a re-size calibrates through the synchronising path)
            img = st.render(kf).clone()
        st.fuse_projection = True
        assert torch.equal(st.render(kf), img), it
        assert st.engine.check() and st.engine._last_resident
        ng = st.neural
        ref = RasterEngine(ng.P_capacity, r.cam.width, r.cam.height, DEV, resident=False, skip_nonpositive_opacity=True)
        ref.set_active(ng.P)
        img_ref = ref.forward(st.bg, ng.means3D, ng.colors, ng.opacity, ng.scales, ng.rotations, kf.view, kf.proj, kf.campos,
                              kf.tanfovx, kf.tanfovy)
        assert torch.equal(img, img_ref), it
        assert 0 < st.engine.R <= ref.R
        del ref
        seen.append((it, A))

    res = run.run(3000, 200_000, after_adjust=after_adjust)
    assert res["anchors_end"] >= 200_000 and res["gaussians_end"] >= 2_000_000, res
    assert res["dropped_steps"] == 0
    sizes = [a for _, a in seen]
    assert len(sizes) >= 3 and sizes[-1] > 3 * run.anchors_start, sizes
    assert all(b >= a for a, b in zip(sizes, sizes[1:])), sizes        # this scene only grows until the target is reached
    assert res["final_loss"] < first_loss
    assert bool(torch.isfinite(run.model.params).all())
