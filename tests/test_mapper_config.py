"""segs_slam_amd.mapper_config: the reference's OpenCV-FileStorage YAML configuration (src/gaussian_mapper.cpp:224-520)."""
import glob
import os

import numpy as np
import pytest
import torch

from segs_slam_amd import mapper_config as mc

# written for this test in the reference's format: directive line, flat dotted keys, trailing comments (also non-ASCII),
# integer booleans, a duplicated key, a key the reader maps that is absent (Mapper.lambda_frequency_low)
SAMPLE = """%YAML:1.0

#--------------------------------------------------------------------------------------------
Model.white_background: 1  # 0:false, 1 or other integer:true
Model.feat_dim: 32
Model.n_offsets: 10
Model.voxel_size:  0.02 # if voxel_size<=0, using 1nn dist
Model.update_depth: 2
Model.update_init_factor: 8
Model.update_hierachy_factor: 4
Model.use_feat_bank: 0 #False
Model.appearance_dim: 16 #32
Model.add_opacity_dist: 0
Model.add_cov_dist: 0
Model.add_color_dist: 1
Camera.z_near: 0.01
Camera.z_far: 100.0
Optimization.max_num_iterations: 7000
Optimization.position_lr_init: 0.0 #0.00016
Optimization.position_lr_final: 0.0
Optimization.position_lr_max_steps: 30000
Optimization.feature_lr: 0.0075
Optimization.scaling_lr: 0.007 #这个参数调大会掉点
Optimization.lambda_dssim: 0.25
Optimization.densify_grad_threshold: 0.001
Optimization.offset_lr_init: 0.01
Optimization.offset_lr_final: 0.0001
Optimization.offset_lr_max_steps: 30000
Optimization.mlp_opacity_lr_init: 0.002
Optimization.mlp_opacity_lr_final: 0.00002
Optimization.mlp_opacity_lr_max_steps: 30000
Optimization.mlp_cov_lr_init: 0.004
Optimization.mlp_cov_lr_final: 0.004
Optimization.mlp_cov_lr_max_steps: 30000
Optimization.mlp_color_lr_init: 0.008
Optimization.mlp_color_lr_final: 0.00005
Optimization.mlp_color_lr_max_steps: 30000
Optimization.mlp_featurebank_lr_init: 0.01
Optimization.mlp_featurebank_lr_final: 0.00001
Optimization.mlp_featurebank_lr_max_steps: 30000
Optimization.appearance_lr_init: 0.05
Optimization.appearance_lr_final: 0.0005
Optimization.appearance_lr_max_steps: 30000
Optimization.start_stat: 2
Optimization.update_from: 4
Optimization.update_interval: 5
Optimization.update_until: 1000
Optimization.min_opacity: 0.005
Optimization.success_threshold: 0.8
Optimization.densify_grad_threshold: 0.0002
Mapper.use_frequency_regularization: 1
Mapper.use_multi_resolution: 1
Mapper.scale_num: 3
Mapper.frequency_regulization_until: 25500
Mapper.high_frequency_regularization_start: 3
Mapper.lambda_frequency_high: 0.01
Model.use_coarse_anchor: 0 #0(fasle)
"""


def _write(tmp_path, text=SAMPLE):
    p = tmp_path / "mapper.yaml"
    p.write_text(text, encoding="utf-8")
    return str(p)


def test_reads_the_reference_format(tmp_path):
    cfg = mc.load_mapper_config(_write(tmp_path))
    assert cfg.white_background is True and cfg.max_num_iterations == 7000
    assert (cfg.model.feat_dim, cfg.model.n_offsets, cfg.model.appearance_dim) == (32, 10, 16)
    assert cfg.model.use_feat_bank is False and cfg.model.add_color_dist is True and cfg.model.add_cov_dist is False
    assert cfg.opt.lambda_dssim == 0.25 and cfg.opt.scaling_lr == 0.007 and cfg.opt.offset_lr_max_steps == 30000
    assert cfg.opt.eps == 1e-15 and cfg.opt.beta1 == 0.9                      # not configurable (gaussian_model.cpp:632-634)
    assert cfg.densify.voxel_size == 0.02 and cfg.densify.update_depth == 2 and cfg.densify.update_interval == 5
    assert cfg.densify.densify_grad_threshold == 0.001                         # duplicated key: first occurrence (OpenCV 4 lookup)
    assert mc.load_mapper_config(_write(tmp_path), duplicates="last").densify.densify_grad_threshold == 0.0002
    assert cfg.lambda_frequency_low == 0.0                                     # absent key reads as 0, like an empty cv::FileNode
    assert cfg.scales == (1.0, 0.5, 0.25)
    assert cfg.raw["Model.white_background"] == 1 and "Model.lowpoly" not in cfg.raw


def test_rejects_what_the_format_does_not_contain(tmp_path):
    with pytest.raises(ValueError):
        mc.read_opencv_yaml(_write(tmp_path, "%YAML:1.0\nModel:\n  feat_dim: 32\n"))
    with pytest.raises(ValueError):
        mc.read_opencv_yaml(_write(tmp_path, "%YAML:1.0\njust some text\n"))
    with pytest.raises(ValueError):
        mc.load_mapper_config(_write(tmp_path, "%YAML:1.0\nModel.feat_dim: thirty-two\n"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/cfg/gaussian_mapper"), reason="reference tree not present")
def test_every_shipped_configuration_parses():
    files = sorted(glob.glob("/root/reference/cfg/gaussian_mapper/**/*.yaml", recursive=True))
    assert len(files) > 10
    anchor_cfgs = 0
    for f in files:
        cfg = mc.load_mapper_config(f)
        if "Model.feat_dim" not in cfg.raw:      # files inherited from Photo-SLAM without the anchor keys (e.g. Monocular/ETH3D)
            assert cfg.model.feat_dim == 0
            continue
        anchor_cfgs += 1
        assert cfg.model.feat_dim == 32 and cfg.model.n_offsets == 10, f      # what csrc/neural.hip is compiled for
        assert cfg.opt.lambda_dssim > 0 and cfg.densify.update_interval > 0, f
    assert anchor_cfgs > 10
    cfg = mc.load_mapper_config("/root/reference/cfg/gaussian_mapper/RGB-D/Replica/office0.yaml")
    assert cfg.model.appearance_dim == 32 and cfg.model.use_feat_bank and cfg.opt.offset_lr_init == 0.08
    assert (cfg.densify.start_stat, cfg.densify.update_from, cfg.densify.update_until) == (500, 1500, 25500)
    assert cfg.use_frequency_regularization and cfg.scale_num == 3 and cfg.lambda_frequency_high == 0.01


@pytest.mark.gpu
def test_mapper_step_from_configuration_applies_the_row_mask(tmp_path):
    from segs_slam_amd import neural_gaussians as ng, scenes
    dev = torch.device("cuda:0")
    cfg = mc.load_mapper_config(_write(tmp_path))
    cam = scenes.make_camera(160, 96, 150.0, 150.0, np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32))
    model = ng.synthetic_model(1500, cfg.model, cam, dev, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    step = mc.make_mapper_step(cfg, model, cam.width, cam.height)
    assert step.row_mask and float(step.bg.min()) == 1.0 and step.scaling_reg_weight == 0.01 and step.freq_reg["scales"] == (1.0, 0.5, 0.25)
    gt = torch.rand(3, cam.height, cam.width, device=dev) * 0.5 + 0.25
    gt[:, :6, :] = 0.0                     # six rows without target data in every channel
    gt[1, 40, :] = 0.0                     # and one row of one channel
    seen = []
    real_backward = step.engine.backward
    step.engine.backward = lambda dL: (seen.append(dL.clone()), real_backward(dL))[1]
    for _ in range(8):                     # crosses high_frequency_regularization_start and a densify iteration
        loss = step.training_once([kf], [gt])
        assert np.isfinite(float(loss))
    for dL in seen:
        assert float(dL[:, :6, :].abs().max()) == 0.0 and float(dL[1, 40, :].abs().max()) == 0.0
        assert float(dL[0, 40, :].abs().max()) > 0.0 and float(dL[:, 6:, :].abs().max()) > 0.0
    assert len(step._row_mask_cache) == 1


def test_coarse_anchor_configuration_is_parsed_and_priced():
    """Model.use_coarse_anchor = 1 (cfg/colmap/gaussian_splatting.yaml:141-185): the `_coarse` keys, the shapes of the five coarse
    Sequentials (src/gaussian_model.cpp:112-148) and the learning rates of their optimizer groups (:686-787, 917-982)."""
    from segs_slam_amd import coarse_anchors as ca, mapper_config as mc
    cfg = mc.load_committed_config("cfg/colmap/gaussian_splatting.yaml")
    assert cfg.use_coarse_anchor and cfg.coarse is not None
    cp = cfg.coarse
    assert (cp.feat_dim_coarse, cp.n_offsets_coarse, cp.appearance_dim_coarse) == (32, 10, 32) and abs(cp.coarse_voxel_size - 0.2) < 1e-12
    assert cp.offset_lr_init_coarse == 0.01 and cp.mlp_color_lr_final_coarse == 0.00005 and cp.appearance_lr_max_steps_coarse == 30000
    shapes = ca.coarse_mlp_shapes(cfg.model, cp)
    assert shapes["mlp_opacity_c.0.weight"] == (32, 35) and shapes["mlp_opacity_c.2.weight"] == (10, 32)
    assert shapes["mlp_cov_c.2.weight"] == (70, 32) and shapes["mlp_color_c.0.weight"] == (32, 35 + 32) and shapes["mlp_color_c.2.bias"] == (30,)
    assert shapes["mlp_apperance_c.0.weight"] == (32, 7) and not any(k.startswith("mlp_feature_bank_c") for k in shapes)
    assert ca.expon_lr(0, 0.01, 0.0001, 0.01, 30000) == pytest.approx(0.01) and ca.expon_lr(30000, 0.01, 0.0001, 0.01, 30000) == pytest.approx(0.0001)
    assert ca.expon_lr(15000, 0.01, 0.0001, 0.01, 30000) == pytest.approx(0.001) and ca.expon_lr(100, 0.0, 0.0, 0.01, 30000) == 0.0
    # the RGB-D configurations leave it off, and a step for a configuration with it on wants the coarse set made with the model
    assert mc.load_committed_config("cfg/gaussian_mapper/RGB-D/Replica/office0.yaml").coarse is None

    class _NoCoarse:
        coarse = None
    with pytest.raises(ValueError, match="use_coarse_anchor"):
        mc.make_mapper_step(cfg, _NoCoarse(), 64, 48)
