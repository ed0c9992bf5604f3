"""Reader for the reference's Gaussian-mapper configuration files (cfg/gaussian_mapper/**/*.yaml).

Mirrors GaussianMapper::readConfigFromFile (src/gaussian_mapper.cpp:224-520): the files are OpenCV FileStorage YAML 1.0
-- a `%YAML:1.0` directive followed by flat `Section.key: value  # comment` lines -- and every key is fetched with
`settings_file["..."].operator int()/float()`.  Semantics kept:
  * booleans are integers compared with 0 (`!= 0`, e.g. :240);
  * a key that is absent reads as 0 (an empty cv::FileNode converts to 0), not as our dataclass default;
  * a key that appears twice (the shipped Replica files define Optimization.densify_grad_threshold at lines 91 and 137)
    resolves to its FIRST occurrence: OpenCV 4's FileNode::operator[] scans the mapping in file order and returns the
    first match.  OpenCV is not installed in this image, so this rule is restated from its source, not executed;
    `duplicates="last"` selects the other reading.
Only the keys that shape the hot path (model dimensions, optimisation schedule, densification, loss terms) are mapped to
typed objects; the rest (viewer, recording, SLAM front-end keys) stays available in `MapperConfig.raw`.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field, fields
from typing import Dict, Optional, Tuple, Union

from .densify import DensifyParams
from .neural_gaussians import ModelDims, ScaffoldOptimizationParams

Scalar = Union[int, float, str]
_LINE = re.compile(r"^([A-Za-z_][\w.]*)\s*:\s*(.*)$")


def _scalar(text: str) -> Scalar:
    text = text.strip()
    if len(text) >= 2 and text[0] == text[-1] and text[0] in "\"'":
        return text[1:-1]
    try:
        return int(text)
    except ValueError:
        pass
    try:
        return float(text)
    except ValueError:
        return text


def read_opencv_yaml(path: str, duplicates: str = "first") -> Dict[str, Scalar]:
    """Flat scalar mapping of an OpenCV FileStorage YAML file.  Nested mappings / sequences (unused by the mapper's
    configuration files) are rejected rather than mis-read."""
    if duplicates not in ("first", "last"):
        raise ValueError("duplicates must be 'first' or 'last'")
    out: Dict[str, Scalar] = {}
    with open(path, "r", encoding="utf-8") as f:
        for n, raw in enumerate(f, 1):
            line = raw.split("#", 1)[0].rstrip()
            if not line.strip() or line.startswith("%") or line.strip() == "---":
                continue
            if line[0] in " \t-":
                raise ValueError(f"{path}:{n}: nested YAML is not part of the mapper configuration format")
            m = _LINE.match(line)
            if not m or m.group(2).strip() == "":
                raise ValueError(f"{path}:{n}: expected `key: value`, got {raw.strip()!r}")
            key, val = m.group(1), _scalar(m.group(2))
            if key in out and duplicates == "first":
                continue
            out[key] = val
    return out


@dataclass
class MapperConfig:
    model: ModelDims
    opt: ScaffoldOptimizationParams
    densify: DensifyParams
    white_background: bool = False
    z_near: float = 0.01
    z_far: float = 100.0
    max_num_iterations: int = 30000
    use_frequency_regularization: bool = False
    use_multi_resolution: bool = False
    scale_num: int = 0
    frequency_regulization_until: int = 0
    high_frequency_regularization_start: int = 0
    lambda_frequency_high: float = 0.0
    lambda_frequency_low: float = 0.0
    use_coarse_anchor: bool = False
    coarse: Optional["CoarseParams"] = None          # Model.*_coarse / Optimization.*_coarse when use_coarse_anchor (coarse_anchors.py)
    raw: Dict[str, Scalar] = field(default_factory=dict)

    @property
    def scales(self) -> Tuple[float, ...]:
        """scales[i] = 1 / 2^i, i < Mapper.scale_num (src/gaussian_mapper.cpp:514-517)."""
        return tuple(1.0 / (2 ** i) for i in range(self.scale_num))


def load_mapper_config(path: str, duplicates: str = "first") -> MapperConfig:
    return mapper_config_from_values(read_opencv_yaml(path, duplicates), path)


def mapper_config_from_values(raw: Dict[str, Scalar], path: str = "<values>") -> MapperConfig:
    """The typed configuration from a flat key -> value mapping: what read_opencv_yaml returns, or a committed extract of a
    reference configuration file (segs-slam_amd/data/mapper_cfgs.json) where the reference tree itself is absent (the GPU box)."""
    raw = dict(raw)

    def num(key, cast):
        v = raw.get(key, 0)                     # absent node -> 0, like cv::FileNode
        if isinstance(v, str):
            raise ValueError(f"{path}: {key} is not numeric: {v!r}")
        return cast(v)

    I = lambda k: num(k, int)                   # noqa: E731,E741
    F = lambda k: num(k, float)                 # noqa: E731
    B = lambda k: num(k, int) != 0              # noqa: E731

    model = ModelDims(feat_dim=I("Model.feat_dim"), n_offsets=I("Model.n_offsets"), appearance_dim=I("Model.appearance_dim"),
                      use_feat_bank=B("Model.use_feat_bank"), add_opacity_dist=B("Model.add_opacity_dist"),
                      add_cov_dist=B("Model.add_cov_dist"), add_color_dist=B("Model.add_color_dist"))
    opt = ScaffoldOptimizationParams()
    for f_ in fields(ScaffoldOptimizationParams):
        key = "Optimization." + f_.name
        if f_.name in ("beta1", "beta2", "eps"):     # fixed in GaussianModel::trainingSetup (src/gaussian_model.cpp:632-634)
            continue
        setattr(opt, f_.name, I(key) if f_.type in (int, "int") else F(key))
    dens = DensifyParams(voxel_size=F("Model.voxel_size"), update_depth=I("Model.update_depth"),
                         update_init_factor=I("Model.update_init_factor"), update_hierachy_factor=I("Model.update_hierachy_factor"),
                         start_stat=I("Optimization.start_stat"), update_from=I("Optimization.update_from"),
                         update_interval=I("Optimization.update_interval"), update_until=I("Optimization.update_until"),
                         min_opacity=F("Optimization.min_opacity"), success_threshold=F("Optimization.success_threshold"),
                         densify_grad_threshold=F("Optimization.densify_grad_threshold"))
    coarse = None
    if B("Model.use_coarse_anchor"):                  # src/gaussian_mapper.cpp:433-491
        from .coarse_anchors import CoarseParams
        coarse = CoarseParams()
        for f_ in fields(CoarseParams):
            key = ("Model." if f_.name in ("feat_dim_coarse", "n_offsets_coarse", "coarse_voxel_size", "appearance_dim_coarse") else "Optimization.") + f_.name
            setattr(coarse, f_.name, I(key) if f_.type in (int, "int") else F(key))
    return MapperConfig(model=model, opt=opt, densify=dens, coarse=coarse, white_background=B("Model.white_background"),
                        z_near=F("Camera.z_near"), z_far=F("Camera.z_far"), max_num_iterations=I("Optimization.max_num_iterations"),
                        use_frequency_regularization=B("Mapper.use_frequency_regularization"),
                        use_multi_resolution=B("Mapper.use_multi_resolution"), scale_num=I("Mapper.scale_num"),
                        frequency_regulization_until=I("Mapper.frequency_regulization_until"),
                        high_frequency_regularization_start=I("Mapper.high_frequency_regularization_start"),
                        lambda_frequency_high=F("Mapper.lambda_frequency_high"), lambda_frequency_low=F("Mapper.lambda_frequency_low"),
                        use_coarse_anchor=B("Model.use_coarse_anchor"), raw=raw)


def load_committed_config(rel_path: str) -> MapperConfig:
    """A reference configuration by its path inside the reference tree (e.g. "cfg/gaussian_mapper/RGB-D/Replica/office0.yaml"),
    from the committed value extract segs-slam_amd/data/mapper_cfgs.json (made by tests/golden/make_cfg_golden.py)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mapper_cfgs.json")) as f:
        table = json.load(f)
    if rel_path not in table:
        raise KeyError(f"{rel_path} is not in segs-slam_amd/data/mapper_cfgs.json (has: {sorted(table)})")
    return mapper_config_from_values(table[rel_path], rel_path)


def make_mapper_step(cfg: MapperConfig, model, width: int, height: int, spatial_lr_scale: float = 1.0, process_group=None,
                     densify_seed: int = 0):
    """The anchor-level step configured like GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:823-1032) for
    this configuration: mapper loss (0.01 * scaling regulariser :926-928, row mask :917-922, FFT regularisers :930-945),
    densification schedule :961-968, background :61-67."""
    from .densify import AnchorDensifier
    from .neural_gaussians import ScaffoldTrainerStep
    if cfg.use_coarse_anchor and getattr(model, "coarse", None) is None:
        # the coarse set is made from the point cloud together with the fine one (src/gaussian_model.cpp:379-380)
        raise ValueError("Model.use_coarse_anchor = 1: build the model with neural_gaussians.create_from_pcd(..., coarse=cfg.coarse)")
    step = ScaffoldTrainerStep(model, width, height, cfg.opt, spatial_lr_scale, process_group, scaling_reg_weight=0.01)
    step.row_mask = True
    step.set_background(cfg.white_background)
    if cfg.use_frequency_regularization:
        step.enable_frequency_regularization(cfg.lambda_frequency_high, cfg.lambda_frequency_low, cfg.scales,
                                             cfg.high_frequency_regularization_start, cfg.frequency_regulization_until,
                                             cfg.use_multi_resolution)
    step.enable_densification(AnchorDensifier(model, cfg.densify), seed=densify_seed)
    return step
