#!/usr/bin/env python3
"""Extracts the key -> value content of the reference's mapper configuration files into segs-slam_amd/data/mapper_cfgs.json.

Runs only where /root/reference exists.  The fixture holds hyper-parameter VALUES (data) keyed by the file they come from
(cfg/gaussian_mapper/**/*.yaml, read with segs_slam_amd.mapper_config.read_opencv_yaml: first occurrence of a duplicated key,
comments dropped) so that tests and bench.py can build the step a configuration describes on the GPU box, where the reference
tree is absent.  Only the files BASELINE.json's configurations use are extracted."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from segs_slam_amd import mapper_config as mc  # noqa: E402

REF = "/root/reference"
FILES = ["cfg/gaussian_mapper/RGB-D/Replica/office0.yaml",
         "cfg/gaussian_mapper/RGB-D/TUM/tum_freiburg3_long_office_household.yaml",
         "cfg/gaussian_mapper/RGB-D/ScanNet/scannet_rgbd.yaml",
         "cfg/colmap/gaussian_splatting.yaml"]        # (the offline configuration: Model.use_coarse_anchor = 1)
out = {rel: mc.read_opencv_yaml(os.path.join(REF, rel)) for rel in FILES}
with open(os.path.join(ROOT, "segs-slam_amd", "data", "mapper_cfgs.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items()})
