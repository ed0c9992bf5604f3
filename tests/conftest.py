import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_scene():
    from segs_slam_amd import scenes
    return scenes.make_scene(1000, 64, 64, 60.0, 60.0, seed=1234, bg=(0.1, 0.2, 0.3), name="small")
