import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scenes():
    from raytracer_amd import scene as S

    return {name: S.load_scene(S.scene_path(name)) for name in ("spheres", "boxes", "suzanne")}
