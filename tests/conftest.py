import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

SCENES = ["two_spheres", "cornell", "globe", "blocks_ball", "knight", "blue_ball"]
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name: str) -> str:
    return os.path.join(ROOT, "examples", name + ".json5")


@pytest.fixture(scope="session", autouse=True)
def _knight_obj():
    """examples/knight.obj is generated (the reference asset is missing, see scripts/gen_knight.py)."""
    import gen_knight

    gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import binding

    binding.lib()
    return binding


@pytest.fixture(scope="session")
def rt():
    """The product bindings; loading fails loudly if librt_mi355x.so is not built."""
    import raytracer_amd

    raytracer_amd.load_library()
    return raytracer_amd


@pytest.fixture(scope="session")
def oracle_scenes(orc):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = orc.OracleScene.load(scene_path(name), ROOT)
        return cache[name]

    return get


@pytest.fixture(scope="session")
def gpu_scenes(rt):
    """(Scene, Renderer) per example scene on HIP device 0, f64 arithmetic."""
    cache = {}

    def get(name, precision=0):
        key = (name, precision)
        if key not in cache:
            sc = rt.Scene.load(scene_path(name), ROOT)
            r = rt.Renderer(device=0, precision=precision)
            r.upload(sc)
            cache[key] = (sc, r)
        return cache[key]

    return get
