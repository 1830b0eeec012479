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


def assert_image_parity(sums, want_sums, rays, want_rays, spp, max_bounces, noise=None, label=""):
    """The bar of every GPU-vs-oracle image test.  Measured on MI355X (scripts/measure_parity.py, profiles/r02/parity.txt):
    on all six example scenes EVERY pixel's f64 sums equal the oracle's to <= 1e-13 relative and the ray counts are
    identical - the GPU walks the oracle's paths.  What may ever differ is a path whose decision (a rejection draw, a hit
    at an edge) falls within an ulp of its threshold and flips: such a pixel differs by whole paths.  So:
      (i)   at most max(1, 0.05 %) of the pixels may differ at all (rtol 1e-9), and the measured fraction is reported;
      (ii)  a pixel that differs must differ by no more than the oracle's own sample noise: `noise()` returns the
            oracle's sums for another seed, and |dL| must stay within the largest two-seed difference of the image;
      (iii) the ray counts may differ by at most max_bounces per differing pixel (a flipped path's length)."""
    import numpy as np

    close = np.isclose(sums, want_sums, rtol=1e-9, atol=1e-12).all(axis=-1)
    n_bad = int((~close).sum())
    budget = max(1, int(0.0005 * close.size))
    assert n_bad <= budget, f"{label}: {n_bad} of {close.size} pixels differ from the oracle (measured agreement {close.mean():.5f}; allowed {budget})"
    assert abs(int(rays) - int(want_rays)) <= n_bad * max_bounces, f"{label}: ray counts {rays} vs {want_rays} with {n_bad} differing pixels"
    if n_bad:
        lum = np.array([0.2126, 0.7152, 0.0722])
        dl = np.abs((sums - want_sums) @ lum)[~close] / spp
        bound = np.abs(want_sums @ lum).max() / spp  # one whole sample of the brightest pixel
        if noise is not None:
            bound = np.abs((noise() - want_sums) @ lum).max() / spp
        assert dl.max() <= bound, f"{label}: a differing pixel is off by {dl.max():.4g} > {bound:.4g}"
    return close.mean()
