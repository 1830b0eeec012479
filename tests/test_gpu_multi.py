"""rt_multi_*: one image over several GPUs from one process (sample split, one reduce per band; replaces the rayon row
loop of reference src/main.rs:117-154 at node scale).  The box these tests run on has ONE GPU, so the device lists name
GPU 0 once (no reduce) or twice (two contexts on one GPU, summed by the copy-and-add path: RCCL does not allow a device
twice).  The RCCL ncclReduce leg needs distinct GPUs and has not run on hardware: see DESIGN.md section 7."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu

W, SPP, MB = 40, 9, 12


@pytest.fixture(scope="module")
def single(rt):
    scene = rt.Scene.load(scene_path("cornell"), ROOT)
    r = rt.Renderer(device=0)
    r.upload(scene)
    H = scene.image_height(W)
    return scene, H, r.render_rows(W, H, 0, H, SPP, MB, seed=3)


def test_one_device_is_rt_render_rows(rt, single):
    scene, H, (rgb8, sums, st) = single
    m = rt.MultiRenderer([0])
    m.upload(scene)
    got8, got, gst = m.render_rows(W, H, 0, H, SPP, MB, seed=3)
    assert np.array_equal(got, sums) and np.array_equal(got8, rgb8)
    assert gst.rays == st.rays and gst.paths == st.paths


@pytest.mark.parametrize("band_rows", [0, 7])
def test_two_contexts_on_one_gpu_sum_to_the_same_image(rt, single, band_rows):
    scene, H, (rgb8, sums, st) = single
    m = rt.MultiRenderer([0, 0])
    m.upload(scene)
    seen = []
    got8, got, gst = m.render_rows(W, H, 0, H, SPP, MB, seed=3, band_rows=band_rows, on_band=lambda b0, b1: seen.append((b0, b1)))
    # the same samples, added as (4 samples) + (5 samples) instead of 9 in a row: equal up to that one reordering
    assert np.allclose(got, sums, rtol=1e-12, atol=1e-13)
    assert gst.rays == st.rays and gst.paths == st.paths
    assert (got8 != rgb8).mean() <= 0.001 and np.abs(got8.astype(int) - rgb8.astype(int)).max() <= 1
    step = band_rows or H
    assert seen == [(b, min(b + step, H)) for b in range(0, H, step)]   # every row handed over once, in band order
    # a band of rows in the middle of the picture
    part8, part, _ = m.render_rows(W, H, 5, 17, SPP, MB, seed=3, band_rows=5)
    assert np.allclose(part, sums[5:17], rtol=1e-12, atol=1e-13)


def test_fewer_samples_than_devices_splits_rows_and_reduces_nothing(rt, single):
    scene, H, _ = single
    r = rt.Renderer(device=0)
    r.upload(scene)
    want8, want, wst = r.render_rows(W, H, 0, H, 1, MB, seed=9)
    m = rt.MultiRenderer([0, 0])
    m.upload(scene)
    seen = []
    got8, got, gst = m.render_rows(W, H, 0, H, 1, MB, seed=9, band_rows=16, on_band=lambda b0, b1: seen.append((b0, b1)))
    assert np.array_equal(got, want) and np.array_equal(got8, want8)   # no sum is reordered: bit-identical
    assert gst.rays == wst.rays
    assert seen == [(b, min(b + 16, H)) for b in range(0, H, 16)]


def test_cli_with_two_contexts_writes_the_one_gpu_picture(tmp_path):
    cli = os.path.join(ROOT, "bin", "raytracer")
    args = [scene_path("cornell"), "-w", "48", "-s", "8", "-m", "12", "--seed", "5", "-q"]
    a = subprocess.run([cli, args[0], "one.ppm", *args[1:]], cwd=tmp_path, capture_output=True, timeout=110)
    b = subprocess.run([cli, args[0], "two.ppm", *args[1:], "--devices", "0,0", "--band-rows", "10"], cwd=tmp_path, capture_output=True, timeout=110)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    one = np.frombuffer(open(tmp_path / "one.ppm", "rb").read(), dtype=np.uint8)
    two = np.frombuffer(open(tmp_path / "two.ppm", "rb").read(), dtype=np.uint8)
    assert one.shape == two.shape and (one != two).mean() <= 0.001
    assert sorted(os.listdir(tmp_path)) == ["one.ppm", "two.ppm"]   # part files removed


def test_one_rank_rccl_reduce(rt, single, monkeypatch):
    # RT_REDUCE=rccl: RCCL is loaded (dlopen), a one-rank communicator made and ncclReduce(sum, f64) called per band
    import torch  # noqa: F401 - a process that uses both loads torch (and with it torch's RCCL) first: INTEGRATION.md

    scene, H, (rgb8, sums, st) = single
    monkeypatch.setenv("RT_REDUCE", "rccl")
    m = rt.MultiRenderer([0])
    m.upload(scene)
    got8, got, gst = m.render_rows(W, H, 0, H, SPP, MB, seed=3, band_rows=11)
    assert np.array_equal(got, sums) and np.array_equal(got8, rgb8)
    m.close()


def test_bad_device_list(rt):
    with pytest.raises(rt.RtError):
        rt.MultiRenderer([0, 99])
    with pytest.raises(rt.RtError):
        rt.MultiRenderer([])
