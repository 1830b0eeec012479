"""bin/raytracer end to end on the GPU: the command line of reference src/main.rs (row loop,
monitor thread, .PART files, image output) over the C ABI.  The image it writes must be the
library's own render of the same seed (bit-exact) and the oracle's (>= 99 % of the bytes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "raytracer")
SCENE = os.path.join(ROOT, "examples", "cornell.json5")
W = H = 48
SPP, MB = 8, 12


def _cli(*args, cwd):
    return subprocess.run([CLI, *args], capture_output=True, cwd=cwd, timeout=110)


def _read_ppm(path):
    raw = open(path, "rb").read()
    head = b"P6\n%d %d\n255\n" % (W, H)
    assert raw.startswith(head)
    return np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(H, W, 3)


def _render(tmp_path, *extra, seed=5, out="out.ppm"):
    r = _cli(SCENE, out, "-w", str(W), "-s", str(SPP), "-m", str(MB), "--seed", str(seed), *extra, cwd=tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    return _read_ppm(os.path.join(tmp_path, out)), r


def test_cli_image_is_the_library_render_and_the_oracle_render(tmp_path, gpu_scenes, oracle_scenes):
    img, r = _render(tmp_path, "--stats", "--progress-bar-len", "60")
    sc, ren = gpu_scenes("cornell")
    assert sc.image_height(W) == H
    rgb8, _, stats = ren.render_rows(W, H, 0, H, SPP, MB, seed=5)
    assert np.array_equal(img, rgb8[::-1])  # the file has the top row (j = H-1) first
    o8, _, _ = oracle_scenes("cornell").render(W, H, 0, H, SPP, MB, seed=5, threads=8)
    assert (img == o8[::-1]).mean() >= 0.9995  # measured: 1.0
    err = r.stderr.decode()
    assert "Completed in 0:0" in err and "Rendering: [" in err and f"{W * H * SPP} paths" in err
    assert os.listdir(tmp_path) == ["out.ppm"]  # the part file is removed after a complete render


def test_cli_bands_quiet_and_formats(tmp_path):
    whole, _ = _render(tmp_path)
    banded, r = _render(tmp_path, "--band-rows", "7", "-q", "--no-part-file", out="banded.ppm")
    assert np.array_equal(whole, banded)  # counter-based streams: the batching does not change a pixel
    assert r.stderr == b""
    f32, _ = _render(tmp_path, "--precision", "f32", out="f32.ppm")
    assert (np.abs(f32.astype(int) - whole.astype(int)) <= 2).mean() >= 0.97
    r = _cli(SCENE, "out.png", "-w", str(W), "-s", "2", "-m", "4", "-q", cwd=tmp_path)
    assert r.returncode == 0 and open(os.path.join(tmp_path, "out.png"), "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    r = _cli(SCENE, "out.jpg", "-w", str(W), "-s", "2", "-m", "4", "-q", cwd=tmp_path)
    raw = open(os.path.join(tmp_path, "out.jpg"), "rb").read()
    assert r.returncode == 0 and raw[:2] == b"\xff\xd8" and raw[6:10] == b"JFIF" and raw[-2:] == b"\xff\xd9"
    r = _cli(SCENE, "out.gif", "-w", str(W), "-s", "1", "-m", "2", cwd=tmp_path)
    assert r.returncode == 1 and b"image format" in r.stderr


def test_cli_recovers_rows_from_a_part_file(tmp_path):
    import raytracer_amd as rt

    a, _ = _render(tmp_path, seed=1, out="a.ppm")
    b, _ = _render(tmp_path, seed=2, out="b.ppm")
    assert not np.array_equal(a, b)
    # a part file holding rows 40..47 and 10..19 of render A (j counts from the bottom of the image)
    kept = list(range(47, 39, -1)) + list(range(19, 9, -1))
    old = os.getcwd()
    os.chdir(tmp_path)
    try:
        pf = rt.PartFile("c.ppm", W, H)
        for j in kept:
            pf.write(j, a[H - 1 - j])
        pf.finish(remove=False)
    finally:
        os.chdir(old)
    assert pf.path == "c.ppm.0.PART"
    # truncate the last record: without --recover-corrupt the read fails, with it the row is dropped
    path = os.path.join(tmp_path, pf.path)
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:-20])
    r = _cli(SCENE, "c.ppm", "-w", str(W), "-s", str(SPP), "-m", str(MB), "--seed", "2", "-r", pf.path, cwd=tmp_path)
    assert r.returncode == 1 and b"failed to fill whole buffer" in r.stderr
    c, _ = _render(tmp_path, "-r", pf.path, "--recover-corrupt", seed=2, out="c.ppm")
    kept = kept[:-1]
    for j in range(H):
        want = a if j in kept else b
        assert np.array_equal(c[H - 1 - j], want[H - 1 - j]), f"row j={j}"
    # the recovered file is left alone, the new part file (c.ppm.1.PART) is gone
    assert sorted(os.listdir(tmp_path)) == ["a.ppm", "b.ppm", "c.ppm", "c.ppm.0.PART"]


def _render_py(tmp_path, *extra, seed=5, out="py.ppm"):
    r = subprocess.run([sys.executable, "-m", "raytracer_amd.render", SCENE, out, "-w", str(W), "-s", str(SPP), "-m", str(MB),
                        "--seed", str(seed), *extra], capture_output=True, cwd=tmp_path, timeout=110,
                       env={**os.environ, "PYTHONPATH": ROOT})
    assert r.returncode == 0, r.stderr.decode()
    return _read_ppm(os.path.join(tmp_path, out)), r


def test_node_render_driver_is_the_cli_on_one_gpu(tmp_path):
    """python -m raytracer_amd.render (the N-GPU driver: bands, sample split, reduce, .PART records) with one
    rank writes the picture bin/raytracer writes, recovers rows from a part file and leaves none behind."""
    import raytracer_amd as rt

    a, _ = _render(tmp_path, seed=1, out="a.ppm")
    b, r = _render_py(tmp_path, "--band-rows", "5", "--progress-bar-len", "40", "--stats", seed=1, out="b.ppm")
    assert np.array_equal(a, b)
    err = r.stderr.decode()
    assert "Rendering: [" in err and "Completed in 0:" in err and "rays on 1 GPU(s)" in err
    other, _ = _render(tmp_path, seed=2, out="other.ppm")
    kept = list(range(47, 41, -1)) + list(range(30, 20, -1)) + [0]
    old = os.getcwd()
    os.chdir(tmp_path)
    try:
        pf = rt.PartFile("c.ppm", W, H)
        for j in kept:
            pf.write(j, other[H - 1 - j])
        pf.finish(remove=False)
    finally:
        os.chdir(old)
    c, r = _render_py(tmp_path, "-r", pf.path, "--band-rows", "4", "-q", seed=1, out="c.ppm")
    assert b"Rendering" not in r.stderr and b"Completed" not in r.stderr  # (the ROCm runtime may print a line of its own)
    for j in range(H):
        want = other if j in kept else a
        assert np.array_equal(c[H - 1 - j], want[H - 1 - j]), f"row j={j}"
    assert sorted(os.listdir(tmp_path)) == ["a.ppm", "b.ppm", "c.ppm", "c.ppm.0.PART", "other.ppm"]
    bad = subprocess.run([sys.executable, "-m", "raytracer_amd.render", SCENE, "d.ppm", "-w", str(W), "-r", "missing.PART"],
                         capture_output=True, cwd=tmp_path, timeout=110, env={**os.environ, "PYTHONPATH": ROOT})
    assert bad.returncode == 1 and b"Error:" in bad.stderr and "d.ppm" not in os.listdir(tmp_path)
    # two concurrent half-batches per GPU: the same picture up to the order of two f64 additions per sum
    e, _ = _render_py(tmp_path, "--overlap", "2", "-q", "--no-part-file", seed=1, out="e.ppm")
    assert np.abs(e.astype(int) - a.astype(int)).max() <= 1 and (e != a).mean() <= 1e-3
