"""Host-side file formats of the command line (no GPU): .PART recovery files
(reference src/part_file.rs), output images (src/main.rs:161-162), the progress bar
(src/progress.rs) and the argument handling of bin/raytracer (struct Opt, src/main.rs:32-71).
The reference has no tests for these; the expectations below restate its format strings."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "raytracer")


@pytest.fixture
def in_tmp(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    return tmp_path


def _rows(h, w, seed=0):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


# ------------------------------------------------------------------ part files
def test_part_file_bytes_and_round_trip(in_tmp):
    h, w = 5, 4
    rows = _rows(h, w)
    pf = rt.PartFile("some/dir/out.png", w, h)
    assert pf.path == "out.png.0.PART"  # basename of the image, in the working directory
    for j in (4, 2, 0):
        pf.write(j, rows[j])
    pf.finish(remove=False)
    raw = open("out.png.0.PART", "rb").read()
    expect = b"4 5\n" + b"".join(b"L%d " % j + rows[j].tobytes() + b"\n" for j in (4, 2, 0))
    assert raw == expect
    got, present = rt.read_part_file("out.png.0.PART", h, w)
    assert present.tolist() == [True, False, True, False, True]
    assert np.array_equal(got[present], rows[present])


def test_part_file_picks_first_free_number_and_remove(in_tmp):
    a = rt.PartFile("img.ppm", 2, 2)
    b = rt.PartFile("img.ppm", 2, 2)
    assert (a.path, b.path) == ("img.ppm.0.PART", "img.ppm.1.PART")
    a.finish(remove=True)
    assert not os.path.exists("img.ppm.0.PART") and os.path.exists("img.ppm.1.PART")
    c = rt.PartFile("img.ppm", 2, 2)
    assert c.path == "img.ppm.0.PART"
    b.finish(remove=True)
    c.finish(remove=True)
    assert os.listdir(".") == []


def test_part_file_rows_may_contain_newlines_and_L(in_tmp):
    h, w = 3, 2
    rows = np.full((h, w, 3), ord("\n"), dtype=np.uint8)
    rows[1] = ord("L")
    pf = rt.PartFile("x.png", w, h)
    for j in range(h):
        pf.write(j, rows[j])
    pf.finish(False)
    got, present = rt.read_part_file(pf.path, h, w)
    assert present.all() and np.array_equal(got, rows)


def test_part_file_dimension_mismatch(in_tmp):
    pf = rt.PartFile("x.png", 4, 5)
    pf.finish(False)
    with pytest.raises(rt.RtError, match="Image dimensions expected to be 8x5, but recovery file says they're 4x5"):
        rt.read_part_file(pf.path, 5, 8)


@pytest.mark.parametrize(
    "tail, message",
    [
        (b"X1 " + b"\0" * 12 + b"\n", "Missing leading L"),
        (b"L1 " + b"\0" * 5, "failed to fill whole buffer"),
        (b"L1 " + b"\0" * 12 + b"X", "Missing newline on line 1"),
        (b"L1 " + b"\0" * 12, "failed to fill whole buffer"),
        (b"Lx " + b"\0" * 12 + b"\n", "invalid digit found in string"),
        (b"L7 " + b"\0" * 12 + b"\n", "out of range"),
    ],
)
def test_part_file_corruption(in_tmp, tail, message):
    h, w = 3, 4
    rows = _rows(h, w, 1)
    good = b"4 3\n" + b"L2 " + rows[2].tobytes() + b"\n"
    open("c.PART", "wb").write(good + tail)
    with pytest.raises(rt.RtError, match=message):
        rt.read_part_file("c.PART", h, w)
    got, present = rt.read_part_file("c.PART", h, w, recover_corrupt=True)
    # everything before the bad record survives; a record whose row was read completely but whose
    # newline is missing is kept, as the reference stores the row before checking the newline
    assert present[2] and np.array_equal(got[2], rows[2])
    assert present.sum() == (2 if message.startswith("Missing newline") or tail == b"L1 " + b"\0" * 12 else 1)


def test_part_file_header_errors(in_tmp):
    open("e.PART", "wb").write(b"")
    with pytest.raises(rt.RtError, match="Reading width"):
        rt.read_part_file("e.PART", 2, 2)
    open("e.PART", "wb").write(b"2 ")
    with pytest.raises(rt.RtError, match="Reading height"):
        rt.read_part_file("e.PART", 2, 2)
    with pytest.raises(rt.RtError, match="cannot open"):
        rt.read_part_file("missing.PART", 2, 2)


# ------------------------------------------------------------------ images
def _decode_png(raw):
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        (n,) = struct.unpack(">I", raw[pos:pos + 4])
        kind, data = raw[pos + 4:pos + 8], raw[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(kind + data)
        chunks.append((kind, data))
        pos += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, filt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, filt, lace) == (8, 2, 0, 0, 0)
    lines = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (lines[:, 0] == 0).all()
    return lines[:, 1:].reshape(h, w, 3)


def test_save_png_ppm_bmp(in_tmp):
    img = _rows(7, 5, 2)
    rt.save_image("a.png", img)
    assert np.array_equal(_decode_png(open("a.png", "rb").read()), img)
    rt.save_image("a.ppm", img)
    assert open("a.ppm", "rb").read() == b"P6\n5 7\n255\n" + img.tobytes()
    rt.save_image("a.BMP", img)
    raw = open("a.BMP", "rb").read()
    assert raw[:2] == b"BM" and struct.unpack("<I", raw[2:6])[0] == len(raw)
    w, h, planes, bpp = struct.unpack("<iiHH", raw[18:30])
    assert (w, h, planes, bpp) == (5, 7, 1, 24)
    stride = 16
    body = np.frombuffer(raw[54:], dtype=np.uint8).reshape(7, stride)[:, :15].reshape(7, 5, 3)
    assert np.array_equal(body[::-1, :, ::-1], img)


def test_save_tga_tiff_farbfeld_pam(in_tmp):
    from PIL import Image

    img = _rows(7, 5, 2)
    rt.save_image("a.tga", img)
    raw = open("a.tga", "rb").read()
    assert raw[2] == 2 and struct.unpack("<HHBB", raw[12:18]) == (5, 7, 24, 0x20) and len(raw) == 18 + 5 * 7 * 3
    assert np.array_equal(np.frombuffer(raw[18:], dtype=np.uint8).reshape(7, 5, 3)[:, :, ::-1], img)
    assert np.array_equal(np.asarray(Image.open("a.tga").convert("RGB")), img)
    for name in ("a.tiff", "b.TIF"):
        rt.save_image(name, img)
        assert np.array_equal(np.asarray(Image.open(name).convert("RGB")), img)
    odd = _rows(3, 3, 5)  # 27 bytes of pixels: the tables after them start on a word boundary
    rt.save_image("odd.tiff", odd)
    assert np.array_equal(np.asarray(Image.open("odd.tiff").convert("RGB")), odd)
    rt.save_image("a.ff", img)
    raw = open("a.ff", "rb").read()
    assert raw[:8] == b"farbfeld" and struct.unpack(">II", raw[8:16]) == (5, 7)
    px = np.frombuffer(raw[16:], dtype=">u2").reshape(7, 5, 4)
    assert np.array_equal(px[:, :, :3], img.astype(np.uint16) * 257) and (px[:, :, 3] == 65535).all()
    rt.save_image("a.pam", img)
    assert open("a.pam", "rb").read() == b"P7\nWIDTH 5\nHEIGHT 7\nDEPTH 3\nMAXVAL 255\nTUPLTYPE RGB\nENDHDR\n" + img.tobytes()
    with pytest.raises(Exception, match="not supported"):
        rt.save_image("a.gif", img)


def _test_picture(h, w):
    y, x = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(3)
    img = np.stack([128 + 100 * np.sin(x / 9.0), 128 + 100 * np.cos(y / 7.0), 40 + 2.0 * x + 1.0 * y], axis=2)
    img += rng.normal(0, 4, img.shape)
    img[h // 3: h // 3 + 5, :, :] = 255          # a hard edge and saturated values
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(48, 64), (45, 61), (8, 8), (1, 1), (3, 130)])
def test_save_jpeg_decodes_to_the_input(in_tmp, shape):
    # lossy, so parity with the image crate's encoder is unpinned; any baseline decoder must read the file
    # and give the picture back within the error of quality 75
    from PIL import Image

    img = _test_picture(*shape)
    rt.save_image("a.jpg", img)
    raw = open("a.jpg", "rb").read()
    assert raw[:4] == b"\xff\xd8\xff\xe0" and raw[6:10] == b"JFIF" and raw[-2:] == b"\xff\xd9"
    back = np.asarray(Image.open("a.jpg").convert("RGB"), dtype=np.float64)
    assert back.shape == img.shape
    mse = ((back - img) ** 2).mean()
    assert 10 * np.log10(255.0 ** 2 / max(mse, 1e-9)) > (30.0 if min(shape) >= 8 else 20.0)
    rt.save_image("b.JPEG", img)
    assert open("b.JPEG", "rb").read() == raw


def test_saved_jpeg_is_read_by_the_scene_loader(in_tmp):
    # the library's own JPEG decoder (image textures, src/textures.rs:32-36) reads what the encoder writes
    img = _test_picture(40, 56)
    rt.save_image("tex.jpg", img)
    text = """{
      camera: {lookFrom: [0,0,-5], lookAt: [0,0,0], directionUp: [0,1,0], fieldOfView: 40, aspectRatio: [1,1],
               aperture: 0, distanceToFocus: 5, startTime: 0, endTime: 1},
      background: {type: "plainColour", colour: [0,0,0]},
      textures: {t: {type: "imageTexture", filename: "tex.jpg"}},
      materials: {m: {type: "lambertian", texture: "t"}},
      objects: {s: {type: "sphere", centre: [0,0,0], radius: 1, material: "m"}},
      world: ["s"],
    }"""
    sc = rt.Scene.from_text(text, ".")
    assert sc.counts()["texels"] == img.size
    raw = (C.c_uint64 * 24).from_address(sc.desc_ptr)
    got = np.ctypeslib.as_array((C.c_uint8 * img.size).from_address(raw[22])).reshape(img.shape).astype(np.float64)
    mse = ((got - img) ** 2).mean()
    assert 10 * np.log10(255.0 ** 2 / mse) > 30.0


def test_save_unknown_extension(in_tmp):
    with pytest.raises(rt.RtError, match="image format"):
        rt.save_image("a.xyz", _rows(2, 2))
    assert not os.path.exists("a.xyz")


# ------------------------------------------------------------------ progress bar
def _bar_output(length, glyphs, total, updates, clear=False, label="Rendering"):
    r, w = os.pipe()
    bar = rt.ProgressBar(length, label, glyphs, 100, total, fd=w)
    for _ in range(updates):
        bar.update()
    if clear:
        bar.clear()
    bar.close()
    os.close(w)
    with os.fdopen(r, "rb") as f:
        return f.read().decode("utf-8")


def test_progress_bar_frames():
    # length 34 - len("Rendering") - 15 = 10 cells; 3 glyphs -> 2 sub-steps per cell
    frames = _bar_output(34, " .:", 20, 20).split("\r")[1:]
    assert len(frames) == 20
    assert frames[0] == "Rendering: [.         ]  0:00:00 "  # progress 1 of 20 sub-steps: the half glyph
    assert frames[1] == "Rendering: [:         ]  0:00:00 "
    assert frames[9] == "Rendering: [:::::     ]  0:00:00 "
    assert frames[19] == "Rendering: [::::::::::]  0:00:00 "
    # 40 updates over 20 sub-steps: integer division, as in the reference
    frames = _bar_output(34, " .:", 40, 4).split("\r")[1:]
    assert frames[0] == "Rendering: [          ]  0:00:00 "
    assert frames[1] == "Rendering: [.         ]  0:00:00 "
    assert frames[2] == "Rendering: [.         ]  0:00:00 "
    assert frames[3] == "Rendering: [:         ]  0:00:00 "


def test_progress_bar_unicode_glyphs_and_clear():
    out = _bar_output(34, " -=≡", 10, 1, clear=True)
    frames = out.split("\r")
    assert frames[1] == "Rendering: [≡" + " " * 9 + "]  0:00:00 "
    assert frames[2] == " " * (34 + len("Rendering") + 12) and frames[3] == ""
    frames = _bar_output(34, " -=≡", 30, 2).split("\r")[1:]
    assert frames[0].startswith("Rendering: [-") and frames[1].startswith("Rendering: [=")


# ------------------------------------------------------------------ command line (no GPU needed for these)
def _cli(*args, cwd=None):
    return subprocess.run([CLI, *args], capture_output=True, text=True, cwd=cwd, timeout=60)


def test_cli_help_version_and_usage_errors(in_tmp):
    assert os.path.exists(CLI), "bin/raytracer is not built (make)"
    r = _cli("--help")
    assert r.returncode == 0
    for flag in ("--width", "--ray-samples", "--max-bounces", "--quiet", "--ascii-symbols-only", "--recover-from",
                 "--recover-corrupt", "--progress-bar-len", "--progress-bar-samples", "--no-part-file"):
        assert flag in r.stdout
    assert _cli("-V").stdout.startswith("raytracer ")
    r = _cli()
    assert r.returncode == 1 and "<input-file>" in r.stderr and "<output-file>" in r.stderr
    r = _cli("a.json5")
    assert r.returncode == 1 and "<output-file>" in r.stderr and "<input-file>" not in r.stderr.split("USAGE")[0]
    r = _cli("a.json5", "b.png", "--frobnicate")
    assert r.returncode == 1 and "--frobnicate" in r.stderr
    r = _cli("a.json5", "b.png", "-w", "12x")
    assert r.returncode == 1 and "invalid digit" in r.stderr
    r = _cli("a.json5", "b.png", "-s")
    assert r.returncode == 1 and "requires a value" in r.stderr
    assert os.listdir(".") == []


def test_cli_scene_and_part_file_errors_leave_no_part_file(in_tmp):
    r = _cli("missing.json5", "out.png")
    assert r.returncode == 1 and r.stderr.startswith("Error: ")
    scene = os.path.join(ROOT, "examples", "cornell.json5")
    open("bad.PART", "wb").write(b"10 10\n")
    r = _cli(scene, "out.png", "-w", "32", "-r", "bad.PART")
    assert r.returncode == 1 and "Image dimensions expected to be 32x32" in r.stderr
    assert sorted(os.listdir(".")) == ["bad.PART"]


def _texture_scene(name):
    return """{
      camera: {lookFrom: [0,0,-5], lookAt: [0,0,0], directionUp: [0,1,0], fieldOfView: 40, aspectRatio: [1,1],
               aperture: 0, distanceToFocus: 5, startTime: 0, endTime: 1},
      background: {type: "plainColour", colour: [0,0,0]},
      textures: {t: {type: "imageTexture", filename: "%s"}},
      materials: {m: {type: "lambertian", texture: "t"}},
      objects: {s: {type: "sphere", centre: [0,0,0], radius: 1, material: "m"}},
      world: ["s"],
    }""" % name


def test_malformed_jpeg_textures_are_errors_not_faults(in_tmp):
    # A texture file is untrusted input: truncated or garbage segments must come back as RT_ERR_SCENE (or decode to
    # something), never read out of bounds.  Run under the library as built; the checks are in image_io.cpp.
    img = _test_picture(24, 40)
    rt.save_image("good.jpg", img)
    raw = bytearray(open("good.jpg", "rb").read())

    def load(data):
        open("bad.jpg", "wb").write(bytes(data))
        return rt.Scene.from_text(_texture_scene("bad.jpg"), ".")

    def segments(data):
        i, out = 2, {}
        while i + 4 <= len(data) and data[i] == 0xFF:
            m, ln = data[i + 1], (data[i + 2] << 8) | data[i + 3]
            out.setdefault(m, i)
            if m == 0xDA:
                break
            i += 2 + ln
        return out

    seg = segments(raw)
    assert {0xDB, 0xC0, 0xC4, 0xDA} <= set(seg)
    # every truncation of the headers
    for cut in range(2, seg[0xDA] + 14):  # (beyond the scan header the missing entropy data reads as zeros)
        with pytest.raises(rt.RtError):
            load(raw[:cut])
    # SOS names Huffman tables 15 / 15 (used to index hdc[4] / hac[4] out of bounds)
    bad = bytearray(raw)
    bad[seg[0xDA] + 6] = 0xFF
    with pytest.raises(rt.RtError, match="Huffman table id"):
        load(bad)
    # DQT whose length says one byte of table
    bad = bytearray(raw)
    q = seg[0xDB]
    bad[q + 2:q + 4] = b"\x00\x04"
    with pytest.raises(rt.RtError):
        load(bad[:q + 6] + raw[q + 2 + ((raw[q + 2] << 8) | raw[q + 3]):])
    # SOF / SOS / DRI shorter than their fields
    for m, ln in ((0xC0, 4), (0xC0, 9), (0xDA, 2), (0xDA, 4)):
        bad = bytearray(raw)
        at = seg[m]
        old = (raw[at + 2] << 8) | raw[at + 3]
        with pytest.raises(rt.RtError):
            load(bad[:at + 2] + bytes([0, ln]) + raw[at + 4:at + 2 + ln] + raw[at + 2 + old:])
    with pytest.raises(rt.RtError):
        load(raw[:seg[0xDA]] + b"\xff\xdd\x00\x02" + raw[seg[0xDA]:])
    # random corruption of single header bytes: an error or a picture, never a crash
    rng = np.random.default_rng(11)
    for _ in range(300):
        bad = bytearray(raw)
        bad[int(rng.integers(2, seg[0xDA] + 14))] = int(rng.integers(0, 256))
        try:
            load(bad)
        except rt.RtError:
            pass


def test_partition_rule_of_the_library_matches_the_python_one():
    # rt_split_samples (several GPUs behind the C ABI) and parallel.split_samples (one process per GPU) must agree
    from raytracer_amd.parallel import split_samples

    for total in [0, 1, 7, 8, 13, 100, 1000]:
        for parts in [1, 2, 3, 4, 8]:
            covered = []
            for part in range(parts):
                got = rt.split_samples_c(total, parts, part, first=5)
                assert got == split_samples(total, parts, part, first_sample=5)
                covered += list(range(got[0], got[0] + got[1]))
            assert covered == list(range(5, 5 + total))
    assert rt.split_samples_c(10, 0, 0) == (0, 0) and rt.split_samples_c(10, 2, 2) == (0, 0)


def test_multi_create_without_a_gpu_is_an_error_code():
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    with pytest.raises(rt.RtError) as e:
        rt.MultiRenderer([0, 1])
    assert e.value.code == -2
    r = subprocess.run([CLI, "--help"], capture_output=True)
    assert b"--gpus <n>" in r.stdout and b"--devices <a,b,..>" in r.stdout


# ---- one ROCm stack per process, whatever the loading order.  PyTorch wheels bundle
# their own ROCm libraries and ask for them by unversioned file names that match no SONAME of /opt/rocm's copies, so a process
# that loads /opt/rocm's RCCL and then imports torch holds everything twice and aborts at exit - reproduced here without a GPU
# (`forbidden`), and what the library and its bindings do so that it cannot happen through them.
_LOADER_SCRIPT = r"""
import ctypes, os, sys
order = sys.argv[1]
lib_path = sys.argv[2]
def mapped(key):
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if key in l and '.so' in l})
if order == 'ffi_host_rccl_first':          # a host that links the system RCCL before it loads the library; no torch
    ctypes.CDLL('/opt/rocm/lib/librccl.so.1', mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(lib_path)
elif order == 'bindings_then_torch':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(lib_path))))
    import raytracer_amd
    lib = raytracer_amd.load_library()
elif order == 'torch_then_bindings':
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(lib_path))))
    import raytracer_amd
    lib = raytracer_amd.load_library()
elif order == 'forbidden':                  # the system stack, then torch, behind the library's back
    ctypes.CDLL('/opt/rocm/lib/librccl.so.1', mode=ctypes.RTLD_GLOBAL)
    import torch
    print('LIBS', len(mapped('librccl')), len(mapped('libamdhip64')))
    sys.exit(0)
lib.rt_multi_rccl_library.restype = ctypes.c_char_p
path = lib.rt_multi_rccl_library()
print('RCCL', path.decode() if path else None)
if order == 'bindings_then_torch':
    import torch
print('LIBS', len(mapped('librccl')), len(mapped('libamdhip64')), len(mapped('libhsa-runtime64')))
"""


@pytest.mark.parametrize("order", ["ffi_host_rccl_first", "bindings_then_torch", "torch_then_bindings"])
def test_one_rocm_stack_per_process_whatever_the_loading_order(order, tmp_path):
    import subprocess
    import sys

    import raytracer_amd

    if not os.path.exists("/opt/rocm/lib/librccl.so.1"):
        pytest.skip("no system RCCL")
    script = tmp_path / "order.py"
    script.write_text(_LOADER_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("RT_RCCL_LIB", "RT_REDUCE")}
    p = subprocess.run([sys.executable, str(script), order, raytracer_amd.LIB_PATH], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.returncode, p.stderr[-500:])            # in particular: no abort at exit
    assert "free()" not in p.stderr and "double free" not in p.stderr
    out = dict(line.split(" ", 1) for line in p.stdout.splitlines() if " " in line)
    assert out["LIBS"].split() == ["1", "1", "1"], p.stdout               # one RCCL, one HIP runtime, one HSA runtime
    if order == "ffi_host_rccl_first":
        assert out["RCCL"].startswith("/opt/rocm")                        # the copy the host holds already
    else:
        assert "/torch/lib/librccl.so" in out["RCCL"]                      # the one next to the HIP runtime the process runs on


def test_the_order_the_library_avoids_does_abort(tmp_path):
    # the cause, kept visible: /opt/rocm's RCCL first, then `import torch` -> two of everything and an abort at exit
    import subprocess
    import sys

    import raytracer_amd

    if not os.path.exists("/opt/rocm/lib/librccl.so.1"):
        pytest.skip("no system RCCL")
    script = tmp_path / "order.py"
    script.write_text(_LOADER_SCRIPT)
    p = subprocess.run([sys.executable, str(script), "forbidden", raytracer_amd.LIB_PATH], capture_output=True, text=True, timeout=300)
    if p.returncode == 0 and "LIBS 1 1" in p.stdout:
        pytest.skip("this torch build shares the system ROCm libraries")
    assert "LIBS 2 2" in p.stdout    # two RCCLs, two HIP runtimes in one process
