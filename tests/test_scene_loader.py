"""CPU tests of the boundary: the C-ABI library loads and exports every declared symbol, and
the host side (JSON5 scene files -> flattened arrays) behaves like config::load_config
(reference src/config.rs).  No compute entry point is called here (no GPU)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES, scene_path


def test_library_exports_every_declared_symbol(rt):
    header = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    lib = C.CDLL(rt.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/rt_mi355x.h but not exported"
    assert declared == {n for n, _, _ in rt.EXPORTS}


def test_struct_layouts(rt, orc):
    assert rt.RAY_DTYPE.itemsize == 72      # rt_ray: 9 doubles
    assert rt.HIT_DTYPE.itemsize == 112     # rt_hit: 6 int32 + 9 doubles + second_t, inst_id, pad
    assert C.sizeof(rt.RtStats) == 96      # 56 + ms_tail, rays_tail, tail_launches, extend_kernel, paths_tail, paths_first
    assert orc.HIT_DTYPE.itemsize == 96 + 16


def test_compute_without_gpu_fails_loudly(rt):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.Renderer(device=0)
    assert e.value.code == -2  # RT_ERR_DEVICE: there is no CPU fallback


EXPECTED_COUNTS = {
    # name: (rects, spheres, planes, media, triangles, objects, top_list)
    "two_spheres": dict(rects=0, spheres=4, planes=2, media=0, triangles=0, objects=6, top_list=3),
    "cornell": dict(rects=18, spheres=0, planes=0, media=0, triangles=0, objects=8, top_list=1, xforms=2),
    "globe": dict(rects=5, spheres=1, planes=0, media=0, images=1, texels=1024 * 512 * 3, top_list=1),
    "blocks_ball": dict(rects=400 * 6 + 6 + 5, spheres=1, media=1, objects=404, top_list=1),
    "knight": dict(rects=5, planes=1, triangles=100352, top_list=2),
    "blue_ball": dict(rects=6 + 6, spheres=3, media=1, objects=12, top_list=1),
}


@pytest.mark.parametrize("name", SCENES)
def test_example_scenes_flatten(rt, name):
    sc = rt.Scene.load(scene_path(name), ROOT)
    counts = sc.counts()
    for k, v in EXPECTED_COUNTS[name].items():
        assert counts[k] == v, (k, counts)
    assert 0 < counts["stack_depth"] <= 64


def test_image_height_rounding(rt):
    # src/main.rs:82: (width / aspect).round(); blocks_ball is 16:9 -> 600 / (16/9) = 337.5 -> 338
    assert rt.Scene.load(scene_path("blocks_ball"), ROOT).image_height(600) == 338
    assert rt.Scene.load(scene_path("two_spheres"), ROOT).image_height(600) == 375
    assert rt.Scene.load(scene_path("cornell"), ROOT).image_height(3840) == 3840


BASE = """{
  camera: {lookFrom: [0,0,-5], lookAt: [0,0,0], directionUp: [0,1,0], fieldOfView: 40, aspectRatio: [1,1],
           aperture: 0, distanceToFocus: 5, startTime: 0, endTime: 1},
  background: {type: "plainColour", colour: [0,0,0]},
  textures: {t: {type: "solidColour", colour: [1,1,1]} TEXS},
  materials: {m: {type: "lambertian", texture: "t"} MATS},
  objects: {OBJS},
  world: [WORLD],
}"""


def scene_text(objs='s: {type: "sphere", centre: [0,0,0], radius: 1, material: "m"}', world='"s"', mats="", texs=""):
    return BASE.replace("OBJS", objs).replace("WORLD", world).replace("MATS", mats).replace("TEXS", texs)


def test_json5_features(rt):
    text = scene_text() + "\n// trailing comment\n/* block */"
    text = text.replace('"sphere"', "'sphere'").replace("radius: 1", "radius: +.5e1").replace("[0,0,0], radius", "[0x0, -0., 0,], radius")
    sc = rt.Scene.from_text(text)
    assert sc.counts()["spheres"] == 1


@pytest.mark.parametrize(
    "text,message",
    [
        (scene_text().replace("aperture: 0,", "aperture: 0, bogus: 1,"), "unknown field `bogus`"),          # deny_unknown_fields
        (scene_text().replace("aperture: 0,", ""), "missing field `aperture`"),
        (scene_text().replace('material: "m"', 'material: "nope"'), "Material nope does not exist"),          # config.rs:197
        (scene_text(world='"ghost"'), "Object ghost does not exist"),                                          # config.rs:33
        (scene_text(objs='a: {type: "translate", prototype: "b", offset: [0,0,0]}, b: {type: "rotateY", prototype: "a", degrees: 1}', world='"a"'),
         "is impossible to construct"),                                                                         # config.rs:397-400
        (scene_text(objs='r: {type: "rect", corner0: [0,0,0], corner1: [1,1,1], facingForward: true, material: "m"}', world='"r"'),
         "Rectangles are 2d"),                                                                                  # config.rs:278
        (scene_text(mats=', c: {type: "checkered", odd: "c", even: "m", tileDensity: 2}'), "Material c is impossible to construct"),
        (scene_text().replace("endTime: 1", "endTime: 2"), "0 <= start_time <= end_time <= 1"),                # camera.rs:36-38
        (scene_text().replace('type: "sphere"', 'type: "blob"'), "unknown variant `blob`"),
        (scene_text(world=""), "empty slice"),                                                                 # hitting.rs:164
        (scene_text().replace("world:", "world: 3, //"), "`world` must be an array"),
        (scene_text().replace("world:", "world: [,"), "parse error"),
        (scene_text(objs='m1: {type: "mesh", filename: "/nonexistent.obj", material: "m"}', world='"m1"'), "cannot open"),
    ],
)
def test_scene_errors(rt, text, message):
    with pytest.raises(rt.RtError) as e:
        rt.Scene.from_text(text)
    assert message in e.value.message, e.value.message
    assert e.value.code == -4


def test_forward_references_and_instancing(rt):
    # objects may name prototypes that appear later (fixpoint, config.rs:186-401); one prototype may be
    # instanced by several transforms, be a medium boundary and sit in `world` itself.
    objs = ('t2: {type: "translate", prototype: "r", offset: [3,0,0]}, r: {type: "rotateZ", prototype: "b", degrees: 30},'
            't1: {type: "translate", prototype: "b", offset: [-3,0,0]}, b: {type: "block", corner0: [0,0,0], corner1: [1,1,1], material: "m"},'
            'fog: {type: "constantMedium", boundary: "t1", phaseFunction: "m", density: 0.1}')
    sc = rt.Scene.from_text(scene_text(objs=objs, world='"t1", "t2", "b", "fog"'))
    c = sc.counts()
    assert c["rects"] == 6          # one Block, shared by every instance
    assert c["objects"] == 5        # t1, t2, b, fog + fog's boundary instance
    assert c["xforms"] == 3 and c["media"] == 1


def test_medium_bounded_by_a_medium_loads_and_deeper_nestings_are_refused(rt):
    # ConstantMedium::new takes any Hittable as its boundary (src/objects.rs:388-391), another medium included: two
    # levels flatten (the kernels answer the nested boundary query); three are refused with a message
    objs = ('b: {type: "block", corner0: [0,0,0], corner1: [1,1,1], material: "m"},'
            'f1: {type: "constantMedium", boundary: "b", phaseFunction: "m", density: 0.1},'
            'f2: {type: "constantMedium", boundary: "f1", phaseFunction: "m", density: 0.1}')
    sc = rt.Scene.from_text(scene_text(objs=objs, world='"f2"'))
    c = sc.counts()
    assert c["media"] == 2 and c["objects"] == 3 and c["rects"] == 6     # f2, its boundary f1, f1's boundary b
    objs += ', f3: {type: "constantMedium", boundary: "f2", phaseFunction: "m", density: 0.1}'
    with pytest.raises(rt.RtError) as e:
        rt.Scene.from_text(scene_text(objs=objs, world='"f3"'))
    assert "not supported" in e.value.message and "two deep" in e.value.message


def test_obj_loader(rt, tmp_path):
    obj = tmp_path / "m.obj"
    obj.write_text(
        "# two objects; polygons are fan-triangulated; lines / points ignored\n"
        "o first\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\nl 1 2\n"
        "o second\nv 0 0 1\nv 1 0 1\nv 0 1 1\nv 0 0 2\nf 5 6 7\nf -4 -3 -1\nf 5/1 7/1 8/1\n"
    )
    mesh = 'k: {type: "mesh", filename: "%s", objectName: "%s", material: "m"}'
    assert rt.Scene.from_text(scene_text(objs=mesh % (obj, "first"), world='"k"')).counts()["triangles"] == 2
    assert rt.Scene.from_text(scene_text(objs=mesh % (obj, "second"), world='"k"')).counts()["triangles"] == 3
    with pytest.raises(rt.RtError) as e:   # default objectName "" (config.rs:323) is not in this file
        rt.Scene.from_text(scene_text(objs='k: {type: "mesh", filename: "%s", material: "m"}' % obj, world='"k"'))
    assert "No object with name" in e.value.message and '"first", "second"' in e.value.message
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")   # no `o` line: one object named ""
    assert rt.Scene.from_text(scene_text(objs='k: {type: "mesh", filename: "%s", material: "m"}' % obj, world='"k"')).counts()["triangles"] == 1


def test_jpeg_decoder_against_pil(rt):
    from PIL import Image

    want = np.asarray(Image.open(os.path.join(ROOT, "examples", "globe.jpg")).convert("RGB"), dtype=np.int32)
    sc = rt.Scene.load(scene_path("globe"), ROOT)
    c = sc.counts()
    assert c["texels"] == want.size
    # texels pointer: rt_scene_desc field 11 (pointer, count pairs)
    raw = (C.c_uint64 * 24).from_address(sc.desc_ptr)
    got = np.ctypeslib.as_array((C.c_uint8 * want.size).from_address(raw[22])).reshape(want.shape).astype(np.int32)
    diff = np.abs(got - want)
    # another decoder may differ by an LSB or two per channel (IDCT / colour rounding): parity unpinned
    assert diff.max() <= 3 and diff.mean() < 0.5


def _texels(rt, path, shape):
    """The RGB8 texels the loader decodes `path` to (through an imageTexture of a scene), as an array of `shape`."""
    text = scene_text(objs='s: {type: "sphere", centre: [0,0,0], radius: 1, material: "p"}', world='"s"',
                      texs=', i: {type: "imageTexture", filename: "%s"}' % path, mats=', p: {type: "lambertian", texture: "i"}')
    sc = rt.Scene.from_text(text)
    n = int(np.prod(shape))
    assert sc.counts()["texels"] == n
    raw = (C.c_uint64 * 24).from_address(sc.desc_ptr)
    return np.ctypeslib.as_array((C.c_uint8 * n).from_address(raw[22])).reshape(shape).copy()


def _picture(h, w, seed=3):
    y, x = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(seed)
    img = np.stack([128 + 100 * np.sin(x / 7.0 + y / 13.0), 128 + 90 * np.cos(x / 5.0) * np.sin(y / 9.0), (x * 3 + y * 5) % 256], axis=2)
    img += rng.normal(scale=6.0, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_png_textures_decode_exactly(rt, tmp_path):
    # image 0.23 decodes whatever the file is (src/textures.rs:32-36) and into_rgb8() drops alpha, replicates grey and
    # narrows 16-bit samples; PNG is lossless, so the texels must be PIL's (16-bit: (x + 128) / 257, stated in image_io.cpp)
    from PIL import Image

    img = _picture(37, 53)
    cases = {
        "rgb.png": Image.fromarray(img),
        "rgba.png": Image.fromarray(np.dstack([img, 255 - img[:, :, :1]])),
        "grey.png": Image.fromarray(img[:, :, 0]),
        "grey_alpha.png": Image.fromarray(np.dstack([img[:, :, 0], img[:, :, 1]]), mode="LA"),
        "palette.png": Image.fromarray(img).quantize(200),
        "palette16.png": Image.fromarray(img).quantize(16),
        "palette4.png": Image.fromarray(img).quantize(4),
        "bilevel.png": Image.fromarray(img[:, :, 0]).convert("1"),
    }
    for name, im in cases.items():
        path = str(tmp_path / name)
        im.save(path, optimize=name.startswith("pal"))
        want = np.asarray(Image.open(path).convert("RGB"))
        assert np.array_equal(_texels(rt, path, want.shape), want), name
    # interlaced (Adam7), every filter type, several IDAT chunks: written by hand below since PIL writes neither
    import struct
    import zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    h, w = img.shape[:2]
    # non-interlaced, filter types 0-4 cycling over the rows
    rows, prev = [], np.zeros(w * 3, dtype=np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        f = y % 5
        a = np.concatenate([np.zeros(3, dtype=np.int32), cur[:-3]])
        c = np.concatenate([np.zeros(3, dtype=np.int32), prev[:-3]])
        if f == 0:
            pred = 0
        elif f == 1:
            pred = a
        elif f == 2:
            pred = prev
        elif f == 3:
            pred = (a + prev) // 2
        else:
            pa, pb, pc = np.abs(prev - c), np.abs(a - c), np.abs(a + prev - 2 * c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        rows.append(bytes([f]) + ((cur - pred) % 256).astype(np.uint8).tobytes())
        prev = cur
    z = zlib.compress(b"".join(rows), 6)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"gAMA", struct.pack(">I", 45455)) + \
        chunk(b"IDAT", z[:100]) + chunk(b"IDAT", z[100:]) + chunk(b"IEND", b"")
    path = str(tmp_path / "filters.png")
    open(path, "wb").write(png)
    assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), img)      # the hand-written file is a valid PNG
    assert np.array_equal(_texels(rt, path, img.shape), img)
    # Adam7: the seven passes, filter 0
    x0, y0, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    data = b""
    for p in range(7):
        sub = img[y0[p]::dy[p], x0[p]::dx[p]]
        if sub.size:
            data += b"".join(b"\x00" + sub[r].tobytes() for r in range(sub.shape[0]))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(data)) + chunk(b"IEND", b"")
    path = str(tmp_path / "adam7.png")
    open(path, "wb").write(png)
    assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), img)
    assert np.array_equal(_texels(rt, path, img.shape), img)
    # 16-bit RGB: narrowed by (x + 128) / 257
    wide = (img.astype(np.uint32) * 257 + np.arange(w, dtype=np.uint32)[None, :, None] % 200).astype(">u2")
    data = b"".join(b"\x00" + wide[r].tobytes() for r in range(h))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(data)) + chunk(b"IEND", b"")
    path = str(tmp_path / "rgb16.png")
    open(path, "wb").write(png)
    assert np.array_equal(_texels(rt, path, img.shape), ((wide.astype(np.uint32) + 128) // 257).astype(np.uint8))


def test_malformed_png_textures_are_errors_not_faults(rt, tmp_path):
    from PIL import Image

    good = tmp_path / "good.png"
    Image.fromarray(_picture(19, 23)).save(str(good))
    data = good.read_bytes()
    text = scene_text(objs='s: {type: "sphere", centre: [0,0,0], radius: 1, material: "p"}', world='"s"',
                      texs=', i: {type: "imageTexture", filename: "%s"}', mats=', p: {type: "lambertian", texture: "i"}')
    bad = tmp_path / "bad.png"
    rng = np.random.default_rng(11)
    cases = [data[:k] for k in (0, 4, 8, 20, 33, 40, len(data) // 2, len(data) - 5)]
    for _ in range(200):   # random byte flips: a CRC, a length, the zlib stream, the header ...
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(8, len(b)))] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    errors = 0
    for c in cases:
        bad.write_bytes(c)
        try:
            rt.Scene.from_text(text % bad)
        except rt.RtError as e:
            assert e.code == -4
            errors += 1
    assert errors >= len(cases) - 5   # (a flip inside an ancillary chunk's body with its CRC is still an error; only a no-op flip decodes)


def test_progressive_and_subsampled_jpeg_textures(rt, tmp_path):
    # progressive (SOF2) files, 4:2:2 (h2v1) and 4:2:0 (h2v2) chroma, restart intervals, grey: against PIL / libjpeg within the
    # tolerance test_jpeg_decoder_against_pil states (another IDCT and rounding; the up-sampling filter is jpeg-decoder's triangle,
    # libjpeg's "fancy" one differs in the last bit) - unpinned by the reference, which holds no decoded texel
    from PIL import Image

    img = _picture(75, 101)
    smooth = (np.stack(np.meshgrid(np.linspace(20, 230, 101), np.linspace(40, 200, 75)), axis=2).sum(axis=2) / 2)[:, :, None].repeat(3, axis=2).astype(np.uint8)
    smooth[:, :, 1] = 255 - smooth[:, :, 1]
    for pic_name, pic in (("noisy", img), ("smooth", smooth)):
        for name, kw in {"prog444": dict(progressive=True, subsampling=0), "prog422": dict(progressive=True, subsampling=1),
                         "prog420": dict(progressive=True, subsampling=2), "base422": dict(subsampling=1), "base420": dict(subsampling=2),
                         "base444_opt": dict(subsampling=0, optimize=True)}.items():
            path = str(tmp_path / f"{pic_name}_{name}.jpg")
            Image.fromarray(pic).save(path, quality=90, **kw)
            want = np.asarray(Image.open(path).convert("RGB"), dtype=np.int32)
            got = _texels(rt, path, want.shape).astype(np.int32)
            diff = np.abs(got - want)
            limit = 3 if "444" in name else 24   # subsampled: the two decoders' chroma filters differ at sharp colour edges
            assert diff.max() <= limit and diff.mean() < (0.5 if "444" in name else 1.5), (pic_name, name, diff.max(), diff.mean())
    path = str(tmp_path / "grey_prog.jpg")
    Image.fromarray(img[:, :, 0]).save(path, quality=85, progressive=True)
    want = np.asarray(Image.open(path).convert("RGB"), dtype=np.int32)
    assert np.abs(_texels(rt, path, want.shape).astype(np.int32) - want).max() <= 2


def test_h2v1_upsampling_is_the_triangle_filter(rt, tmp_path):
    # the filter itself, on a file whose chroma plane is known: two flat chroma halves; jpeg-decoder's h2v1 up-sampler gives
    # (3 near + far + 2) >> 2 at the seam and the plane's own values elsewhere
    from PIL import Image

    pic = np.zeros((16, 32, 3), dtype=np.uint8)
    pic[:, :16] = (200, 60, 60)
    pic[:, 16:] = (60, 60, 200)
    path = str(tmp_path / "seam.jpg")
    Image.fromarray(pic).save(path, quality=100, subsampling=1)
    got = _texels(rt, path, pic.shape).astype(np.int32)
    # away from the seam the halves are flat; next to it the chroma is a 3:1 blend: monotone from one half to the other
    assert np.abs(got[:, :12] - pic[:, :12].astype(np.int32)).max() <= 4 and np.abs(got[:, 20:] - pic[:, 20:].astype(np.int32)).max() <= 4
    red = got[8, 12:20, 0]
    assert np.all(np.diff(red) <= 2) and red[0] > red[-1] + 100


def test_bmp_and_pnm_textures(rt, tmp_path):
    from PIL import Image

    img = _picture(21, 34)
    for name, im in {"a.bmp": Image.fromarray(img), "pal.bmp": Image.fromarray(img).quantize(64), "rgba.bmp": Image.fromarray(np.dstack([img, img[:, :, :1]]))}.items():
        path = str(tmp_path / name)
        im.save(path)
        want = np.asarray(Image.open(path).convert("RGB"))
        assert np.array_equal(_texels(rt, path, want.shape), want), name
    (tmp_path / "g.pgm").write_bytes(b"P5\n# grey\n34 21\n255\n" + img[:, :, 0].tobytes())
    assert np.array_equal(_texels(rt, str(tmp_path / "g.pgm"), img.shape), img[:, :, :1].repeat(3, axis=2))
    (tmp_path / "t.ppm").write_text("P3\n2 2\n15\n15 0 0  0 15 0\n0 0 15  15 15 15\n")
    assert _texels(rt, str(tmp_path / "t.ppm"), (2, 2, 3)).tolist() == [[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [255, 255, 255]]]
    (tmp_path / "x.gif").write_bytes(b"GIF89a" + bytes(40))
    with pytest.raises(rt.RtError) as e:
        _texels(rt, str(tmp_path / "x.gif"), (1,))
    assert "unsupported image format" in e.value.message


def test_split_samples():
    from raytracer_amd.parallel import split_rows, split_samples

    for total in [0, 1, 7, 100, 1000]:
        for world in [1, 2, 3, 8]:
            ranges = [split_samples(total, world, r, first_sample=5) for r in range(world)]
            assert ranges[0][0] == 5 and sum(n for _, n in ranges) == total
            for (a, n), (b, _) in zip(ranges, ranges[1:]):
                assert a + n == b
            assert max(n for _, n in ranges) - min(n for _, n in ranges) <= 1
    assert [split_rows(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]


# ---- independent check of the loader / flattener: VALUES derived by hand from examples/*.json5 and the reference's
# construction rules (src/config.rs:247-279 rect axis rule, src/objects.rs:149-168 Block sides, :642-700 Spotlight,
# src/transforms.rs rotation conventions, src/camera.rs:25-66).  The oracle shares the product's scene front end
# (oracle/Makefile), so GPU-vs-oracle tests cannot see a wrong axis rule or index; these numbers can.
class _Rect(C.Structure):
    _fields_ = [("a0", C.c_double), ("a1", C.c_double), ("b0", C.c_double), ("b1", C.c_double), ("k", C.c_double),
                ("axis", C.c_int32), ("facing_positive", C.c_int32), ("material", C.c_int32), ("owner", C.c_int32),
                ("sub", C.c_int32), ("pad", C.c_int32)]


class _Xform(C.Structure):
    _fields_ = [("m", C.c_double * 9), ("b", C.c_double * 3)]


class _Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("image", C.c_int32), ("colour", C.c_double * 3), ("param", C.c_double),
                ("odd", C.c_int32), ("even", C.c_int32)]


class _Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("xform", C.c_int32), ("first", C.c_int32), ("count", C.c_int32),
                ("world_index", C.c_int32), ("pad", C.c_int32 * 3)]


class _Sphere(C.Structure):
    _fields_ = [("centre", C.c_double * 3), ("radius", C.c_double), ("material", C.c_int32), ("owner", C.c_int32)]


class _Triangle(C.Structure):
    _fields_ = [("point", C.c_double * 3), ("vec1", C.c_double * 3), ("vec2", C.c_double * 3), ("normal", C.c_double * 3),
                ("material", C.c_int32), ("owner", C.c_int32), ("sub", C.c_int32), ("pad", C.c_int32)]


class _Plane(C.Structure):
    _fields_ = [("point", C.c_double * 3), ("vec1", C.c_double * 3), ("vec2", C.c_double * 3), ("normal", C.c_double * 3),
                ("uv_repeat", C.c_double), ("material", C.c_int32), ("owner", C.c_int32)]


class _Medium(C.Structure):
    _fields_ = [("neg_inv_density", C.c_double), ("boundary_object", C.c_int32), ("material", C.c_int32), ("owner", C.c_int32),
                ("rng_key", C.c_uint32)]


class _Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("texel_offset", C.c_uint64)]


class _Camera(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v")] + \
               [("lens_radius", C.c_double), ("start_time", C.c_double), ("end_time", C.c_double)]


class _Sky(C.Structure):
    _fields_ = [("gradient", C.c_int32), ("pad", C.c_int32), ("colour", C.c_double * 3), ("unit_dir", C.c_double * 3),
                ("colour0", C.c_double * 3), ("colour1", C.c_double * 3)]


_ARRAYS = [("nodes", None), ("objects", _Object), ("xforms", _Xform), ("spheres", _Sphere), ("moving_spheres", None),
           ("rects", _Rect), ("triangles", _Triangle), ("planes", _Plane), ("media", _Medium), ("materials", _Material),
           ("images", _Image), ("texels", C.c_uint8), ("top_list", C.c_int32)]


def _desc(sc):
    """rt_scene_desc as {name: ctypes array}, camera and sky (layout: 13 (pointer, count) pairs, 2 int32, camera, sky)."""
    raw = (C.c_uint64 * 26).from_address(sc.desc_ptr)
    out = {}
    for i, (name, typ) in enumerate(_ARRAYS):
        if typ is not None:
            out[name] = (typ * int(raw[2 * i + 1])).from_address(raw[2 * i]) if raw[2 * i + 1] else []
    base = sc.desc_ptr + 16 * len(_ARRAYS) + 8
    out["camera"] = _Camera.from_address(base)
    out["sky"] = _Sky.from_address(base + C.sizeof(_Camera))
    return out


def _one(items, **want):
    hits = [x for x in items if all(getattr(x, k) == v for k, v in want.items())]
    assert len(hits) == 1, (want, len(hits))
    return hits[0]


def test_cornell_flattens_to_the_values_written_by_hand(rt):
    sc = rt.Scene.load(scene_path("cornell"), ROOT)  # (kept alive: the arrays below are views of its memory)
    d = _desc(sc)
    mats = d["materials"]

    def material(i):
        return mats[i].type, mats[i].image, tuple(mats[i].colour)

    LAMB, LIGHT = 0, 3
    # the six walls: the rect's fixed axis is the coordinate both corners share, tested x, then y, then z (src/config.rs:247-279);
    # rt_rect.axis 0 = YZRect (a = y, b = z), 1 = XZRect (a = x, b = z), 2 = XYRect (a = x, b = y)
    left = _one(d["rects"], axis=0, k=555.0, facing_positive=0)
    assert (left.a0, left.a1, left.b0, left.b1) == (0.0, 555.0, 0.0, 555.0) and material(left.material) == (LAMB, -1, (0.12, 0.45, 0.15))
    right = _one(d["rects"], axis=0, k=0.0, facing_positive=1, a1=555.0)
    assert (right.a0, right.b0, right.b1) == (0.0, 0.0, 555.0) and material(right.material) == (LAMB, -1, (0.65, 0.05, 0.05))
    light = _one(d["rects"], axis=1, k=554.0)
    assert (light.a0, light.a1, light.b0, light.b1, light.facing_positive) == (214.0, 343.0, 227.0, 332.0, 0)
    assert material(light.material) == (LIGHT, -1, (15.0, 15.0, 15.0))
    floor = _one(d["rects"], axis=1, k=0.0, a1=555.0)
    assert (floor.a0, floor.b0, floor.b1, floor.facing_positive) == (0.0, 0.0, 555.0, 1) and material(floor.material) == (LAMB, -1, (0.73, 0.73, 0.73))
    ceiling = _one(d["rects"], axis=1, k=555.0)
    assert (ceiling.a0, ceiling.a1, ceiling.b0, ceiling.b1, ceiling.facing_positive) == (0.0, 555.0, 0.0, 555.0, 0)
    back = _one(d["rects"], axis=2, k=555.0)
    assert (back.a0, back.a1, back.b0, back.b1, back.facing_positive) == (0.0, 555.0, 0.0, 555.0, 0)
    # `world` order: leftWall, rightWall, floor, ceiling, backWall, light, tallBlock2, shortBlock2
    objs = d["objects"]
    rect_index = {id(r): i for i, r in enumerate(d["rects"])}
    for wall, wi in ((left, 0), (right, 1), (floor, 2), (ceiling, 3), (back, 4), (light, 5)):
        o = _one(objs, world_index=wi)
        assert (o.kind, o.xform, o.count) == (2, -1, 1) and d["rects"][o.first].k == wall.k and d["rects"][o.first].axis == wall.axis
    # the blocks: six sides in Block::new's order (src/objects.rs:149-168): XY@zmin(+), XY@zmax(-), XZ@ymin(+), XZ@ymax(-), YZ@xmin(+), YZ@xmax(-)
    for wi, (sx, sy, sz), degrees, offset in ((6, (165.0, 330.0, 165.0), 15.0, (265.0, 0.0, 295.0)),
                                             (7, (165.0, 165.0, 165.0), -18.0, (130.0, 0.0, 65.0))):
        o = _one(objs, world_index=wi)
        assert (o.kind, o.count) == (2, 6) and o.xform >= 0
        sides = [d["rects"][o.first + k] for k in range(6)]
        want = [(2, 0.0, 1, 0.0, sx, 0.0, sy), (2, sz, 0, 0.0, sx, 0.0, sy), (1, 0.0, 1, 0.0, sx, 0.0, sz), (1, sy, 0, 0.0, sx, 0.0, sz),
                (0, 0.0, 1, 0.0, sy, 0.0, sz), (0, sx, 0, 0.0, sy, 0.0, sz)]
        for k, (s, w) in enumerate(zip(sides, want)):
            assert (s.axis, s.k, s.facing_positive, s.a0, s.a1, s.b0, s.b1) == w and s.sub == k
            assert material(s.material) == (LAMB, -1, (0.73, 0.73, 0.73))
        # Translate(RotateY(block, degrees), offset): world -> object is p' = Ry (p - offset) with RotateY's forward
        # map x' = c x - s z, z' = s x + c z (src/transforms.rs:194-203)
        th = degrees * np.pi / 180.0
        c, s = np.cos(th), np.sin(th)
        m = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
        x = d["xforms"][o.xform]
        assert np.allclose(np.array(x.m).reshape(3, 3), m, rtol=0, atol=1e-15)
        assert np.allclose(np.array(x.b), -m @ np.array(offset), rtol=0, atol=1e-12)
    assert np.allclose(np.array(d["xforms"][_one(objs, world_index=6).xform].b), [-179.61872566135948, 0.0, -353.53516570744313], atol=1e-9)
    # camera (src/camera.rs:25-66): fov 40, aspect 1, focus 900 from (278, 278, -800) looking down +z
    cam = d["camera"]
    half = 900.0 * np.tan(np.radians(40.0) / 2)   # 327.5732108395821
    assert tuple(cam.origin) == (278.0, 278.0, -800.0) and cam.lens_radius == 0.0 and (cam.start_time, cam.end_time) == (0.0, 1.0)
    assert np.allclose(cam.horizontal, [-2 * half, 0, 0], atol=1e-10) and np.allclose(cam.vertical, [0, 2 * half, 0], atol=1e-10)
    assert np.allclose(cam.lower_left_corner, [278 + half, 278 - half, 100.0], atol=1e-10)
    assert np.allclose(cam.lower_left_corner, [605.5732108395822, -49.57321083958215, 100.0], atol=1e-9)
    assert d["sky"].gradient == 0 and tuple(d["sky"].colour) == (0.0, 0.0, 0.0)


def test_globe_flattens_to_the_values_written_by_hand(rt):
    sc = rt.Scene.load(scene_path("globe"), ROOT)
    d = _desc(sc)
    (s,) = list(d["spheres"])
    assert tuple(s.centre) == (0.0, 0.0, 0.0) and s.radius == 1.0
    m = d["materials"][s.material]
    assert m.type == 0 and m.image == 0     # lambertian over the image texture
    (im,) = list(d["images"])
    assert (im.width, im.height, im.texel_offset) == (1024, 512, 0)
    # Spotlight (src/objects.rs:631-717): five panes over the box [(-w/2, -len, -w/2), (w/2, 0, w/2)], w = 0.3, len = 4:
    # XY@zmin dark(+), XY@zmax dark(-), XZ@ymin LIGHT(+), YZ@xmin dark(+), YZ@xmax dark(-); no pane at y = 0
    o = _one(d["objects"], world_index=1)
    assert (o.kind, o.count) == (2, 5) and o.xform >= 0
    panes = [d["rects"][o.first + k] for k in range(5)]
    want = [(2, -0.15, 1, -0.15, 0.15, -4.0, 0.0), (2, 0.15, 0, -0.15, 0.15, -4.0, 0.0), (1, -4.0, 1, -0.15, 0.15, -0.15, 0.15),
            (0, -0.15, 1, -4.0, 0.0, -0.15, 0.15), (0, 0.15, 0, -4.0, 0.0, -0.15, 0.15)]
    for k, (p, w) in enumerate(zip(panes, want)):
        assert (p.axis, p.k, p.facing_positive, p.a0, p.a1, p.b0, p.b1) == w and p.sub == k
        pm = d["materials"][p.material]
        if k == 2:
            assert pm.type == 3 and tuple(pm.colour) == (9.0, 9.0, 2.0)      # DiffuseLight(light)
        else:
            assert pm.type == 0 and tuple(pm.colour) == (0.0, 0.0, 0.0)      # Lambertian(0, 0, 0)
    # Translate(RotateY(RotateZ(prim, t1), t2), from), dir = at - from = (3, -3, 3):
    # t1 = -acos(dot((0,1,0), unit(dir))), t2 = acos(dot((1,0,0), unit((dir.x, 0, dir.z)))) * -signum(dir.z)
    t1 = -np.arccos(-1.0 / np.sqrt(3.0))
    t2 = -np.arccos(1.0 / np.sqrt(2.0))
    c1, s1, c2, s2 = np.cos(t1), np.sin(t1), np.cos(t2), np.sin(t2)
    rz = np.array([[c1, s1, 0], [-s1, c1, 0], [0, 0, 1]])   # RotateZ forward: x' = s y + c x, y' = c y - s x (src/transforms.rs:285-294)
    ry = np.array([[c2, 0, -s2], [0, 1, 0], [s2, 0, c2]])   # RotateY forward: x' = c x - s z, z' = s x + c z
    mat = rz @ ry
    x = d["xforms"][o.xform]
    assert np.allclose(np.array(x.m).reshape(3, 3), mat, rtol=0, atol=1e-14)
    assert np.allclose(np.array(x.b), -mat @ np.array([-3.0, 3.0, -3.0]), rtol=0, atol=1e-13)
    # the housing lies behind lookFrom and the pane that glows (normal +y at y = -len) looks through the open end
    # towards lookAt: object-space +y is the world direction from -> at
    assert np.allclose(mat.T @ np.array([0, 1.0, 0]), np.array([3.0, -3.0, 3.0]) / np.sqrt(27.0), atol=1e-12)
    sky = d["sky"]
    assert sky.gradient == 1 and tuple(sky.unit_dir) == (0.0, 1.0, 0.0) and tuple(sky.colour1) == (0.05, 0.05, 0.05)
    cam = d["camera"]
    assert tuple(cam.origin) == (0.0, 0.0, -2.0)
    assert np.allclose(cam.horizontal, [-2 * 900 * np.tan(np.radians(55.0)), 0, 0], rtol=1e-14)


def test_obj_faces_flatten_to_the_vertices_written_by_hand(rt, tmp_path):
    # wavefront_obj semantics restated (src/objects.rs:719-762): 1-based indices, polygons as fans around their first vertex,
    # negative indices from the end, per-face normal = the first corner's `vn`, flipped to agree with cross(b - a, c - a)
    obj = tmp_path / "quad.obj"
    obj.write_text("v 0 0 0\nv 2 0 0\nv 2 3 0\nv 0 3 1\nvn 0 0 -1\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\nf -1 -2 -4\n")
    text = scene_text(objs='k: {type: "mesh", filename: "%s", material: "m"}' % obj, world='"k"')
    sc = rt.Scene.from_text(text)
    d = _desc(sc)
    tris = list(d["triangles"])
    assert len(tris) == 3
    by_sub = {t.sub: t for t in tris}
    t0, t1, t2 = by_sub[0], by_sub[1], by_sub[2]
    assert (tuple(t0.point), tuple(t0.vec1), tuple(t0.vec2)) == ((0.0, 0.0, 0.0), (2.0, 0.0, 0.0), (2.0, 3.0, 0.0))     # (1, 2, 3)
    assert (tuple(t1.point), tuple(t1.vec1), tuple(t1.vec2)) == ((0.0, 0.0, 0.0), (2.0, 3.0, 0.0), (0.0, 3.0, 1.0))     # (1, 3, 4)
    assert (tuple(t2.point), tuple(t2.vec1), tuple(t2.vec2)) == ((0.0, 3.0, 1.0), (2.0, 0.0, -1.0), (0.0, -3.0, -1.0))  # (4, 3, 1)
    # triangle 0: cross((2,0,0), (2,3,0)) = (0, 0, 6); the file's normal (0, 0, -1) points the other way and is flipped
    assert tuple(t0.normal) == (0.0, 0.0, 1.0)
    # triangle 2 has no normal index: unit(cross(vec1, vec2))
    n = np.cross([2.0, 0.0, -1.0], [0.0, -3.0, -1.0])
    assert np.allclose(tuple(t2.normal), n / np.linalg.norm(n), atol=1e-15)


# ---- the same independent check for the other example scenes (the oracle shares scene_config.cpp with the product, so
# only numbers worked out by hand from the .json5 text and the reference's constructors can see a wrong rule in that layer)
RT_TOP_BVH = -1
LAMB, METAL, GLASS, LIGHT, ISO, CHECK = 0, 1, 2, 3, 4, 5


def _spot_matrix(look_from, look_at):
    """world -> object rotation of Spotlight::new (src/objects.rs:665-700): Translate(RotateY(RotateZ(prim, t1), t2), from) with
    t1 = -acos(dot((0,1,0), unit(dir))), t2 = acos(dot((1,0,0), unit((dir.x, 0, dir.z)))) * -signum(dir.z), dir = at - from;
    forward maps RotateZ: x' = s y + c x, y' = c y - s x; RotateY: x' = c x - s z, z' = s x + c z (src/transforms.rs:194-203, 285-294)."""
    d = np.array(look_at, dtype=float) - np.array(look_from, dtype=float)
    t1 = -np.arccos(d[1] / np.linalg.norm(d))
    t2 = np.arccos(d[0] / np.hypot(d[0], d[2])) * -np.copysign(1.0, d[2])
    c1, s1, c2, s2 = np.cos(t1), np.sin(t1), np.cos(t2), np.sin(t2)
    rz = np.array([[c1, s1, 0], [-s1, c1, 0], [0, 0, 1]])
    ry = np.array([[c2, 0, -s2], [0, 1, 0], [s2, 0, c2]])
    return rz @ ry, d / np.linalg.norm(d)


def _check_spotlight(d, o, look_from, look_at, length, width, light):
    assert (o.kind, o.count) == (2, 5) and o.xform >= 0
    h = width / 2.0
    want = [(2, -h, 1, -h, h, -length, 0.0), (2, h, 0, -h, h, -length, 0.0), (1, -length, 1, -h, h, -h, h),
            (0, -h, 1, -length, 0.0, -h, h), (0, h, 0, -length, 0.0, -h, h)]
    for k, w in enumerate(want):
        p = d["rects"][o.first + k]
        assert (p.axis, p.k, p.facing_positive, p.a0, p.a1, p.b0, p.b1) == w and p.sub == k
        pm = d["materials"][p.material]
        assert (pm.type, tuple(pm.colour)) == ((LIGHT, tuple(light)) if k == 2 else (LAMB, (0.0, 0.0, 0.0)))
    mat, unit_dir = _spot_matrix(look_from, look_at)
    x = d["xforms"][o.xform]
    assert np.allclose(np.array(x.m).reshape(3, 3), mat, rtol=0, atol=1e-14)
    assert np.allclose(np.array(x.b), -mat @ np.array(look_from, dtype=float), rtol=0, atol=1e-10)
    assert np.allclose(mat.T @ np.array([0, 1.0, 0]), unit_dir, atol=1e-12)   # the open end looks from lookFrom towards lookAt


def _check_block(d, o, lo, hi, material):
    # Block::new (src/objects.rs:149-168): XY@zmin(+), XY@zmax(-), XZ@ymin(+), XZ@ymax(-), YZ@xmin(+), YZ@xmax(-)
    assert (o.kind, o.count) == (2, 6)
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    want = [(2, z0, 1, x0, x1, y0, y1), (2, z1, 0, x0, x1, y0, y1), (1, y0, 1, x0, x1, z0, z1), (1, y1, 0, x0, x1, z0, z1),
            (0, x0, 1, y0, y1, z0, z1), (0, x1, 0, y0, y1, z0, z1)]
    for k, w in enumerate(want):
        q = d["rects"][o.first + k]
        assert (q.axis, q.k, q.facing_positive, q.a0, q.a1, q.b0, q.b1) == w and q.sub == k
        m = d["materials"][q.material]
        assert (m.type, m.image) == material[:2] and np.allclose(tuple(m.colour) if material[2] is not None else 0, material[2] if material[2] is not None else 0)


def test_two_spheres_flattens_to_the_values_written_by_hand(rt):
    sc = rt.Scene.load(scene_path("two_spheres"), ROOT)
    d = _desc(sc)
    objs, mats = d["objects"], d["materials"]
    # world: sun, floor, floorGlass, sphere, sphereGlass, sphere2; the planes have no box and follow the BVH in the top-level list
    # (BVHNode::from_vec, src/hitting.rs:188-194), in world order
    tops = list(d["top_list"])
    assert tops[0] == RT_TOP_BVH and [objs[t].world_index for t in tops[1:]] == [1, 2] and all(objs[t].kind == 4 for t in tops[1:])
    for wi, centre, radius, mtype, mparam in ((0, (4000.0, 0.0, 0.0), 1000.0, LIGHT, None), (3, (-250.0, 100.0, 50.0), 99.5, LAMB, None),
                                              (4, (-250.0, 100.0, 50.0), 100.0, GLASS, 1.5), (5, (0.0, 100.0, 0.0), 100.0, GLASS, 1.5)):
        o = _one(objs, world_index=wi)
        s = d["spheres"][o.first]
        assert (o.kind, o.xform) == (0, -1) and tuple(s.centre) == centre and s.radius == radius and mats[s.material].type == mtype
        if mparam is not None:
            assert mats[s.material].param == mparam
    assert tuple(mats[d["spheres"][_one(objs, world_index=0).first].material].colour) == (150.0, 55.0, 0.0)
    # Plane::new (src/objects.rs:577-597): vec1 = unit(b - a); normal = unit(vec1 x (c - a)); vec2 = normal x vec1
    # floor: a = (0,-1,0), b = (0,-1,1), c = (1,-1,0): vec1 = (0,0,1), normal = (0,0,1) x (1,0,0) = (0,1,0), vec2 = (0,1,0) x (0,0,1) = (1,0,0)
    for wi, y, repeat, mtype in ((1, -1.0, 400.0, CHECK), (2, 0.0, 80.0, GLASS)):
        p = d["planes"][_one(objs, world_index=wi).first]
        assert tuple(p.point) == (0.0, y, 0.0) and tuple(p.vec1) == (0.0, 0.0, 1.0) and tuple(p.normal) == (0.0, 1.0, 0.0) and tuple(p.vec2) == (1.0, 0.0, 0.0)
        assert p.uv_repeat == repeat and mats[p.material].type == mtype
    chk = mats[d["planes"][_one(objs, world_index=1).first].material]
    assert chk.param == 2.0 and (mats[chk.odd].type, mats[chk.even].type) == (METAL, LAMB) and mats[chk.odd].param == 0.0
    assert tuple(mats[chk.odd].colour) == (0.73, 0.73, 0.73)
    # camera (src/camera.rs:25-66): w = unit(from - at) = (0, 300, -800) / sqrt(730000); u = unit(up x w) = (-1, 0, 0); v = w x u
    cam = d["camera"]
    w = np.array([0.0, 300.0, -800.0]) / np.sqrt(730000.0)
    assert np.allclose(cam.u, [-1.0, 0.0, 0.0], atol=1e-15) and np.allclose(cam.v, [0.0, -w[2], w[1]], atol=1e-15)
    vh = 2.0 * np.tan(np.radians(40.0) / 2.0)
    assert cam.lens_radius == 0.05 and np.allclose(cam.horizontal, [-100.0 * 1.6 * vh, 0, 0], rtol=1e-14)
    assert np.allclose(cam.horizontal, [-116.4704749649, 0, 0], rtol=1e-10)   # 160 x 2 tan(20 deg)
    assert np.allclose(cam.vertical, 100.0 * vh * np.array([0.0, -w[2], w[1]]), rtol=1e-14)
    llc = np.array([-100.0, 400.0, -800.0]) - np.array(cam.horizontal) / 2 - np.array(cam.vertical) / 2 - 100.0 * w
    assert np.allclose(cam.lower_left_corner, llc, rtol=1e-13)
    sky = d["sky"]
    assert sky.gradient == 1 and np.allclose(sky.unit_dir, np.array([1.0, -0.2, 0.0]) / np.sqrt(1.04), atol=1e-15)
    assert tuple(sky.colour0) == (0.0, 0.0, 0.0) and tuple(sky.colour1) == (3.0, 1.7, 0.4)


def test_knight_flattens_to_the_values_written_by_hand(rt):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_knight

    gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
    sc = rt.Scene.load(scene_path("knight"), ROOT)
    d = _desc(sc)
    objs, mats = d["objects"], d["materials"]
    # world: knight (translate of the mesh), board (plane), spotlight; "beacon" is defined but not in `world` and must not be flattened
    tops = list(d["top_list"])
    assert tops[0] == RT_TOP_BVH and len(tops) == 2 and objs[tops[1]].world_index == 1
    assert len(list(objs)) == 3 and sc.counts()["rects"] == 5
    k = _one(objs, world_index=0)
    x = d["xforms"][k.xform]
    # Translate::hit moves the ray by -offset (src/transforms.rs:22-23): world -> object p' = p - (-11, -16, 0.001)
    assert k.kind == 5 and np.array_equal(np.array(x.m).reshape(3, 3), np.eye(3)) and tuple(x.b) == (11.0, 16.0, -0.001)
    tri = d["triangles"][0]
    assert mats[tri.material].type == GLASS and mats[tri.material].param == 1.5 and sc.counts()["triangles"] > 50000
    # board: a = (0,0,0), b = (1,0,0), c = (0,1,0): vec1 = (1,0,0), normal = (1,0,0) x (0,1,0) = (0,0,1), vec2 = (0,0,1) x (1,0,0) = (0,1,0)
    p = d["planes"][_one(objs, world_index=1).first]
    assert tuple(p.point) == (0.0, 0.0, 0.0) and tuple(p.vec1) == (1.0, 0.0, 0.0) and tuple(p.normal) == (0.0, 0.0, 1.0) and tuple(p.vec2) == (0.0, 1.0, 0.0)
    chk = mats[p.material]
    assert p.uv_repeat == 12.0 and (chk.type, chk.param) == (CHECK, 2.0) and (mats[chk.odd].type, mats[chk.even].type) == (METAL, LAMB)
    assert tuple(mats[chk.even].colour) == (0.05, 0.05, 0.05)
    _check_spotlight(d, _one(objs, world_index=2), (-30.0, -30.0, 15.0), (0.0, 0.0, 5.0), 30.0, 10.0, (60.0, 60.0, 2.0))
    # the angles themselves: dir = (30, 30, -10); t1 = -acos(30 / sqrt(1900)), t2 = +acos(30 / sqrt(1000)) (dir.z < 0)
    mat, _ = _spot_matrix((-30.0, -30.0, 15.0), (0.0, 0.0, 5.0))
    t1, t2 = -0.8117261175, 0.3217505544
    assert np.allclose(mat, np.array([[np.cos(t1), np.sin(t1), 0], [-np.sin(t1), np.cos(t1), 0], [0, 0, 1]]) @
                       np.array([[np.cos(t2), 0, -np.sin(t2)], [0, 1, 0], [np.sin(t2), 0, np.cos(t2)]]), atol=1e-9)
    cam = d["camera"]
    assert tuple(cam.origin) == (-17.0, -8.0, 15.0) and cam.lens_radius == 0.0
    wv = np.array([-17.0, -8.0, 10.0]) / np.sqrt(453.0)              # unit(from - at)
    u = np.cross([0.0, 0.0, 1.0], wv)
    u /= np.linalg.norm(u)
    assert np.allclose(cam.u, u, atol=1e-15) and np.allclose(cam.v, np.cross(wv, u), atol=1e-15)
    assert np.allclose(cam.horizontal, 75.0 * 2.0 * np.tan(np.radians(20.0)) * u, rtol=1e-14)
    assert d["sky"].gradient == 0 and tuple(d["sky"].colour) == (0.4, 0.4, 0.6)


def test_blocks_ball_flattens_to_the_values_written_by_hand(rt):
    sc = rt.Scene.load(scene_path("blocks_ball"), ROOT)
    d = _desc(sc)
    objs, mats = d["objects"], d["materials"]
    c = sc.counts()
    # mist (medium) + its boundary block, ball, light (5 panes), 20 x 20 blocks
    assert (c["objects"], c["rects"], c["spheres"], c["media"], c["planes"]) == (404, 6 + 5 + 2400, 1, 1, 0)
    assert list(d["top_list"]) == [RT_TOP_BVH]
    mist = _one(objs, world_index=0)
    md = d["media"][mist.first]
    assert mist.kind == 6 and md.neg_inv_density == -1.0 / 0.0003 and (mats[md.material].type, tuple(mats[md.material].colour)) == (ISO, (1.0, 1.0, 1.0))
    bound = objs[md.boundary_object]
    assert bound.world_index == -1 and bound.xform == -1
    _check_block(d, bound, (-2000.0, -2000.0, -2000.0), (2000.0, 2000.0, 2000.0), (ISO, -1, (1.0, 1.0, 1.0)))
    ball = d["spheres"][_one(objs, world_index=1).first]
    assert tuple(ball.centre) == (310.0, 200.0, 100.0) and ball.radius == 100.0 and (mats[ball.material].type, mats[ball.material].param) == (GLASS, 1.5)
    _check_spotlight(d, _one(objs, world_index=2), (750.0, 330.0, 200.0), (310.0, 150.0, 100.0), 600.0, 120.0, (80.0, 80.0, 80.0))
    # dir = (-440, -180, -100): t1 = -acos(-180 / sqrt(236000)) = -1.950370, t2 = +acos(-440 / sqrt(203600)) = 2.918116 (dir.z < 0)
    mat, _ = _spot_matrix((750.0, 330.0, 200.0), (310.0, 150.0, 100.0))
    t1, t2 = -np.arccos(-180.0 / np.sqrt(236000.0)), np.arccos(-440.0 / np.sqrt(203600.0))
    assert abs(t1 + 1.9503696) < 1e-6 and abs(t2 - 2.9181161) < 1e-6
    assert np.allclose(mat[1], [-np.sin(t1) * np.cos(t2), np.cos(t1), np.sin(t1) * np.sin(t2)], atol=1e-14)
    # block_i_j sits at world index 3 + 20 i + j with corners (-1000 + 100 i, -1, -1000 + 100 j) .. (+100, height, +100):
    # block_0_0 is 50 high, block_0_3 79, block_12_10 56 (examples/blocks_ball.json5:70)
    for (i, j), height in (((0, 0), 50.0), ((0, 3), 79.0), ((12, 10), 56.0)):
        o = _one(objs, world_index=3 + 20 * i + j)
        assert o.xform == -1
        _check_block(d, o, (-1000.0 + 100 * i, -1.0, -1000.0 + 100 * j), (-900.0 + 100 * i, height, -900.0 + 100 * j), (LAMB, -1, (0.48, 0.83, 0.53)))
    cam = d["camera"]
    assert tuple(cam.origin) == (478.0, 378.0, -600.0) and cam.lens_radius == 0.0
    assert np.allclose(np.linalg.norm(cam.horizontal), 300.0 * (16.0 / 9.0) * 2.0 * np.tan(np.radians(20.0)), rtol=1e-14)
    assert sc.image_height(600) == 338


def test_blue_ball_flattens_to_the_values_written_by_hand(rt):
    sc = rt.Scene.load(scene_path("blue_ball"), ROOT)
    d = _desc(sc)
    objs, mats = d["objects"], d["materials"]
    c = sc.counts()
    # six walls + ONE Block shared by the world entry `tallBlock` and by the boundary of `smoke` (true instancing); three spheres
    assert (c["rects"], c["spheres"], c["media"], c["objects"]) == (12, 3, 1, 12)
    back = _one(d["rects"], axis=2, k=555.0)
    assert mats[back.material].type == GLASS and back.facing_positive == 0       # the back wall is glass here
    tall = _one(objs, world_index=6)
    smoke = _one(objs, world_index=7)
    md = d["media"][smoke.first]
    bound = objs[md.boundary_object]
    assert smoke.kind == 6 and md.neg_inv_density == -1.0 / 0.0013 and mats[md.material].type == ISO
    assert bound.world_index == -1 and bound.first == tall.first and bound.count == 6     # the same six rects
    _check_block(d, tall, (0.0, 0.1, 0.0), (165.0, 330.0, 165.0), (GLASS, -1, None))
    th = np.radians(15.0)
    m = np.array([[np.cos(th), 0, -np.sin(th)], [0, 1, 0], [np.sin(th), 0, np.cos(th)]])      # RotateY forward map (src/transforms.rs:194-203)
    for o in (tall, bound):
        x = d["xforms"][o.xform]
        assert np.allclose(np.array(x.m).reshape(3, 3), m, rtol=0, atol=1e-15)
        assert np.allclose(np.array(x.b), -m @ np.array([285.0, 0.0, 295.0]), rtol=0, atol=1e-12)
    for wi, centre, radius, mtype in ((8, (300.0, 100.1, 50.0), 100.0, GLASS), (9, (160.0, 120.1, 350.0), 120.0, GLASS), (10, (160.0, 120.1, 350.0), 110.0, LAMB)):
        s = d["spheres"][_one(objs, world_index=wi).first]
        assert tuple(s.centre) == centre and s.radius == radius and mats[s.material].type == mtype
    light = _one(d["rects"], axis=1, k=554.0)
    assert (light.a0, light.a1, light.b0, light.b1, light.facing_positive) == (214.0, 343.0, 227.0, 332.0, 0) and mats[light.material].type == LIGHT
