"""TEST INFRASTRUCTURE - ctypes access to the CPU oracle (oracle/liboracle.so).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this
module; the product (raytracer_amd, librt_mi355x.so) never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("time", "<f8"), ("tmin", "<f8"), ("tmax", "<f8")])
HIT_DTYPE = np.dtype(
    [("hit", "<i4"), ("owner", "<i4"), ("sub", "<i4"), ("world_index", "<i4"), ("front_face", "<i4"),
     ("material", "<i4"), ("t", "<f8"), ("p", "<f8", 3), ("n", "<f8", 3), ("u", "<f8"), ("v", "<f8"),
     ("second_t", "<f8"), ("nodes", "<u4"), ("prims", "<u4")]
)

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C oracle`")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.orc_last_error.restype = C.c_char_p
        L.orc_load_scene.restype = vp
        L.orc_load_scene.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_load_scene_text.restype = vp
        L.orc_load_scene_text.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_free_scene.argtypes = [vp]
        L.orc_image_height.restype = u32
        L.orc_image_height.argtypes = [vp, u32]
        L.orc_trace.argtypes = [vp, vp, u64, u64, vp]
        L.orc_primary_rays.argtypes = [vp, u32, u32, u32, u32, u32, u64, vp]
        L.orc_render.restype = u64
        L.orc_render.argtypes = [vp, u32, u32, u32, u32, u32, u32, u32, u64, C.c_int, vp, vp]
        L.orc_traversal_stats.argtypes = [vp, u32, u32, u32, u32, u64, u32, vp]
        L.orc_kat_cross.argtypes = [vp, vp, vp]
        L.orc_kat_refract.argtypes = [vp, vp, C.c_double, vp]
        L.orc_kat_reflect.argtypes = [vp, vp, vp]
        L.orc_kat_line_plane.restype = C.c_int
        L.orc_kat_line_plane.argtypes = [vp] * 6
        L.orc_kat_unit_vectors.argtypes = [u64, u32, vp]
        L.orc_kat_colour_to_raw.argtypes = [vp, vp]
        L.orc_philox.argtypes = [vp, vp, vp]
        L.orc_uniforms.argtypes = [u64, u32, u32, u32, u32, u32, vp]
        _lib = L
    return _lib


def _v3(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(3))


class OracleScene:
    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, path, asset_root=None):
        h = lib().orc_load_scene(os.fsencode(path), os.fsencode(asset_root) if asset_root else None)
        if not h:
            raise RuntimeError(lib().orc_last_error().decode("utf-8", "replace"))
        return cls(h)

    @classmethod
    def from_text(cls, text, asset_root=None):
        h = lib().orc_load_scene_text(text.encode("utf-8"), os.fsencode(asset_root) if asset_root else None)
        if not h:
            raise RuntimeError(lib().orc_last_error().decode("utf-8", "replace"))
        return cls(h)

    def image_height(self, width):
        return int(lib().orc_image_height(self._h, width))

    def trace(self, rays, seed=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        lib().orc_trace(self._h, rays.ctypes.data, rays.shape[0], seed, out.ctypes.data)
        return out

    def primary_rays(self, width, height, j0, j1, sample=0, seed=0):
        out = np.zeros((j1 - j0) * width, dtype=RAY_DTYPE)
        lib().orc_primary_rays(self._h, width, height, j0, j1, sample, seed, out.ctypes.data)
        return out

    def render(self, width, height, j0, j1, spp, max_bounces, seed=0, s0=0, threads=1):
        rows = j1 - j0
        sums = np.zeros((rows, width, 3), dtype=np.float64)
        rgb8 = np.zeros((rows, width, 3), dtype=np.uint8)
        rays = lib().orc_render(self._h, width, height, j0, j1, s0, spp, max_bounces, seed, threads,
                                sums.ctypes.data, rgb8.ctypes.data)
        return rgb8, sums, int(rays)

    def traversal_stats(self, width, height, spp, max_bounces, seed=0, pixel_stride=1):
        out = np.zeros(5)
        lib().orc_traversal_stats(self._h, width, height, spp, max_bounces, seed, pixel_stride, out.ctypes.data)
        return {"rays": float(out[0]), "nodes_per_ray": float(out[1]), "prims_per_ray": float(out[2]),
                "leaves_per_ray": float(out[3]), "xforms_per_ray": float(out[4])}

    def close(self):
        if self._h:
            lib().orc_free_scene(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cross(a, b):
    out = np.zeros(3)
    a, b = _v3(a), _v3(b)  # keep the temporaries alive across the call
    lib().orc_kat_cross(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def refract(d, n, eta):
    out = np.zeros(3)
    d, n = _v3(d), _v3(n)
    lib().orc_kat_refract(d.ctypes.data, n.ctypes.data, float(eta), out.ctypes.data)
    return out


def reflect(d, n):
    out = np.zeros(3)
    d, n = _v3(d), _v3(n)
    lib().orc_kat_reflect(d.ctypes.data, n.ctypes.data, out.ctypes.data)
    return out


def line_plane(la, lab, p0, p01, p02):
    out = np.zeros(3)
    args = [_v3(x) for x in (la, lab, p0, p01, p02)]
    ok = lib().orc_kat_line_plane(*[a.ctypes.data for a in args], out.ctypes.data)
    return out if ok else None


def unit_vectors(seed, n):
    out = np.zeros((n, 3))
    lib().orc_kat_unit_vectors(seed, n, out.ctypes.data)
    return out


def colour_to_raw(c):
    out = np.zeros(3, dtype=np.uint8)
    c = _v3(c)
    lib().orc_kat_colour_to_raw(c.ctypes.data, out.ctypes.data)
    return out


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def uniforms(seed, domain, pixel, sample, bounce, n):
    out = np.zeros(n)
    lib().orc_uniforms(seed, domain, pixel, sample, bounce, n, out.ctypes.data)
    return out
