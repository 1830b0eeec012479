"""raytracer_amd - Python bindings (ctypes) over the C ABI of librt_mi355x.so.

The product is the shared library (include/rt_mi355x.h); this module only loads it and
gives tests / bench.py numpy-friendly access.  There is no Python or CPU fallback: if the
library is missing, importing the bindings' entry points raises.

Reference mapping (FBemf/raytracer @ v0):
  Scene.load      <- config::load_config            (src/config.rs:19-63)
  Renderer.trace  <- Hittable::hit on the world     (src/hitting.rs:82-86, :21)
  Renderer.render_rows <- the row closure           (src/main.rs:124-146)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

__all__ = [
    "load_library", "Scene", "Renderer", "MultiRenderer", "split_samples_c", "RtError", "RtStats", "RtUploadStats", "RAY_DTYPE", "HIT_DTYPE",
    "PRECISION_F64", "PRECISION_F32", "LIB_PATH", "PartFile", "read_part_file", "save_image", "ProgressBar",
]

# RT_MI355X_LIB overrides the library path (A/B runs of kernel variants); default: the in-tree build.
LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librt_mi355x.so")

NOALLOC_MAX_PATHS = 131072  # RT_NOALLOC_MAX_PATHS
PRECISION_F64 = 0
PRECISION_F32 = 1

# rt_ray / rt_hit as numpy structured dtypes (layout checked against the C structs in tests).
RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("time", "<f8"), ("tmin", "<f8"), ("tmax", "<f8")])
HIT_DTYPE = np.dtype(
    [("hit", "<i4"), ("owner", "<i4"), ("sub", "<i4"), ("world_index", "<i4"), ("front_face", "<i4"),
     ("material", "<i4"), ("t", "<f8"), ("p", "<f8", 3), ("n", "<f8", 3), ("u", "<f8"), ("v", "<f8"),
     ("second_t", "<f8"), ("inst_id", "<i4"), ("pad", "<i4")]
)


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt error {code}: {message}")
        self.code = code
        self.message = message


class RtOptions(C.Structure):
    _fields_ = [("precision", C.c_int32), ("lanes", C.c_int32), ("max_paths", C.c_uint64)]


class RtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64), ("paths", C.c_uint64), ("ms_total", C.c_double), ("ms_extend", C.c_double),
        ("ms_shade", C.c_double), ("extend_launches", C.c_uint32), ("shade_launches", C.c_uint32),
        ("batches", C.c_uint32), ("max_depth_reached", C.c_uint32),
        ("ms_tail", C.c_double), ("rays_tail", C.c_uint64), ("tail_launches", C.c_uint32), ("extend_kernel", C.c_uint32),
        ("paths_tail", C.c_uint64), ("paths_first", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class RtUploadStats(C.Structure):
    _fields_ = [
        ("ms_validate", C.c_double), ("ms_build", C.c_double), ("ms_copy", C.c_double), ("ms_total", C.c_double),
        ("bytes_device", C.c_uint64), ("n_nodes2", C.c_uint32), ("n_nodes4", C.c_uint32), ("sah_trees", C.c_uint32),
        ("needs_uv", C.c_uint32), ("gpu_built_trees", C.c_uint32), ("walk", C.c_uint32), ("ms_gpu_build", C.c_double),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


# Every symbol include/rt_mi355x.h declares: (name, restype, argtypes).
_u32, _u64, _vp = C.c_uint32, C.c_uint64, C.c_void_p
EXPORTS = [
    ("rt_scene_load_file", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    ("rt_scene_load_text", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    ("rt_scene_free", None, [_vp]),
    ("rt_scene_get_desc", _vp, [_vp]),
    ("rt_scene_image_height", _u32, [_vp, _u32]),
    ("rt_last_global_error", C.c_char_p, []),
    ("rt_create", C.c_int, [C.c_int, C.POINTER(RtOptions), C.POINTER(_vp)]),
    ("rt_destroy", None, [_vp]),
    ("rt_last_error", C.c_char_p, [_vp]),
    ("rt_upload_scene", C.c_int, [_vp, _vp]),
    ("rt_get_upload_stats", C.c_int, [_vp, C.POINTER(RtUploadStats)]),
    ("rt_set_launch_timing", C.c_int, [_vp, C.c_int]),
    ("rt_trace", C.c_int, [_vp, _vp, C.c_size_t, _u64, _vp]),
    ("rt_primary_rays", C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u64, _vp]),
    ("rt_render_accum_device", C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u64, _vp, C.POINTER(RtStats)]),
    ("rt_render_accum_device_noalloc", C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u64, _vp]),
    ("rt_quantize_device", C.c_int, [_vp, _vp, _u64, _u32, _vp]),
    ("rt_render_rows", C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u64, _vp, _vp, C.POINTER(RtStats)]),
    # several GPUs, one process
    ("rt_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(RtOptions), C.POINTER(_vp)]),
    ("rt_multi_destroy", None, [_vp]),
    ("rt_multi_device_count", C.c_int, [_vp]),
    ("rt_multi_last_error", C.c_char_p, [_vp]),
    ("rt_multi_rccl_library", C.c_char_p, []),
    ("rt_multi_upload_scene", C.c_int, [_vp, _vp]),
    ("rt_multi_set_launch_timing", C.c_int, [_vp, C.c_int]),
    ("rt_multi_render_rows", C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u64, _u32, _vp, _vp, C.POINTER(RtStats)]),
    ("rt_multi_render_rows_cb", C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u64, _u32, _vp, _vp, C.POINTER(RtStats), _vp, _vp]),
    ("rt_split_samples", None, [_u32, _u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    # host I/O
    ("rt_part_file_create", C.c_int, [C.c_char_p, _u32, _u32, C.POINTER(_vp)]),
    ("rt_part_file_path", C.c_char_p, [_vp]),
    ("rt_part_file_write", C.c_int, [_vp, _u32, _vp, C.c_size_t]),
    ("rt_part_file_finish", C.c_int, [_vp, C.c_int]),
    ("rt_part_file_read", C.c_int, [C.c_char_p, _u32, _u32, C.c_int, _vp, _vp]),
    ("rt_save_image", C.c_int, [C.c_char_p, _u32, _u32, _vp]),
    ("rt_progress_create", C.c_int, [C.c_int, C.c_size_t, C.c_char_p, C.c_char_p, _u32, _u32, C.POINTER(_vp)]),
    ("rt_progress_update", None, [_vp]),
    ("rt_progress_clear", None, [_vp]),
    ("rt_progress_free", None, [_vp]),
]

_lib = None


def _share_torch_hip_runtime() -> None:
    """PyTorch wheels bundle their own libamdhip64.so (soname libamdhip64.so.7).  If this
    library pulled in /opt/rocm's copy first, a later `import torch` would load a second HIP
    runtime into the process and find no GPU.  Loading torch's copy first (no torch import
    needed) makes both share one runtime whatever the import order; without torch installed
    the system runtime is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    # (RCCL, which rt_multi_create loads on demand, is then taken from next to that runtime by the library itself: multi.cpp)


def load_library() -> C.CDLL:
    """Loads librt_mi355x.so (built in-tree by `make` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` (hipcc, gfx950). There is no fallback path."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in EXPORTS:
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class Scene:
    """A flattened scene on the host (rt_scene)."""

    def __init__(self, handle: int):
        self._h = handle

    @classmethod
    def load(cls, path: str, asset_root: Optional[str] = None) -> "Scene":
        lib = load_library()
        out = _vp()
        rc = lib.rt_scene_load_file(os.fsencode(path), os.fsencode(asset_root) if asset_root else None, C.byref(out))
        if rc != 0:
            raise RtError(rc, lib.rt_last_global_error().decode("utf-8", "replace"))
        return cls(out.value)

    @classmethod
    def from_text(cls, text: str, asset_root: Optional[str] = None) -> "Scene":
        lib = load_library()
        out = _vp()
        rc = lib.rt_scene_load_text(text.encode("utf-8"), os.fsencode(asset_root) if asset_root else None, C.byref(out))
        if rc != 0:
            raise RtError(rc, lib.rt_last_global_error().decode("utf-8", "replace"))
        return cls(out.value)

    @property
    def desc_ptr(self) -> int:
        return load_library().rt_scene_get_desc(self._h)

    def image_height(self, width: int) -> int:
        return int(load_library().rt_scene_image_height(self._h, width))

    def counts(self) -> dict:
        """Element counts of the flattened arrays (reads rt_scene_desc)."""
        names = ["nodes", "objects", "xforms", "spheres", "moving_spheres", "rects", "triangles", "planes",
                 "media", "materials", "images", "texels", "top_list"]
        raw = (C.c_uint64 * (2 * len(names))).from_address(self.desc_ptr)
        out = {n: int(raw[2 * i + 1]) for i, n in enumerate(names)}
        tail = (C.c_int32 * 2).from_address(self.desc_ptr + 16 * len(names))
        out["tlas_root"], out["stack_depth"] = int(tail[0]), int(tail[1])
        return out

    def close(self) -> None:
        if self._h:
            load_library().rt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """One GPU's render context (rt_context)."""

    def __init__(self, device: int = 0, precision: int = PRECISION_F64, max_paths: int = 0, lanes: int = 0):
        lib = load_library()
        opts = RtOptions(precision, lanes, max_paths)
        out = _vp()
        rc = lib.rt_create(device, C.byref(opts), C.byref(out))
        if rc != 0:
            raise RtError(rc, lib.rt_last_global_error().decode("utf-8", "replace"))
        self._h = out.value
        self._lib = lib
        self.device = device

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(self._h).decode("utf-8", "replace"))

    def upload(self, scene: Scene) -> None:
        self._check(self._lib.rt_upload_scene(self._h, scene.desc_ptr))

    def upload_stats(self) -> RtUploadStats:
        out = RtUploadStats()
        self._check(self._lib.rt_get_upload_stats(self._h, C.byref(out)))
        return out

    def set_launch_timing(self, on: bool) -> None:
        """HIP events around every launch (fills RtStats.ms_extend / ms_shade / ms_tail; costs ~5 us per event)."""
        self._check(self._lib.rt_set_launch_timing(self._h, int(on)))

    def trace(self, rays: np.ndarray, seed: int = 0) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self._check(self._lib.rt_trace(self._h, rays.ctypes.data, rays.shape[0], seed, out.ctypes.data))
        return out

    def primary_rays(self, width: int, height: int, j0: int, j1: int, sample: int = 0, seed: int = 0) -> np.ndarray:
        out = np.zeros((max(j1 - j0, 0)) * width, dtype=RAY_DTYPE)
        self._check(self._lib.rt_primary_rays(self._h, width, height, j0, j1, sample, seed, out.ctypes.data))
        return out

    def render_rows(self, width: int, height: int, j0: int, j1: int, spp: int, max_bounces: int, seed: int = 0,
                    want_sums: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray], RtStats]:
        rows = max(j1 - j0, 0)
        rgb8 = np.zeros((rows, width, 3), dtype=np.uint8)
        sums = np.zeros((rows, width, 3), dtype=np.float64) if want_sums else None
        stats = RtStats()
        self._check(self._lib.rt_render_rows(self._h, width, height, j0, j1, spp, max_bounces, seed, rgb8.ctypes.data,
                                             sums.ctypes.data if want_sums else None, C.byref(stats)))
        return rgb8, sums, stats

    def render_accum_device(self, width: int, height: int, j0: int, j1: int, s0: int, n_samples: int,
                            max_bounces: int, seed: int, accum_ptr: int, want_stats: bool = True) -> Optional[RtStats]:
        """accum_ptr: device pointer to (j1-j0)*width*3 float64 (e.g. torch tensor .data_ptr())."""
        stats = RtStats() if want_stats else None
        self._check(self._lib.rt_render_accum_device(self._h, width, height, j0, j1, s0, n_samples, max_bounces, seed,
                                                     accum_ptr, C.byref(stats) if want_stats else None))
        return stats

    def render_accum_device_noalloc(self, width: int, height: int, j0: int, j1: int, s0: int, n_samples: int,
                                    max_bounces: int, seed: int, accum_ptr: int) -> None:
        """rt_render_accum_device for callers that cannot allocate: work buffers in the library's static device array,
        at most NOALLOC_MAX_PATHS pixels per call."""
        self._check(self._lib.rt_render_accum_device_noalloc(self._h, width, height, j0, j1, s0, n_samples, max_bounces,
                                                             seed, accum_ptr))

    def quantize_device(self, accum_ptr: int, n_pixels: int, spp_total: int, rgb8_ptr: int) -> None:
        self._check(self._lib.rt_quantize_device(self._h, accum_ptr, n_pixels, spp_total, rgb8_ptr))

    def close(self) -> None:
        if self._h:
            self._lib.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiRenderer:
    """One image over several GPUs from this process (rt_multi): sample split, one RCCL reduce per band."""

    def __init__(self, devices, precision: int = PRECISION_F64, max_paths: int = 0, lanes: int = 0):
        lib = load_library()
        opts = RtOptions(precision, lanes, max_paths)
        ids = (C.c_int * len(devices))(*devices)
        out = _vp()
        rc = lib.rt_multi_create(ids, len(devices), C.byref(opts), C.byref(out))
        if rc != 0:
            raise RtError(rc, lib.rt_multi_last_error(None).decode("utf-8", "replace"))
        self._h = out.value
        self._lib = lib
        self.devices = list(devices)

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RtError(rc, self._lib.rt_multi_last_error(self._h).decode("utf-8", "replace"))

    def upload(self, scene: Scene) -> None:
        self._check(self._lib.rt_multi_upload_scene(self._h, scene.desc_ptr))

    def render_rows(self, width: int, height: int, j0: int, j1: int, spp: int, max_bounces: int, seed: int = 0,
                    band_rows: int = 0, want_sums: bool = True, on_band=None) -> Tuple[np.ndarray, Optional[np.ndarray], RtStats]:
        """on_band(b0, b1): called in band order when rows [b0, b1) are final."""
        rows = max(j1 - j0, 0)
        rgb8 = np.zeros((rows, width, 3), dtype=np.uint8)
        sums = np.zeros((rows, width, 3), dtype=np.float64) if want_sums else None
        stats = RtStats()
        if on_band is None:
            self._check(self._lib.rt_multi_render_rows(self._h, width, height, j0, j1, spp, max_bounces, seed, band_rows,
                                                       rgb8.ctypes.data, sums.ctypes.data if want_sums else None, C.byref(stats)))
        else:
            cb = C.CFUNCTYPE(None, _vp, _u32, _u32)(lambda user, b0, b1: on_band(int(b0), int(b1)))
            self._check(self._lib.rt_multi_render_rows_cb(self._h, width, height, j0, j1, spp, max_bounces, seed, band_rows,
                                                          rgb8.ctypes.data, sums.ctypes.data if want_sums else None, C.byref(stats),
                                                          C.cast(cb, _vp), None))
        return rgb8, sums, stats

    def close(self) -> None:
        if self._h:
            self._lib.rt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def split_samples_c(total: int, parts: int, part: int, first: int = 0) -> Tuple[int, int]:
    """rt_split_samples: the library's partition rule (the same as parallel.split_samples)."""
    b, c = _u32(), _u32()
    load_library().rt_split_samples(total, parts, part, first, C.byref(b), C.byref(c))
    return int(b.value), int(c.value)


def _check_global(rc: int) -> None:
    if rc != 0:
        raise RtError(rc, load_library().rt_last_global_error().decode("utf-8", "replace"))


class PartFile:
    """A .PART recovery file being written (PartFile::new / write_part, src/part_file.rs:13-39)."""

    def __init__(self, image_name: str, width: int, height: int):
        lib = load_library()
        out = _vp()
        _check_global(lib.rt_part_file_create(os.fsencode(image_name), width, height, C.byref(out)))
        self._h = out.value
        self.path = os.fsdecode(lib.rt_part_file_path(self._h))

    def write(self, line_number: int, row) -> None:
        row = np.ascontiguousarray(row, dtype=np.uint8)
        _check_global(load_library().rt_part_file_write(self._h, line_number, row.ctypes.data, row.size))

    def finish(self, remove: bool) -> None:
        h, self._h = self._h, None
        if h:
            _check_global(load_library().rt_part_file_finish(h, int(remove)))

    def __del__(self):
        try:
            self.finish(False)
        except Exception:
            pass


def read_part_file(name: str, height: int, width: int, recover_corrupt: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """PartFile::read (src/part_file.rs:41-99): (rows [height, width, 3] uint8 indexed by j, present [height] bool)."""
    rows = np.zeros((height, width, 3), dtype=np.uint8)
    present = np.zeros(height, dtype=np.uint8)
    _check_global(load_library().rt_part_file_read(os.fsencode(name), height, width, int(recover_corrupt),
                                                   rows.ctypes.data, present.ctypes.data))
    return rows, present.astype(bool)


def save_image(path: str, rgb8) -> None:
    """img.save (src/main.rs:161-162): rgb8 [height, width, 3] uint8, top row first."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if rgb8.ndim != 3 or rgb8.shape[2] != 3:
        raise ValueError("rgb8 must be [height, width, 3]")
    _check_global(load_library().rt_save_image(os.fsencode(path), rgb8.shape[1], rgb8.shape[0], rgb8.ctypes.data))


class ProgressBar:
    """TimedProgressBar (src/progress.rs:12-115) on a file descriptor (default stderr)."""

    def __init__(self, length: int, label: str, glyphs: str, number_samples: int, times_will_be_updated: int, fd: int = 2):
        out = _vp()
        _check_global(load_library().rt_progress_create(fd, length, label.encode(), glyphs.encode(), number_samples,
                                                        times_will_be_updated, C.byref(out)))
        self._h = out.value

    def update(self) -> None:
        load_library().rt_progress_update(self._h)

    def clear(self) -> None:
        load_library().rt_progress_clear(self._h)

    def close(self) -> None:
        h, self._h = self._h, None
        if h:
            load_library().rt_progress_free(h)

    def __del__(self):
        self.close()
