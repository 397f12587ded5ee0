"""ctypes binding of librt_hip.so (the C ABI declared in include/rt_hip.h).

There is no fallback: if the library is missing or no GPU is usable the calls
raise ``RtError``.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIBRARY names another build of the library (A/B runs of compiler options); the default is the in-tree one
LIB_PATH = os.environ.get("RT_HIP_LIBRARY") or os.path.join(_HERE, "librt_hip.so")

RT_MODE_BVH = 0
RT_MODE_BRUTE_FORCE = 1
RT_MODE_BVH_TILES = 2
RT_TUNE_SHADE_NUM, RT_TUNE_LEAF_LANES, RT_TUNE_BLOCKS_PER_CU, RT_TUNE_BATCH_FRAMES, RT_TUNE_SHADOW_DONE_LANES = 0, 1, 2, 3, 4
RT_TUNE_REFILL_LANES = 5
RT_TUNE_MAX_STACK = 11
RT_TUNE_COALESCE_FRAMES = 12
RT_TUNE_MIGRATE_LANES = 13
RT_TUNE_TILE_ORDER = 14
RT_TUNE_LINEAR_SPHERES = 15
RT_TUNE_MAX_LEAF, RT_TUNE_TRI_COST_X10, RT_TUNE_LDS_STATE, RT_TUNE_OVERLAP_FRAMES, RT_TUNE_SHADE_WAIT = 6, 7, 8, 9, 10

# name -> (restype, argtypes); must list every symbol include/rt_hip.h declares
SYMBOLS = {
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_set_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "rt_set_uniforms": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_set_max_bounces": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rt_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_sphere_lights": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_debug_view": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_counting": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_render_frame": (C.c_int, [C.c_void_p]),
    "rt_render_frames": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rt_render_frames_noalloc": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rt_debug_phase_cycles": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_debug_wave_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rt_sync": (C.c_int, [C.c_void_p]),
    "rt_last_render_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rt_last_render_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rt_read_accum": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_present": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]),
    "rt_write_accum": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_read_frame_f16": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_read_primary_ids": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_read_counters": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_take_triangle_tests": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rt_owned_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "rt_export_owned_rows": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_import_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rt_import_gathered_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rt_read_owned_rows": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_rows_of_stripe": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "rt_write_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rt_clear_accum": (C.c_int, [C.c_void_p]),
    "rt_render_frame_sums": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_export_accum": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_import_accum": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_finish_frame_sums": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rt_frames_per_launch": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rt_scene_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
}


class RtError(RuntimeError):
    pass


_lib = None


def _share_torch_hip_runtime():
    """A process can hold one HIP runtime.  PyTorch's wheel bundles its own libamdhip64 (same
    soname as the system one); if the system copy gets loaded first, for librt_hip.so, torch later
    finds no GPU.  So when torch is installed, load its copy first (without importing torch) and
    let librt_hip.so bind to it.  RT_HIP_SYSTEM_RUNTIME=1 keeps the system runtime."""
    if os.environ.get("RT_HIP_SYSTEM_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError):
        pass


def load():
    """Load librt_hip.so and declare every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError("librt_hip.so is not built (run __graft_entry__.build() or make -C raytracer_amd/csrc)")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc: int):
    if rc != 0:
        msg = load().rt_last_error()
        raise RtError("rt_hip error %d: %s" % (rc, msg.decode() if msg else ""))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """Thin object wrapper over one ``rt_context``."""

    def __init__(self, width: int, height: int, device: int = 0, stripe_rows: int = 8,
                 stripe_index: int = 0, stripe_count: int = 1):
        self.lib = load()
        self.width, self.height = width, height
        self.handle = C.c_void_p()
        _check(self.lib.rt_create(device, width, height, stripe_rows, stripe_index, stripe_count,
                                  C.byref(self.handle)))

    def close(self):
        if self.handle:
            self.lib.rt_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, spheres: np.ndarray, triangles: np.ndarray):
        spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 16)
        triangles = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 24)
        _check(self.lib.rt_set_scene(self.handle, _ptr(spheres) if len(spheres) else None, len(spheres),
                                     _ptr(triangles) if len(triangles) else None, len(triangles)))

    def set_uniforms(self, uniforms: np.ndarray):
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        assert u.nbytes == 80
        _check(self.lib.rt_set_uniforms(self.handle, _ptr(u)))

    def set_max_bounces(self, n: int):
        _check(self.lib.rt_set_max_bounces(self.handle, n))

    def set_mode(self, mode: int):
        _check(self.lib.rt_set_mode(self.handle, mode))

    def set_tuning(self, key: int, value: int):
        _check(self.lib.rt_set_tuning(self.handle, key, value))

    def set_counting(self, enable: bool):
        _check(self.lib.rt_set_counting(self.handle, 1 if enable else 0))

    def render_frame(self):
        _check(self.lib.rt_render_frame(self.handle))

    def render_frames(self, first_frame_index: int, n: int):
        _check(self.lib.rt_render_frames(self.handle, first_frame_index, n))

    def render_frames_noalloc(self, first_frame_index: int, n: int):
        _check(self.lib.rt_render_frames_noalloc(self.handle, first_frame_index, n))

    def sync(self):
        _check(self.lib.rt_sync(self.handle))

    def last_render_ms(self) -> float:
        ms = C.c_float()
        _check(self.lib.rt_last_render_ms(self.handle, C.byref(ms)))
        return ms.value

    def last_render_frames(self) -> int:
        v = C.c_uint32()
        _check(self.lib.rt_last_render_frames(self.handle, C.byref(v)))
        return v.value

    def read_accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        _check(self.lib.rt_read_accum(self.handle, _ptr(out)))
        return out

    def present(self, out: np.ndarray = None, wait: bool = False):
        """(RGBA8 image (H, W, 4), frameIndex it ends with); wait=False: the newest image that has arrived."""
        if out is None:
            out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.size == self.width * self.height * 4 and out.flags["C_CONTIGUOUS"]
        frame = C.c_uint32()
        _check(self.lib.rt_present(self.handle, _ptr(out), 1 if wait else 0, C.byref(frame)))
        return out, frame.value

    def write_accum(self, rgb: np.ndarray):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        assert rgb.size == self.width * self.height * 3
        _check(self.lib.rt_write_accum(self.handle, _ptr(rgb)))

    def read_frame_f16(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.uint16)
        _check(self.lib.rt_read_frame_f16(self.handle, _ptr(out)))
        return out

    def read_primary_ids(self) -> np.ndarray:
        out = np.empty((self.height, self.width), dtype=np.int32)
        _check(self.lib.rt_read_primary_ids(self.handle, _ptr(out)))
        return out

    def read_counters(self) -> np.ndarray:
        out = np.zeros(16, dtype=np.uint64)
        _check(self.lib.rt_read_counters(self.handle, _ptr(out)))
        return out

    def wave_trace(self) -> np.ndarray:
        """(waves, 4) uint64: start, queues-empty, end (100 MHz ticks), live lanes at queues-empty."""
        n = C.c_size_t()
        _check(self.lib.rt_debug_wave_trace(self.handle, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            _check(self.lib.rt_debug_wave_trace(self.handle, _ptr(out), n.value, C.byref(n)))
        return out.reshape(-1, 4)

    def take_triangle_tests(self) -> int:
        v = C.c_uint32()
        _check(self.lib.rt_take_triangle_tests(self.handle, C.byref(v)))
        return v.value

    def owned_rows(self) -> int:
        v = C.c_int()
        _check(self.lib.rt_owned_rows(self.handle, C.byref(v)))
        return v.value

    def export_owned_rows(self, device_ptr: int):
        _check(self.lib.rt_export_owned_rows(self.handle, C.c_void_p(device_ptr)))

    def import_rows(self, device_ptr: int, from_index: int):
        _check(self.lib.rt_import_rows(self.handle, C.c_void_p(device_ptr), from_index))

    def import_gathered_rows(self, device_ptr: int, rows_per_block: int, skip_index: int = -1):
        _check(self.lib.rt_import_gathered_rows(self.handle, C.c_void_p(device_ptr), rows_per_block, skip_index))

    def rows_of_stripe(self, stripe_index: int) -> int:
        v = C.c_int()
        _check(self.lib.rt_rows_of_stripe(self.handle, stripe_index, C.byref(v)))
        return v.value

    def read_owned_rows(self) -> np.ndarray:
        out = np.empty((3, self.owned_rows(), self.width), dtype=np.float32)
        _check(self.lib.rt_read_owned_rows(self.handle, _ptr(out)))
        return out

    def write_rows(self, planes: np.ndarray, from_index: int):
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        _check(self.lib.rt_write_rows(self.handle, _ptr(planes), from_index))

    def clear_accum(self):
        _check(self.lib.rt_clear_accum(self.handle))

    def render_frame_sums(self, first_frame_index: int, n_frames: int, stride: int = 1):
        _check(self.lib.rt_render_frame_sums(self.handle, first_frame_index, n_frames, stride))

    def export_accum(self, device_ptr: int):
        _check(self.lib.rt_export_accum(self.handle, C.c_void_p(device_ptr)))

    def import_accum(self, device_ptr: int):
        _check(self.lib.rt_import_accum(self.handle, C.c_void_p(device_ptr)))

    def finish_frame_sums(self, n_frames_total: int):
        _check(self.lib.rt_finish_frame_sums(self.handle, n_frames_total))

    def frames_per_launch(self) -> int:
        v = C.c_uint32()
        _check(self.lib.rt_frames_per_launch(self.handle, C.byref(v)))
        return v.value

    def set_debug_view(self, enable: bool):
        _check(self.lib.rt_set_debug_view(self.handle, 1 if enable else 0))

    def set_sphere_lights(self, enable: bool):
        _check(self.lib.rt_set_sphere_lights(self.handle, 1 if enable else 0))

    def scene_stats(self) -> dict:
        out = np.zeros(4, dtype=np.uint64)
        _check(self.lib.rt_scene_stats(self.handle, _ptr(out)))
        return dict(nodes=int(out[0]), leaves=int(out[1]), max_depth=int(out[2]), device_bytes=int(out[3]))
