"""``HipRenderer``: Python host mirror of the reference's renderer class.

Same public surface and sequencing as ``WebGPURenderer``
(src/webgpu-renderer.ts:20-322) with the WebGPU objects replaced by calls into
librt_hip.so; ``Accumulator`` (src/accumulator.ts) is folded into the same
native context, which owns the three accumulation planes.  The JavaScript
drop-in with the identical surface lives in raytracer_amd/js/.

Additions the reference lacks (it only ever presents to a canvas):
``render_frames``, ``present``, ``read_accum``, ``read_ids``, ``read_frame_f16``, ``sync``.
"""
from __future__ import annotations

import sys
import time
from typing import Optional

import numpy as np

from . import capi
from . import scene as scene_mod


class HipRenderer:
    def __init__(self, width: int, height: int):
        # private constructor in the reference (src/webgpu-renderer.ts:42-46)
        self.resolution = dict(width=int(width), height=int(height))
        self.ctx: Optional[capi.Context] = None
        self.current_scene: Optional[scene_mod.Scene] = None
        self.frame_index = 0
        self.samples_per_pixel = 1
        self.debug_enabled = False
        self._enabled = True
        self._fps_frames = 0
        self._fps_last = time.perf_counter()
        self._fps = 0

    # -- construction -------------------------------------------------------

    @classmethod
    def create(cls, width: int, height: int, scene_path: str, device: int = 0, stripe_rows: int = 8,
               stripe_index: int = 0, stripe_count: int = 1, max_bounces: int = 6,
               sphere_lights: bool = False, apply_model_transforms: bool = False,
               debug_view: bool = False) -> Optional["HipRenderer"]:
        """``WebGPURenderer.Create`` (src/webgpu-renderer.ts:48-103): every
        failure is reported and ``None`` is returned; it never raises."""
        renderer = cls(width, height)
        try:
            try:
                renderer.ctx = capi.Context(width, height, device, stripe_rows, stripe_index, stripe_count)
                renderer.ctx.set_max_bounces(max_bounces)
                renderer.ctx.set_sphere_lights(sphere_lights)
                renderer.ctx.set_debug_view(debug_view)   # opt-in: set_debug(True) then draws the triangles' normal markers
            except Exception as e:  # "Failed to create pipelines"
                raise RuntimeError("Failed to create pipelines: %s" % e)
            try:
                renderer.current_scene = scene_mod.load_scene(scene_path, apply_model_transforms)
            except Exception as e:
                raise RuntimeError("Failed to load '%s': %s" % (scene_path, e))
            try:
                renderer._create_buffers()
            except Exception as e:
                raise RuntimeError("Failed to create scene buffer: %s" % e)
        except Exception as e:
            print("HIP Error:", e, file=sys.stderr)
            return None
        return renderer

    def _create_buffers(self):
        # src/webgpu-renderer.ts:141-188
        sc = self.current_scene
        self.ctx.set_scene(scene_mod.pack_spheres(sc.spheres), scene_mod.pack_triangles(sc.triangles))
        self._write_uniforms()

    def _write_uniforms(self):
        # src/webgpu-renderer.ts:214-230
        if self.ctx is None or self.current_scene is None:
            return
        u = scene_mod.pack_uniforms(self.current_scene.camera, self.frame_index, self.resolution["width"],
                                    self.resolution["height"], self.samples_per_pixel, self.debug_enabled)
        self.ctx.set_uniforms(u)

    # -- controls (src/webgpu-renderer.ts:232-235, 279-299) ------------------

    def set_samples_per_pixel(self, samples: int):
        self.samples_per_pixel = max(1, min(16, samples))
        self._write_uniforms()

    def set_debug(self, checked: bool):
        self.debug_enabled = bool(checked)
        self._write_uniforms()

    def reset_accumulation(self):
        self.frame_index = 0
        self._write_uniforms()

    def get_fps(self) -> int:
        return self._fps

    def get_enabled(self) -> bool:
        return self._enabled

    def set_enabled(self, value: bool):
        self._enabled = bool(value)

    # -- frames ---------------------------------------------------------------

    def render(self):
        """src/webgpu-renderer.ts:237-268: no-op without resources; frameIndex
        is incremented before the uniforms are written; fire and forget."""
        if self.ctx is None or self.current_scene is None:
            return
        self.frame_index += 1
        self._write_uniforms()
        self.ctx.render_frame()

    def render_frames(self, n: int):
        """n consecutive ``render()`` calls as one native submission."""
        if self.ctx is None or self.current_scene is None or n <= 0:
            return
        first = self.frame_index + 1
        self.frame_index += n
        self.ctx.render_frames(first, n)
        self._write_uniforms()

    def start_render_loop(self, frames: Optional[int] = None):
        """src/webgpu-renderer.ts:270-277 without a browser: loops ``frames``
        times (forever if None), updating the FPS counter as
        src/fps-counter.ts:6-16 does."""
        k = 0
        while frames is None or k < frames:
            self._fps_frames += 1
            now = time.perf_counter()
            if (now - self._fps_last) * 1000.0 >= 1000.0:
                self._fps = round(self._fps_frames * 1000.0 / ((now - self._fps_last) * 1000.0))
                self._fps_frames = 0
                self._fps_last = now
            if self._enabled:
                self.render()   # no fence, as in the reference; the native call keeps at most two frames in flight
            k += 1
        if self.ctx is not None:
            self.ctx.sync()

    def get_triangle_tests_since_last_check(self) -> int:
        """src/webgpu-renderer.ts:301-321 (u32, zeroed by the read)."""
        if self.ctx is None:
            return 0
        return self.ctx.take_triangle_tests()

    # -- read-back additions ----------------------------------------------------

    def sync(self):
        self.ctx.sync()

    def present(self, out: Optional[np.ndarray] = None, wait: bool = False):
        """The canvas image (what src/accumulator.ts:119-136 draws: the accumulated colour, clamped, 8 bits per
        channel) as an (H, W, 4) uint8 array, and the frameIndex it ends with.  ``wait=False`` never waits for the
        GPU: it returns the newest image that has arrived and starts the copy of the current one."""
        return self.ctx.present(out, wait)

    def read_accum(self) -> np.ndarray:
        return self.ctx.read_accum()

    def read_ids(self) -> np.ndarray:
        return self.ctx.read_primary_ids()

    def read_frame_f16(self) -> np.ndarray:
        return self.ctx.read_frame_f16()

    def close(self):
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None
