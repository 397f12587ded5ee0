"""ctypes binding of the host build of the renderer's shared core
(tests/harness/cpu_harness.cpp).  Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "tests", "_build", "libcpu_harness.so")


def lib():
    subprocess.check_call(["make", "-C", os.path.join(_ROOT, "tests", "harness"), "-s"])
    L = C.CDLL(_SO)
    L.har_render_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.har_render_frame.restype = C.c_int
    L.har_render_frames_lanes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int]
    L.har_render_frames_lanes.restype = C.c_int
    L.har_check_link_packing.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.har_check_link_packing.restype = C.c_int
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class HarnessRenderer:
    def __init__(self, scene, width, height, brute=False, max_bounces=6, max_leaf=0, sphere_lights=False):
        from raytracer_amd import scene as S

        self.S, self.scene, self.width, self.height = S, scene, width, height
        self.brute, self.max_bounces, self.max_leaf = brute, max_bounces, max_leaf
        self.sphere_lights = sphere_lights
        if isinstance(scene, tuple):  # (camera, packed spheres, packed triangles)
            self.scene = S.Scene(camera=scene[0])
            self.spheres, self.tris = scene[1], scene[2]
        else:
            self.spheres, self.tris = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
        self.accum = np.zeros((3, height, width), np.float32)
        self.frame_f16 = np.zeros((height, width, 4), np.uint16)
        self.ids = np.zeros((height, width), np.int32)
        self.counters = np.zeros(16, np.uint64)
        self.frame_index = 0
        self.L = lib()

    def render(self, spp):
        self.frame_index += 1
        u = self.S.pack_uniforms(self.scene.camera, self.frame_index, self.width, self.height, spp)
        self.L.har_set_sphere_lights(1 if self.sphere_lights else 0)
        rc = self.L.har_render_frame(_p(u), _p(self.spheres), len(self.spheres), _p(self.tris), len(self.tris),
                                     self.max_bounces, 1 if self.brute else 0, self.max_leaf, _p(self.accum),
                                     _p(self.frame_f16), _p(self.ids), _p(self.counters), os.cpu_count() or 1)
        assert rc == 0


    def render_frames_lanes(self, spp, first_frame, n_frames, shade_num=3, leaf_lanes=8, packed_links=False, quant_nodes=False):
        """The persistent kernel's per-lane state machine, 64 lanes per emulated wave.  packed_links:
        child links are read as the kernels read them for this scene (from the low bytes of the node
        planes when the builder packed them there); returns the link width used (0 or 16)."""
        u = self.S.pack_uniforms(self.scene.camera, first_frame, self.width, self.height, spp)
        # quant_nodes: the lanes visit the 64-byte nodes (a build option of the large-scene kernels, -DRT_LARGE_NODES=64)
        self.L.har_set_link_mode(64 if quant_nodes else -1 if packed_links else 0)
        self.L.har_set_sphere_lights(1 if self.sphere_lights else 0)
        rc = self.L.har_render_frames_lanes(_p(u), _p(self.spheres), len(self.spheres), _p(self.tris), len(self.tris),
                                            self.max_bounces, first_frame, n_frames, shade_num, leaf_lanes,
                                            _p(self.accum), _p(self.frame_f16), _p(self.ids), _p(self.counters),
                                            os.cpu_count() or 1)
        self.L.har_set_link_mode(0)
        assert rc >= 0, "lane emulation failed (%d: -2 = a child link led outside the node or triangle table)" % rc
        self.frame_index = first_frame + n_frames - 1
        return rc
