"""The two independent restatements of the reference shaders -- plain C
(oracle/wgsl_oracle.c) and numpy float32 scalars (oracle/wgsl_numpy.py) -- must
agree bit for bit.  The reference has no runnable form here, so this is the
strongest pin the oracle can get; see the header of oracle/wgsl_oracle.h."""
import os
import sys

import numpy as np
import pytest

from raytracer_amd import scene as S
from tests import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import wgsl_numpy as N  # noqa: E402


def test_sincos_agree():
    for x in np.linspace(-7, 7, 1501).astype(np.float32):
        s, c = O.sincos(float(x))
        ns, nc = N.sincos(x)
        assert (np.float32(s), np.float32(c)) == (ns, nc)


def test_primitives_agree(scenes):
    sc = scenes["spheres"]
    spheres, tris = S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles)
    rng = np.random.default_rng(7)
    for _ in range(300):
        o = rng.uniform([-2.4, 0.1, -4.9], [2.4, 4.9, -0.1]).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        d = (d / np.float32(np.sqrt(np.dot(d, d)))).astype(np.float32)
        a = O.ray_all(o, d, spheres, tris)
        b = N.ray_all(N.Scene(spheres, tris), o, d)
        assert a.id == b.id and np.float32(a.t) == b.t
        if a.id >= 0:
            assert np.array_equal(np.array(a.normal[:], np.float32), b.normal)


@pytest.mark.parametrize("name,width,height,frames,spp", [
    ("spheres", 24, 14, 2, 2),
    ("boxes", 16, 9, 1, 3),
])
def test_frames_agree(scenes, name, width, height, frames, spp):
    _frames_agree(scenes[name], width, height, frames, spp)


@pytest.mark.parametrize("name", ["rotated_camera", "camera_down", "emissive_and_mirror_spheres", "many_spheres", "empty"])
def test_edge_scene_frames_agree(name):
    """Camera rotation, emissive and mirror spheres, many spheres: paths the reference's scene files never take."""
    from tests.synth import edge_scenes

    _frames_agree(edge_scenes()[name], 16, 9, 1, 2)


@pytest.mark.parametrize("name", ["emissive_and_mirror_spheres", "many_spheres"])
def test_sphere_light_extension_agrees(name):
    """sample_light_spheres (raytracer_compute.wgsl:109-171) and its opt-in use, in both restatements."""
    from tests.synth import edge_scenes

    _frames_agree(edge_scenes()[name], 16, 9, 1, 2, sphere_lights=True)


def _frames_agree(sc, width, height, frames, spp, sphere_lights=False):
    spheres, tris = S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles)
    orc = O.OracleRenderer(sc, width, height, sphere_lights=sphere_lights)
    orc.set_samples_per_pixel(spp)
    accum = np.zeros((3, height, width), np.float32)
    rays = 0
    for f in range(1, frames + 1):
        orc.render()
        frame, ids, r = N.render_frame(S.pack_uniforms(sc.camera, f, width, height, spp), spheres, tris, accum,
                                       sphere_lights=sphere_lights)
        rays += r
        assert np.array_equal(ids, orc.ids)
        assert np.array_equal(frame, orc.frame_f16)
    assert np.array_equal(accum, orc.accum)
    assert rays == int(orc.counters[0])


def test_debug_view_extension_c_vs_numpy(scenes):
    """The opt-in debug view (ray_triangle_normal_intersect markers in ray_all while debugEnabled is set):
    both restatements agree bit for bit, the markers show up, and the uniform alone changes nothing."""
    scene = scenes["boxes"]
    width, height, spp = 24, 14, 2
    sp, tr = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    orc = O.OracleRenderer(scene, width, height, debug_view=True)
    orc.debug_enabled = True
    orc.set_samples_per_pixel(spp)
    orc.render()
    acc = np.zeros((3, height, width), np.float32)
    frame, ids, rays = N.render_frame(S.pack_uniforms(scene.camera, 1, width, height, spp, True), sp, tr, acc, debug_view=True)
    assert np.array_equal(ids, orc.ids)
    assert np.array_equal(frame, orc.frame_f16)
    assert np.array_equal(acc, orc.accum)
    assert rays == int(orc.counters[0])
    assert (orc.ids >= len(sp) + len(tr)).any(), "no marker visible: the test scene does not exercise the extension"
    # extension off: the uniform is ignored, as the reference's shader ignores it
    plain = O.OracleRenderer(scene, width, height)
    plain.set_samples_per_pixel(spp)
    plain.render()
    ignored = O.OracleRenderer(scene, width, height)
    ignored.debug_enabled = True
    ignored.set_samples_per_pixel(spp)
    ignored.render()
    assert np.array_equal(plain.accum, ignored.accum) and np.array_equal(plain.ids, ignored.ids)
    assert not np.array_equal(plain.ids, orc.ids)
