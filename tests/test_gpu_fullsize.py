"""BASELINE configurations at their full image sizes (1920x1080 and 3840x2160, 16 spp).

The brute-force CPU oracle cannot finish a whole frame of these in test time, so each size is
pinned two ways:
  (i)  the default path (BVH traversal, persistent waves) against the GPU's own brute-force mode
       (the reference's ray_all loop over every primitive, which the small-size tests tie to the
       oracle) on the whole frame: ids, rgba16float frame, accumulator and ray count bit-equal;
  (ii) against the CPU oracle on a 16-row band through the image centre: bit-equal.
A full-HD frame has 32,400 8x8 tiles in a different wave schedule than the small images, unit
numbers up to 2^27 per launch and the automatic frames-per-launch choice; (i) is also the
empirical check that the padded boxes of the acceleration structure never cull a triangle that
the exact test accepts.
"""
import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd import scene as S
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _f16_equal(a, b):
    a = a.view(np.float16)
    b = b.view(np.float16)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("name,width,height,batched", [
    ("boxes", 1920, 1080, 3),      # BASELINE configs[1]
    ("suzanne", 1920, 1080, 3),    # BASELINE configs[2]
    ("suzanne", 3840, 2160, 2),    # BASELINE configs[3] (one GPU renders all rows here)
])
def test_full_size_bvh_vs_brute_force_and_oracle_band(scenes, name, width, height, batched):
    scene = scenes[name]
    spp = 16
    spheres, tris = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    y0 = height // 2 - 8
    band = slice(y0, y0 + 16)
    orc = O.OracleRenderer(scene, width, height, rect=(0, y0, width, y0 + 16))
    orc.set_samples_per_pixel(spp)

    bvh = capi.Context(width, height)
    bvh.set_scene(spheres, tris)
    brute = capi.Context(width, height)
    brute.set_mode(capi.RT_MODE_BRUTE_FORCE)
    brute.set_scene(spheres, tris)

    def compare(stage):
        ids, f16, acc = bvh.read_primary_ids(), bvh.read_frame_f16(), bvh.read_accum()
        assert np.array_equal(ids, brute.read_primary_ids()), "%s: ids differ from brute force" % stage
        assert _f16_equal(f16, brute.read_frame_f16()), "%s: f16 frame differs from brute force" % stage
        assert np.array_equal(acc, brute.read_accum()), "%s: accumulator differs from brute force" % stage
        assert int(bvh.read_counters()[0]) == int(brute.read_counters()[0]), "%s: ray counts differ" % stage
        assert np.array_equal(ids[band], orc.ids[band]), "%s: ids differ from the oracle band" % stage
        assert _f16_equal(f16[band], orc.frame_f16[band]), "%s: f16 differs from the oracle band" % stage
        assert np.array_equal(acc[band], orc.accum_rgb()[band]), "%s: accumulator differs from the oracle band" % stage

    # one frame through rt_render_frame (the reference's render loop)
    u = S.pack_uniforms(scene.camera, 1, width, height, spp)
    for ctx in (bvh, brute):
        ctx.set_uniforms(u)
        ctx.render_frame()
    orc.render()
    compare("frame 1")
    # then a batch through rt_render_frames
    for ctx in (bvh, brute):
        ctx.render_frames(2, batched)
    orc.render_frames(batched)
    compare("frames 2..%d" % (1 + batched))
    bvh.close()
    brute.close()


@pytest.mark.parametrize("which", ["heightfield_131k", "suzanne_split_5x_991k"])
def test_full_size_large_scenes_two_traversals_agree(which):
    """BASELINE config (e) (991,246 triangles, 8 bounces) and the 131,086-triangle height field at
    1920x1080 x 16 spp: the brute-force kernels cannot finish these, so the two independent BVH
    traversals are compared on the whole frame: the persistent kernel (4-wide tree, per-lane state
    machine, LDS stack with its overflow area) against the one-thread-per-pixel kernel (binary tree,
    the shader's loop nest).  The small-image tests tie each of them to the oracle on these scenes."""
    if which == "heightfield_131k":
        from tests.synth import heightfield_scene

        sc = heightfield_scene(256)
        camera, spheres, tris = sc.camera, S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles)
    else:
        camera, spheres, tris = S.subdivided_suzanne_packed(5)
    width, height, spp = 1920, 1080, 16
    out = {}
    for mode in (capi.RT_MODE_BVH, capi.RT_MODE_BVH_TILES):
        ctx = capi.Context(width, height)
        ctx.set_mode(mode)
        ctx.set_max_bounces(8)
        ctx.set_scene(spheres, tris)
        ctx.set_uniforms(S.pack_uniforms(camera, 1, width, height, spp))
        ctx.render_frame()
        ctx.render_frames(2, 2)
        out[mode] = (ctx.read_primary_ids(), ctx.read_frame_f16(), ctx.read_accum(), int(ctx.read_counters()[0]))
        ctx.close()
    a, b = out[capi.RT_MODE_BVH], out[capi.RT_MODE_BVH_TILES]
    assert np.array_equal(a[0], b[0])
    assert _f16_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2])
    assert a[3] == b[3]
