"""rt_render_frames_noalloc (fixed device workspace, no hipMalloc) against rt_render_frames:
bit-equal ids, rgba16float frame, accumulator and ray count, on two calls in a row that share
the GPU's workspace, the second with a smaller image and another scene; then the limits."""
import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd import scene as S

pytestmark = pytest.mark.gpu


def _f16_equal(a, b):
    a = a.view(np.float16)
    b = b.view(np.float16)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _pair(scene, width, height, spp):
    spheres, tris = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    out = []
    for _ in range(2):
        ctx = capi.Context(width, height)
        ctx.set_scene(spheres, tris)
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
        out.append(ctx)
    return out


def _same(a, b):
    assert np.array_equal(a.read_primary_ids(), b.read_primary_ids())
    assert _f16_equal(a.read_frame_f16(), b.read_frame_f16())
    assert np.array_equal(a.read_accum(), b.read_accum())
    assert int(a.read_counters()[0]) == int(b.read_counters()[0])


def test_noalloc_matches_render_frames_twice_in_a_row(scenes):
    import torch

    big, big_ref = _pair(scenes["suzanne"], 200, 120, 3)
    small, small_ref = _pair(scenes["boxes"], 72, 41, 2)
    big_ref.render_frames(1, 5)
    small_ref.render_frames(1, 4)
    big_ref.sync()
    small_ref.sync()

    # first call: frames 1..3 in launches of two (the second launch reuses counter and colour area)
    big.set_tuning(capi.RT_TUNE_BATCH_FRAMES, 2)
    big.render_frames_noalloc(1, 3)
    big.sync()
    # second call, smaller input, other context and scene, same GPU workspace with the first call's leftovers in it
    small.render_frames_noalloc(1, 4)
    small.sync()
    _same(small, small_ref)
    # and back to the first context: two more frames on top of the three
    free_before = torch.cuda.mem_get_info()[0]
    big.render_frames_noalloc(4, 2)
    big.sync()
    free_after = torch.cuda.mem_get_info()[0]
    _same(big, big_ref)
    assert free_after == free_before, "device memory in use changed across rt_render_frames_noalloc"
    for ctx in (big, big_ref, small, small_ref):
        ctx.close()


def test_noalloc_limits(scenes):
    scene = scenes["spheres"]
    spheres, tris = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    # 4096 x 2160 = 8,847,360 owned pixels: more than the workspace's 8,388,608
    ctx = capi.Context(4096, 2160)
    ctx.set_scene(spheres, tris)
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, 4096, 2160, 1))
    with pytest.raises(capi.RtError, match="workspace"):
        ctx.render_frames_noalloc(1, 1)
    ctx.close()
    # the same image split in two stripes fits
    ctx = capi.Context(4096, 2160, stripe_rows=8, stripe_index=1, stripe_count=2)
    ctx.set_scene(spheres, tris)
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, 4096, 2160, 1))
    ctx.render_frames_noalloc(1, 1)
    ref = capi.Context(4096, 2160, stripe_rows=8, stripe_index=1, stripe_count=2)
    ref.set_scene(spheres, tris)
    ref.set_uniforms(S.pack_uniforms(scene.camera, 1, 4096, 2160, 1))
    ref.render_frames(1, 1)
    _same(ctx, ref)
    ctx.close()
    ref.close()
