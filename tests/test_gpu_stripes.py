"""GPU parity of the row-striped contexts (rt_create with stripe_count > 1), which the
multi-GPU row split is made of: several contexts on one GPU stand in for the ranks.

Every kernel that maps local rows to image rows is exercised: the persistent kernel in its
three launch shapes (overlapped single frames, fused single frames, batched frames), the
ordered accumulate kernel, the one-thread-per-pixel kernels (BVH and brute force), and the
export / import of owned rows.  Bar: bit-exact against the CPU oracle's unstriped render.
"""
import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd import scene as S
from raytracer_amd import sharding
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _f16_equal(a, b):
    a = a.view(np.float16)
    b = b.view(np.float16)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


PATHS = ["single_overlapped", "single_fused", "batched", "tiles", "brute_force"]


def _render(ctx, path, camera, width, height, frames, spp):
    if path in ("tiles", "brute_force"):
        ctx.set_mode(capi.RT_MODE_BVH_TILES if path == "tiles" else capi.RT_MODE_BRUTE_FORCE)
    if path == "batched":
        ctx.set_tuning(capi.RT_TUNE_BATCH_FRAMES, 2)     # three frames: a launch of two and a single
        ctx.set_uniforms(S.pack_uniforms(camera, 1, width, height, spp))
        ctx.render_frames(1, frames)
        return
    ctx.set_tuning(capi.RT_TUNE_OVERLAP_FRAMES, 0 if path == "single_fused" else 1)
    for f in range(1, frames + 1):
        ctx.set_uniforms(S.pack_uniforms(camera, f, width, height, spp))
        ctx.render_frame()


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("width,height,stripe_rows,count", [
    (100, 61, 8, 2),    # height not a multiple of the stripe: the last stripe is cut
    (96, 90, 8, 3),
    (80, 54, 8, 8),     # 7 stripes for 8 ranks: rank 7 owns nothing
    (100, 61, 5, 3),    # stripes that do not coincide with the 8x8 pixel tiles
])
def test_striped_contexts_assemble_to_the_unstriped_image(scenes, path, width, height, stripe_rows, count):
    import torch

    scene = scenes["suzanne"]
    frames, spp = 3, 2
    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(frames)
    ref_acc, ref_ids, ref_f16 = orc.accum_rgb(), orc.ids, orc.frame_f16

    spheres, tris = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    ctxs, rays = [], 0
    for index in range(count):
        ctx = capi.Context(width, height, stripe_rows=stripe_rows, stripe_index=index, stripe_count=count)
        ctx.set_scene(spheres, tris)
        _render(ctx, path, scene.camera, width, height, frames, spp)
        ctxs.append(ctx)
    seen = np.zeros(height, bool)
    blocks = []
    for index, ctx in enumerate(ctxs):
        own = sharding.owned_rows(height, stripe_rows, index, count)
        assert ctx.owned_rows() == len(own)
        assert not seen[own].any()
        seen[own] = True
        other = np.setdiff1d(np.arange(height), own)
        acc, ids, f16 = ctx.read_accum(), ctx.read_primary_ids(), ctx.read_frame_f16()
        assert np.array_equal(acc[own], ref_acc[own]), "owned accumulator rows differ (stripe %d)" % index
        assert np.array_equal(ids[own], ref_ids[own]), "owned id rows differ (stripe %d)" % index
        assert _f16_equal(f16[own], ref_f16[own]), "owned f16 rows differ (stripe %d)" % index
        # rows a context does not own are never touched: zero planes, id -2, zero frame
        assert not acc[other].any() and (ids[other] == -2).all() and not f16[other].any()
        rays += int(ctx.read_counters()[0])
        # the dense block of owned rows, as the all-gather carries it
        block = torch.full((3, max(len(own), 1), width), -7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.export_owned_rows(block.data_ptr())
        ctx.sync()
        got = block.cpu().numpy()[:, : len(own)]
        assert np.array_equal(got, np.transpose(ref_acc[own], (2, 0, 1)))
        blocks.append(block)
    assert seen.all()
    assert rays == int(orc.counters[0]), "the stripes' ray counts do not add up to the unstriped count"
    # rank 0 assembles the image from the other ranks' blocks
    for index in range(1, count):
        ctxs[0].import_rows(blocks[index].data_ptr(), index)
    assert np.array_equal(ctxs[0].read_accum(), ref_acc)
    # host-side assembly of the same blocks (what a gloo gather does)
    host = sharding.assemble([b.cpu().numpy()[:, : len(sharding.owned_rows(height, stripe_rows, i, count))]
                              for i, b in enumerate(blocks)], height, width, stripe_rows)
    assert np.array_equal(np.transpose(host, (1, 2, 0)), ref_acc)
    for ctx in ctxs:
        ctx.close()


def test_striped_frame_sums(scenes):
    """Stripes and frame sums together (a row split inside a frame split is not used by the bench,
    but both index maps meet in the accumulate kernel)."""
    scene = scenes["boxes"]
    width, height, frames, spp, count = 72, 41, 3, 2, 2
    spheres, tris = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    full = capi.Context(width, height)
    full.set_scene(spheres, tris)
    full.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
    full.clear_accum()
    full.render_frame_sums(1, frames, 1)
    ref = full.read_accum()
    full.close()
    for index in range(count):
        ctx = capi.Context(width, height, stripe_rows=8, stripe_index=index, stripe_count=count)
        ctx.set_scene(spheres, tris)
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
        ctx.set_tuning(capi.RT_TUNE_BATCH_FRAMES, 2)
        ctx.clear_accum()
        ctx.render_frame_sums(1, frames, 1)
        own = sharding.owned_rows(height, 8, index, count)
        assert np.array_equal(ctx.read_accum()[own], ref[own], equal_nan=True)
        ctx.close()


def test_import_of_a_gathered_buffer(scenes):
    """rt_import_gathered_rows: the padded all-gather result (one block per stripe index, each padded to
    the tallest) scattered by one call, as bench.py's rank 0 does."""
    import torch

    scene = scenes["boxes"]
    width, height, count, spp = 100, 61, 3, 2      # 61 rows in 8-row stripes: blocks of 24, 21 and 16 rows
    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(2)
    spheres, tris = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    counts = [len(sharding.owned_rows(height, 8, i, count)) for i in range(count)]
    tallest = max(counts)
    gathered = torch.full((count, 3, tallest, width), 9.0, dtype=torch.float32, device="cuda")
    ctxs = []
    for i in range(count):
        ctx = capi.Context(width, height, stripe_rows=8, stripe_index=i, stripe_count=count)
        ctx.set_scene(spheres, tris)
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
        ctx.render_frames(1, 2)
        gathered[i, :, : counts[i]] = torch.from_numpy(ctx.read_owned_rows()).cuda()
        ctxs.append(ctx)
    torch.cuda.synchronize()
    ctxs[1].import_gathered_rows(gathered.data_ptr(), tallest, 1)     # assembled on the middle rank, its own block skipped
    assert np.array_equal(ctxs[1].read_accum(), orc.accum_rgb())
    with pytest.raises(capi.RtError):
        ctxs[1].import_gathered_rows(gathered.data_ptr(), tallest - 1, 1)     # block 0 has `tallest` rows
    for ctx in ctxs:
        ctx.close()
