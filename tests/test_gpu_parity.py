"""GPU parity: the HIP path through the C ABI against the CPU oracle on the
same scenes, resolutions, spp and frame indices.  Bar: bit-exact primary-hit
ids, bit-exact rgba16float frame, bit-exact f32 accumulator (which implies the
stated tolerance max-abs <= 1e-3 / PSNR >= 50 dB)."""
import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd import scene as S
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _f16_equal(a, b):
    """Half bit patterns equal, any NaN == any NaN."""
    a = a.view(np.float16)
    b = b.view(np.float16)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _packed(scene):
    """(camera, packed spheres, packed triangles) of a Scene or of such a tuple."""
    if isinstance(scene, tuple):
        return scene
    return scene.camera, S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)


def run_pair(scene, width, height, frames, spp, mode, max_bounces=6, tuning=()):
    orc = O.OracleRenderer(scene, width, height, max_bounces=max_bounces)
    orc.set_samples_per_pixel(spp)
    camera, spheres, triangles = _packed(scene)
    ctx = capi.Context(width, height)
    ctx.set_max_bounces(max_bounces)
    ctx.set_mode(mode)
    for key, value in tuning:
        ctx.set_tuning(key, value)
    ctx.set_scene(spheres, triangles)
    for f in range(1, frames + 1):
        orc.render()
        ctx.set_uniforms(S.pack_uniforms(camera, f, width, height, spp))
        ctx.render_frame()
        ids = ctx.read_primary_ids()
        assert np.array_equal(ids, orc.ids), "primary ids differ in frame %d" % f
        assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16), "f16 frame differs in frame %d" % f
    acc = ctx.read_accum()
    ref = orc.accum_rgb()
    assert np.array_equal(acc, ref), "accumulator differs: max abs %g" % np.abs(acc - ref).max()
    counters = ctx.read_counters()
    assert int(counters[0]) == int(orc.counters[0]), "ray count differs"
    ctx.close()


@pytest.mark.parametrize("mode", [capi.RT_MODE_BRUTE_FORCE, capi.RT_MODE_BVH_TILES, capi.RT_MODE_BVH])
@pytest.mark.parametrize("name,width,height,frames,spp", [
    ("spheres", 64, 36, 2, 16),
    ("spheres", 160, 90, 4, 1),
    ("boxes", 160, 90, 4, 2),
    ("suzanne", 100, 61, 2, 3),   # ragged: neither multiple of 8 nor of 32
])
def test_small_scenes_bit_exact(scenes, mode, name, width, height, frames, spp):
    run_pair(scenes[name], width, height, frames, spp, mode)


def test_config_a_spheres_640x360_16spp(scenes):
    # BASELINE config (a): scene_spheres 640x360, 1 frame x 16 spp, 6 bounces
    run_pair(scenes["spheres"], 640, 360, 1, 16, capi.RT_MODE_BVH)


def test_max_bounces_8(scenes):
    run_pair(scenes["boxes"], 96, 54, 2, 4, capi.RT_MODE_BVH, max_bounces=8)


@pytest.mark.parametrize("lds_state", [0, 1, 2])
def test_lane_state_placement_does_not_change_output(scenes, lds_state):
    """Registers only / shadow-ray state in LDS / per-pixel state in LDS too: three kernel instantiations."""
    run_pair(scenes["suzanne"], 100, 61, 2, 3, capi.RT_MODE_BVH, tuning=((capi.RT_TUNE_LDS_STATE, lds_state),))
    run_batched(scenes["boxes"], 96, 54, 3, 4, 2, ((capi.RT_TUNE_LDS_STATE, lds_state),))


def run_batched(scene, width, height, frames, spp, batch, tuning=(), stripes=None):
    """rt_render_frames (several frames per launch + ordered accumulate) against the oracle."""
    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(frames)
    ctx = capi.Context(width, height)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    ctx.set_tuning(capi.RT_TUNE_BATCH_FRAMES, batch)
    for key, value in tuning:
        ctx.set_tuning(key, value)
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
    ctx.render_frames(1, frames)
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    assert int(ctx.read_counters()[0]) == int(orc.counters[0])
    ctx.close()


@pytest.mark.parametrize("batch", [1, 2, 3, 16])
def test_batched_frames_bit_exact(scenes, batch):
    run_batched(scenes["suzanne"], 100, 61, 5, 2, batch)


@pytest.mark.parametrize("tuning", [
    ((capi.RT_TUNE_SHADE_NUM, 0), (capi.RT_TUNE_LEAF_LANES, 1)),
    ((capi.RT_TUNE_SHADE_NUM, 8), (capi.RT_TUNE_LEAF_LANES, 64)),
    ((capi.RT_TUNE_BLOCKS_PER_CU, 1),),
    ((capi.RT_TUNE_SHADE_WAIT, 0), (capi.RT_TUNE_REFILL_LANES, 64)),
    ((capi.RT_TUNE_SHADE_WAIT, 1), (capi.RT_TUNE_SHADOW_DONE_LANES, 64), (capi.RT_TUNE_REFILL_LANES, 1)),
    ((capi.RT_TUNE_TILE_ORDER, 0),),   # tiles handed out in image order instead of most expensive first
])
def test_scheduling_knobs_do_not_change_output(scenes, tuning):
    run_batched(scenes["boxes"], 96, 54, 3, 4, 2, tuning)


def test_triangle_test_counter_wraps_like_u32(scenes):
    scene = scenes["suzanne"]
    ctx = capi.Context(64, 36)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, 64, 36, 16))
    ctx.render_frame()
    rays = int(ctx.read_counters()[0])
    assert ctx.take_triangle_tests() == (rays * 982) % 2**32
    assert ctx.take_triangle_tests() == 0
    ctx.close()


def test_heightfield_8k_triangles_vs_oracle():
    """A mesh the reference scenes do not have: 8,206 triangles, deeper tree."""
    from tests.synth import heightfield_scene

    run_pair(heightfield_scene(64), 96, 54, 1, 2, capi.RT_MODE_BVH)


def test_heightfield_131k_triangles_bvh_vs_brute_force_on_gpu():
    """131,086 triangles: too slow for the CPU oracle, so the BVH kernels are compared with the GPU's own
    brute-force mode (the reference's ray_all loop), which the other tests tie to the oracle."""
    from tests.synth import heightfield_scene

    scene = heightfield_scene(256)
    sp, tr = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    width, height, spp = 64, 36, 2
    out = {}
    for mode in (capi.RT_MODE_BRUTE_FORCE, capi.RT_MODE_BVH, capi.RT_MODE_BVH_TILES):
        ctx = capi.Context(width, height)
        ctx.set_mode(mode)
        ctx.set_scene(sp, tr)
        if mode == capi.RT_MODE_BVH:
            assert ctx.scene_stats()["max_depth"] >= 16   # deep enough for stack entries beyond the 16 kept in LDS
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
        ctx.render_frames(1, 2)
        out[mode] = (ctx.read_primary_ids(), ctx.read_frame_f16(), ctx.read_accum(), int(ctx.read_counters()[0]))
        ctx.close()
    ref = out[capi.RT_MODE_BRUTE_FORCE]
    for mode in (capi.RT_MODE_BVH, capi.RT_MODE_BVH_TILES):
        assert np.array_equal(out[mode][0], ref[0])
        assert _f16_equal(out[mode][1], ref[1])
        assert np.array_equal(out[mode][2], ref[2])
        assert out[mode][3] == ref[3]


def test_subdivided_suzanne_62k_triangles_vs_oracle():
    """The model's triangles split three times (61,966 triangles, part of them already under the
    reference's determinant threshold), 8 bounces."""
    from tests.synth import subdivided_suzanne_packed

    run_pair(subdivided_suzanne_packed(3), 64, 36, 1, 2, capi.RT_MODE_BVH, max_bounces=8)


def test_subdivided_suzanne_1m_triangles_vs_oracle():
    """991,246 triangles (five splits), 8 bounces: the largest configuration, on an image the
    brute-force oracle finishes in tens of seconds."""
    from tests.synth import subdivided_suzanne_packed

    run_pair(subdivided_suzanne_packed(5), 32, 18, 1, 1, capi.RT_MODE_BVH, max_bounces=8)


def test_frame_split_sums_match_the_recurrence(scenes):
    """Two contexts stand in for two ranks: each sums its share of the frames, the sums are
    added on the device (what the all-reduce does) and finished; against the oracle's sequential
    accumulator the result differs by rounding only, except where the recurrence clamps."""
    import torch

    scene = scenes["boxes"]
    width, height, frames, spp, world = 96, 54, 5, 2, 2
    from raytracer_amd import sharding

    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(spp)
    exact = np.zeros((3, height, width), np.float64)
    clamped = np.zeros((height, width), bool)   # pixels whose running mean left [0, 1] at some frame (or is NaN)
    for f in range(1, frames + 1):
        before = orc.accum.copy()
        orc.render()
        cur = np.transpose(orc.frame_f16.view(np.float16).astype(np.float32)[:, :, :3], (2, 0, 1))
        exact += cur
        w = np.float32(1) / np.float32(f + 1)
        clamped |= ~((before * (np.float32(1) - w) + cur * w >= 0) & (before * (np.float32(1) - w) + cur * w <= 1)).all(axis=0)
    total = torch.zeros((3, height, width), dtype=torch.float32, device="cuda")
    ctxs = []
    for rank in range(world):
        ctx = capi.Context(width, height)
        ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
        ctx.set_tuning(capi.RT_TUNE_BATCH_FRAMES, 2)   # rank 0 has three frames: a batch of two and a single
        first, count, stride = sharding.frames_of_rank(1, frames, rank, world)
        ctx.clear_accum()
        ctx.render_frame_sums(first, count, stride)
        part = torch.empty_like(total)
        torch.cuda.synchronize()
        ctx.export_accum(part.data_ptr())
        ctx.sync()
        total += part
        torch.cuda.synchronize()   # the contexts write on their own streams: nothing of torch's may be in flight
        ctxs.append(ctx)
    sums = total.cpu().numpy()
    finite = np.isfinite(exact)
    assert np.array_equal(np.isfinite(sums), finite)
    assert np.abs(sums - exact)[finite].max() <= 1e-5 * max(1.0, np.abs(exact[finite]).max())
    ctxs[0].import_accum(total.data_ptr())
    ctxs[0].finish_frame_sums(frames)
    got = np.transpose(ctxs[0].read_accum(), (2, 0, 1))
    assert np.array_equal(got, sharding.finish_sums(sums, frames), equal_nan=True)
    diff = np.abs(got - orc.accum)
    assert diff[:, ~clamped].max() <= 2e-6
    for ctx in ctxs:
        ctx.close()


@pytest.mark.parametrize("name", ["empty", "spheres_only", "no_lights", "rotated_camera", "camera_down",
                                  "emissive_and_mirror_spheres", "duplicates_and_degenerates", "many_spheres", "shuffled_duplicates"])
def test_edge_scenes_bit_exact(name):
    """Empty inputs, no lights, rotated cameras, emissive / mirror spheres, coincident and degenerate
    triangles, a long sphere list: single frames (fused epilogue) and a batch, all modes for the first."""
    from tests.synth import edge_scenes

    scene = edge_scenes()[name]
    for mode in (capi.RT_MODE_BVH, capi.RT_MODE_BVH_TILES, capi.RT_MODE_BRUTE_FORCE):
        run_pair(scene, 72, 41, 2, 3, mode)
    run_batched(scene, 72, 41, 3, 2, 3)


@pytest.mark.parametrize("width,height", [(1, 1), (1, 9), (9, 1), (8, 8), (65, 3)])
def test_tiny_and_thin_images(scenes, width, height):
    run_pair(scenes["boxes"], width, height, 2, 2, capi.RT_MODE_BVH)
    run_batched(scenes["boxes"], width, height, 3, 2, 2)


def test_late_frame_indices_and_reset(scenes):
    """frameIndex 1000..1002 in one batch (seed offset and weights), then resetAccumulation's effect:
    frameIndex back to 1 with the planes kept (src/webgpu-renderer.ts:274-276)."""
    scene = scenes["spheres"]
    width, height, spp = 48, 27, 2
    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(spp)
    orc.frame_index = 999
    orc.render_frames(3)
    ctx = capi.Context(width, height)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1000, width, height, spp))
    ctx.render_frames(1000, 3)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    orc.reset_accumulation()
    orc.render_frames(2)
    ctx.render_frames(1, 2)
    assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    ctx.close()


def test_invalid_arguments_are_rejected(scenes):
    scene = scenes["spheres"]
    ctx = capi.Context(16, 9)
    with pytest.raises(capi.RtError):
        ctx.render_frame()                      # no scene yet
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    with pytest.raises(capi.RtError):
        ctx.render_frame()                      # no uniforms yet
    with pytest.raises(capi.RtError):
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, 32, 9, 1))   # resolution differs from the context's
    with pytest.raises(capi.RtError):
        ctx.set_max_bounces(256)
    with pytest.raises(capi.RtError):
        ctx.set_tuning(99, 1)
    with pytest.raises(capi.RtError):
        ctx.set_mode(7)
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, 16, 9, 1))
    ctx.render_frame()
    ctx.sync()
    ctx.close()
    with pytest.raises(capi.RtError):
        capi.Context(0, 9)
    with pytest.raises(capi.RtError):
        capi.Context(16, 9, stripe_index=2, stripe_count=2)
    with pytest.raises(capi.RtError):
        capi.Context(65536, 65536)              # more (frame, pixel) units than a launch can number


@pytest.mark.parametrize("overlap", [0, 1])
def test_back_to_back_single_frames(scenes, overlap):
    """rt_render_frame called six times without waiting in between: with overlap the traces of consecutive
    frames run concurrently on two side streams; the accumulate steps must still apply in frame order and
    the ids and rgba16float frame must be those of the last frame.  overlap 0 = the fused single-kernel path."""
    scene = scenes["suzanne"]
    width, height, spp, frames = 160, 90, 3, 6
    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(frames)
    ctx = capi.Context(width, height)
    ctx.set_tuning(capi.RT_TUNE_OVERLAP_FRAMES, overlap)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    for f in range(1, frames + 1):
        ctx.set_uniforms(S.pack_uniforms(scene.camera, f, width, height, spp))
        ctx.render_frame()
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    assert int(ctx.read_counters()[0]) == int(orc.counters[0])
    # a batch after single frames, then single frames again
    orc.render_frames(5)
    ctx.render_frames(frames + 1, 3)
    for f in range(frames + 4, frames + 6):
        ctx.set_uniforms(S.pack_uniforms(scene.camera, f, width, height, spp))
        ctx.render_frame()
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    ctx.close()


def test_card_deck_uses_the_stack_overflow_area():
    """6,014 triangles stacked so that camera rays leave 18 entries on the traversal stack
    (tests/test_core_cpu.py checks that depth on the emulated lanes): entries beyond the 12 that the
    large-scene kernel keeps in LDS go to global memory."""
    from tests.synth import card_deck_scene

    scene = card_deck_scene(3000)
    run_pair(scene, 40, 23, 2, 2, capi.RT_MODE_BVH)
    run_batched(scene, 64, 36, 3, 2, 3)


def test_python_renderer_mirror(scenes, tmp_path):
    """raytracer_amd.hip_renderer.HipRenderer: the reference class's sequencing (frameIndex incremented
    before the uniforms are written, spp clamp, reset keeps the planes, counter zeroed by the read,
    None on failure) over the native context, with an unfenced render loop."""
    from raytracer_amd.hip_renderer import HipRenderer

    assert HipRenderer.create(32, 18, str(tmp_path / "missing.json")) is None
    width, height = 80, 45
    r = HipRenderer.create(width, height, S.scene_path("spheres"))
    assert r is not None
    r.set_samples_per_pixel(100)
    assert r.samples_per_pixel == 16
    r.set_samples_per_pixel(0)
    assert r.samples_per_pixel == 1
    r.set_samples_per_pixel(3)
    r.start_render_loop(frames=5)
    orc = O.OracleRenderer(scenes["spheres"], width, height)
    orc.set_samples_per_pixel(3)
    orc.render_frames(5)
    assert r.frame_index == 5
    assert np.array_equal(r.read_accum(), orc.accum_rgb())
    assert np.array_equal(r.read_ids(), orc.ids)
    assert r.get_triangle_tests_since_last_check() == (int(orc.counters[0]) * 14) % 2**32
    assert r.get_triangle_tests_since_last_check() == 0
    r.reset_accumulation()
    orc.reset_accumulation()
    r.set_enabled(False)
    r.start_render_loop(frames=2)          # disabled: nothing is rendered
    assert r.frame_index == 0
    r.set_enabled(True)
    r.render_frames(2)
    r.render()
    orc.render_frames(3)
    assert np.array_equal(r.read_accum(), orc.accum_rgb())
    image, at = r.present(wait=True)         # the canvas image of src/accumulator.ts:119-136
    assert at == 3 and np.array_equal(image[:, :, :3], np.floor(orc.accum_rgb().astype(np.float64) * 255 + 0.5).astype(np.uint8))
    r.close()
    r.render()                               # no resources: silent no-op (src/webgpu-renderer.ts:238-246)


def test_scene_with_model_transforms_applied():
    """The opt-in loader extension that applies a model's scale / rotation / position (scene_suzanne's
    model at half size near the back wall): new geometry for the renderer, same parity bar."""
    scene = S.load_scene(S.scene_path("suzanne"), apply_model_transforms=True)
    run_pair(scene, 96, 54, 2, 3, capi.RT_MODE_BVH)


def test_five_thousand_single_frames(scenes):
    """More single-frame submissions than the library has per-launch work counters (4096), at the
    reference UI's 1 spp: the counter ring wraps, the two staging slots alternate 2500 times each, and
    the accumulator weight runs down to 1/5001."""
    scene = scenes["spheres"]
    width, height, frames = 64, 36, 5000
    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(1)
    orc.render_frames(frames)
    ctx = capi.Context(width, height)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    u = S.pack_uniforms(scene.camera, 1, width, height, 1)
    for f in range(1, frames + 1):
        u[12] = f
        ctx.set_uniforms(u)
        ctx.render_frame()
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    assert int(ctx.read_counters()[0]) == int(orc.counters[0])
    ctx.close()


@pytest.mark.parametrize("spp", [17, 4096, 4097])
def test_sample_counts_beyond_the_ui_clamp(scenes, spp):
    """The reference's UI clamps samples per pixel to 1..16 (src/webgpu-renderer.ts:219-221) but the shader loops
    over whatever the uniform says (raytracer_compute.wgsl:299).  Through the C ABI any count is accepted: a lane of
    the persistent kernel counts up to 4096 samples, a frame asking for more is rendered one thread per pixel."""
    scene = scenes["boxes"]
    width, height = 24, 10
    orc = O.OracleRenderer(scene, width, height)
    orc.samples_per_pixel = spp  # past set_samples_per_pixel's clamp
    orc.render_frames(2)
    ctx = capi.Context(width, height)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
    ctx.render_frames(1, 2)
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    assert int(ctx.read_counters()[0]) == int(orc.counters[0])
    ctx.close()


@pytest.mark.parametrize("mode", [capi.RT_MODE_BVH, capi.RT_MODE_BRUTE_FORCE])
def test_sphere_light_extension(mode):
    """rt_set_sphere_lights: emissive spheres sampled at every bounce (the shader's unused
    sample_light_spheres), bit-exact against the oracle built with the same option."""
    from tests.synth import edge_scenes

    for name in ("emissive_and_mirror_spheres", "many_spheres", "no_lights"):
        scene = edge_scenes()[name]
        width, height, spp = 96, 54, 3
        orc = O.OracleRenderer(scene, width, height, sphere_lights=True)
        orc.set_samples_per_pixel(spp)
        ctx = capi.Context(width, height)
        ctx.set_mode(mode)
        ctx.set_sphere_lights(True)
        ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
        orc.render_frames(3)
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
        ctx.render_frame()
        ctx.render_frames(2, 2)
        assert np.array_equal(ctx.read_primary_ids(), orc.ids)
        assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16)
        assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
        assert int(ctx.read_counters()[0]) == int(orc.counters[0])
        if mode == capi.RT_MODE_BVH:
            # these scenes are rendered by the persistent kernel (its instrumented build counts node-step rounds)
            ctx.set_counting(True)
            ctx.render_frames(4, 1)
            assert int(ctx.read_counters()[5]) > 0
        ctx.close()


@pytest.mark.parametrize("n,with_triangles", [(9, True), (60, True), (150, False), (5000, True)])
def test_spheres_as_leaves_of_the_tree(n, with_triangles):
    """Scenes of more than eight spheres: the spheres are leaves of the acceleration structure instead of a
    list every ray walks.  All three modes against the oracle (which walks the list, as the reference does);
    5000 spheres exceed the 4096 leaf positions of the 16-bit links, i.e. take the large-scene kernel."""
    from tests.synth import sphere_field_scene

    scene = sphere_field_scene(n, with_triangles)
    width, height, spp = (64, 36, 2) if n <= 150 else (40, 23, 1)
    for mode in (capi.RT_MODE_BVH, capi.RT_MODE_BVH_TILES, capi.RT_MODE_BRUTE_FORCE):
        run_pair(scene, width, height, 2, spp, mode)
    run_batched(scene, width, height, 3, spp, 2)


@pytest.mark.parametrize("distance,fov", [(1000.0, 0.4), (30000.0, 0.013)])
def test_far_camera_bvh_vs_brute_force_and_oracle(scenes, distance, fov):
    """A camera thousands of units from the geometry, zoomed in: the f32 slack of the triangle test
    and of the box test grows with the ray origin's magnitude, so the acceleration structure is
    rebuilt with wider boxes when the camera is placed (the scene was uploaded before the uniforms)."""
    import copy

    base = scenes["suzanne"]
    cam = copy.deepcopy(base.camera)
    cam["position"] = [0.3, 1.2, distance]
    cam["fov"] = fov
    scene = S.Scene(camera=cam, spheres=base.spheres, triangles=base.triangles)
    for mode in (capi.RT_MODE_BVH, capi.RT_MODE_BVH_TILES, capi.RT_MODE_BRUTE_FORCE):
        run_pair(scene, 96, 54, 2, 2, mode)
    run_batched(scene, 96, 54, 3, 2, 2)


def test_tree_deeper_than_the_accepted_stack_goes_to_the_tile_kernel(scenes):
    """RT_TUNE_MAX_STACK below the scene's stack need: the default mode must hand the frames to the
    one-thread-per-pixel kernel instead of failing, single frames and batches alike."""
    run_pair(scenes["suzanne"], 100, 61, 2, 3, capi.RT_MODE_BVH, tuning=((capi.RT_TUNE_MAX_STACK, 4),))
    run_batched(scenes["suzanne"], 100, 61, 3, 2, 2, tuning=((capi.RT_TUNE_MAX_STACK, 4),))


@pytest.mark.parametrize("coalesce", [0, 3, 16])
def test_single_frame_submissions_are_coalesced_without_changing_results(scenes, coalesce):
    """rt_render_frame in a loop that never waits: frames submitted while the GPU is busy are held back
    and launched as batches.  The camera moves and the sample count changes half way (held-back frames
    must be launched with the uniforms they were asked for), the frame index jumps once, a counter
    read and a batch call come in between; everything must equal the frame-by-frame oracle."""
    import copy

    scene = scenes["suzanne"]
    width, height = 160, 90
    cam2 = copy.deepcopy(scene.camera)
    cam2["position"] = [0.4, 2.6, 4.5]
    cam2["rotation"] = [-8.0, 6.0, 0.0]
    scene2 = S.Scene(camera=cam2, spheres=scene.spheres, triangles=scene.triangles)
    ctx = capi.Context(width, height)
    ctx.set_tuning(capi.RT_TUNE_COALESCE_FRAMES, coalesce)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    orc = O.OracleRenderer(scene, width, height)
    plan = [(scene, 2, f) for f in range(1, 24)] + [(scene2, 3, f) for f in range(24, 40)] + [(scene2, 3, f) for f in range(100, 108)]
    taken = 0
    for sc, spp, f in plan:
        ctx.set_uniforms(S.pack_uniforms(sc.camera, f, width, height, spp))
        ctx.render_frame()
        orc.scene, orc.frame_index = sc, f - 1
        orc.set_samples_per_pixel(spp)
        orc.render()
        if f == 11:
            taken += ctx.take_triangle_tests()
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    assert int(ctx.read_counters()[0]) == int(orc.counters[0])
    assert (taken + ctx.take_triangle_tests()) % 2**32 == (int(orc.counters[0]) * 982) % 2**32
    # held-back frames, then a batch call, then single frames again
    for f in (108, 109):
        ctx.set_uniforms(S.pack_uniforms(scene2.camera, f, width, height, 3))
        ctx.render_frame()
    ctx.render_frames(110, 3)
    ctx.set_uniforms(S.pack_uniforms(scene2.camera, 113, width, height, 3))
    ctx.render_frame()
    orc.frame_index = 107
    orc.render_frames(6)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    ctx.close()


@pytest.mark.parametrize("mode", [capi.RT_MODE_BVH, capi.RT_MODE_BRUTE_FORCE, capi.RT_MODE_BVH_TILES])
def test_debug_view_extension(scenes, mode):
    """rt_set_debug_view: while uniforms.debugEnabled is set, every triangle's normal marker
    (the shader's unused ray_triangle_normal_intersect) is one more candidate in ray_all; bit-exact
    against the oracle built with the same option, in every mode (such frames go to the brute-force
    kernel), single frames and a batch, switched off again in between."""
    scene = scenes["boxes"]
    width, height, spp = 96, 54, 2
    orc = O.OracleRenderer(scene, width, height, debug_view=True)
    orc.set_samples_per_pixel(spp)
    ctx = capi.Context(width, height)
    ctx.set_mode(mode)
    ctx.set_debug_view(True)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    for f, dbg in ((1, True), (2, True), (3, False), (4, True)):
        orc.debug_enabled = dbg
        orc.render()
        ctx.set_uniforms(S.pack_uniforms(scene.camera, f, width, height, spp, dbg))
        ctx.render_frame()
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    assert (orc.ids >= 1 + 34).any()
    orc.render_frames(2)
    ctx.render_frames(5, 2)
    assert np.array_equal(ctx.read_primary_ids(), orc.ids)
    assert _f16_equal(ctx.read_frame_f16(), orc.frame_f16)
    assert np.array_equal(ctx.read_accum(), orc.accum_rgb())
    assert int(ctx.read_counters()[0]) == int(orc.counters[0])
    ctx.close()


def test_host_staged_row_exchange(scenes):
    """rt_read_owned_rows / rt_write_rows / rt_rows_of_stripe: the row exchange through host memory
    that the Node addon uses, against the device-pointer path's result (the unstriped oracle image)."""
    scene = scenes["spheres"]
    width, height, spp, count = 72, 45, 2, 3
    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(2)
    ctxs = []
    for i in range(count):
        ctx = capi.Context(width, height, stripe_rows=8, stripe_index=i, stripe_count=count)
        ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
        ctx.render_frame()
        ctx.set_uniforms(S.pack_uniforms(scene.camera, 2, width, height, spp))
        ctx.render_frame()      # may still be held back: the read below must launch it first
        ctxs.append(ctx)
    from raytracer_amd import sharding
    for i in range(1, count):
        block = ctxs[i].read_owned_rows()
        assert block.shape[1] == ctxs[0].rows_of_stripe(i) == len(sharding.owned_rows(height, 8, i, count))
        ctxs[0].write_rows(block, i)
    assert np.array_equal(ctxs[0].read_accum(), orc.accum_rgb())
    with pytest.raises(capi.RtError):
        ctxs[0].rows_of_stripe(count)
    for ctx in ctxs:
        ctx.close()


def test_presented_image(scenes):
    """rt_present: the canvas image of the accumulator pass (RGBA8, value * 255 rounded, no gamma).  Waited
    for, it is the image of everything submitted; not waited for, it is some earlier frame's image, complete and
    labelled with that frame's index, and catches up once the GPU has."""
    scene = scenes["boxes"]
    width, height, spp, frames = 96, 54, 2, 6
    orc = O.OracleRenderer(scene, width, height)
    orc.set_samples_per_pixel(spp)
    want = {}
    for f in range(1, frames + 1):
        orc.render()
        img = np.full((height, width, 4), 255, np.uint8)
        img[:, :, :3] = np.floor(orc.accum_rgb().astype(np.float64) * 255 + 0.5).astype(np.uint8)
        want[f] = img
    ctx = capi.Context(width, height)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    seen = []
    for f in range(1, frames + 1):
        ctx.set_uniforms(S.pack_uniforms(scene.camera, f, width, height, spp))
        ctx.render_frame()
        img, at = ctx.present(wait=False)
        assert at <= f
        if at:
            assert np.array_equal(img, want[at]), "image labelled frame %d is not that frame's" % at
            seen.append(at)
    assert seen == sorted(seen)
    img, at = ctx.present(wait=True)
    assert at == frames and np.array_equal(img, want[frames])
    img, at = ctx.present(wait=False)          # nothing new: the same image again, no new conversion
    assert at == frames and np.array_equal(img, want[frames])
    ctx.render_frames(frames + 1, 2)
    orc.render_frames(2)
    img, at = ctx.present(wait=True)
    assert at == frames + 2
    assert np.array_equal(img[:, :, :3], np.floor(orc.accum_rgb().astype(np.float64) * 255 + 0.5).astype(np.uint8))
    ctx.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_scenes_bit_exact(seed):
    """Seeded random scenes (tests.synth.random_scene: random primitives, materials, lights and camera): the
    persistent kernel frame by frame, the one-thread-per-pixel kernel, and a batch, against the oracle."""
    from tests.synth import random_scene

    scene = random_scene(seed)
    run_pair(scene, 48, 27, 2, 3, capi.RT_MODE_BVH)
    run_pair(scene, 48, 27, 1, 2, capi.RT_MODE_BVH_TILES)
    run_batched(scene, 48, 27, 3, 4, 2)


@pytest.mark.parametrize("seed,n_tris", [(1000, 4500), (1001, 9000), (1002, 20000)])
def test_random_large_scenes_bit_exact(seed, n_tris):
    """The same with thousands of random triangles: the kernels for scenes of more than 4096 triangles (32-bit
    links, stack overflow area in global memory, a leaf's triangles in one step)."""
    from tests.synth import random_scene

    scene = random_scene(seed, n_tris)
    run_pair(scene, 32, 18, 2, 2, capi.RT_MODE_BVH)
    run_batched(scene, 32, 18, 3, 2, 2)


@pytest.mark.parametrize("seed", range(16))
def test_random_call_sequences(seed):
    """Random sequences of calls on one context (tests/harness/fuzz_sequences.py): unfenced single-frame submissions,
    batches, the entry point that allocates nothing, moving cameras, changing sample counts, frame-index jumps, reads
    and presents in between, with the oracle driven the same way."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness", "fuzz_sequences.py")
    spec = importlib.util.spec_from_file_location("fuzz_sequences", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    ok, size, log = module.run_sequence(seed)
    assert ok, "sequence %d differs: %s %s" % (seed, size, log)

