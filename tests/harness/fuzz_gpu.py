"""Seeded random scenes (tests.synth.random_scene) through the C ABI against the oracle, many of them: every third with
the sphere-light extension, every fifth through the one-thread-per-pixel kernel.  Arguments: first seed, one past the last, optionally the triangle count of every scene.
Prints the seeds that differ; exit status 1 if any does."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytracer_amd import capi, scene as S
from tests import oracle_lib as O
from tests.synth import random_scene

first, last = int(sys.argv[1]), int(sys.argv[2])
n_tris = int(sys.argv[3]) if len(sys.argv) > 3 else -1
import random
bad = []
for seed in range(first, last):
    scene = random_scene(seed, n_tris)
    rng = random.Random(seed * 7919 + 13)
    # image size, samples, bounces, scheduling and launch shape are drawn per seed too
    width, height = rng.choice([(40, 23), (1, 1), (8, 8), (65, 3), (9, 70), (100, 61), (64, 36)])
    spp, frames, bounces = rng.choice([1, 2, 3, 5, 16]), rng.choice([1, 2, 3, 5]), rng.choice([1, 2, 6, 8])
    sl = seed % 3 == 0
    mode = capi.RT_MODE_BVH_TILES if seed % 5 == 0 else capi.RT_MODE_BVH
    tuning = []
    if rng.random() < 0.5:
        tuning = [(capi.RT_TUNE_SHADE_NUM, rng.choice([0, 1, 2, 8])), (capi.RT_TUNE_LEAF_LANES, rng.choice([1, 12, 64])),
                  (capi.RT_TUNE_SHADOW_DONE_LANES, rng.choice([1, 2, 64])), (capi.RT_TUNE_REFILL_LANES, rng.choice([1, 4, 64])),
                  (capi.RT_TUNE_SHADE_WAIT, rng.choice([0, 1, 12])), (capi.RT_TUNE_BATCH_FRAMES, rng.choice([0, 1, 2])),
                  (capi.RT_TUNE_OVERLAP_FRAMES, rng.choice([0, 2, 3])), (capi.RT_TUNE_TILE_ORDER, rng.choice([0, 1])),
                  (capi.RT_TUNE_LDS_STATE, rng.choice([0, 1, 2])), (capi.RT_TUNE_LINEAR_SPHERES, rng.choice([0, 2, 8, 100]))]
    orc = O.OracleRenderer(scene, width, height, max_bounces=bounces, sphere_lights=sl)
    orc.samples_per_pixel = spp
    orc.render_frames(frames)
    ctx = capi.Context(width, height)
    ctx.set_max_bounces(bounces)
    ctx.set_mode(mode)
    ctx.set_sphere_lights(sl)
    for key, value in tuning:
        ctx.set_tuning(key, value)
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, spp))
    if seed % 2:
        ctx.render_frames(1, frames)
    else:
        for f in range(1, frames + 1):
            ctx.set_uniforms(S.pack_uniforms(scene.camera, f, width, height, spp))
            ctx.render_frame()
    same = (np.array_equal(ctx.read_primary_ids(), orc.ids) and np.array_equal(ctx.read_accum(), orc.accum_rgb())
            and int(ctx.read_counters()[0]) == int(orc.counters[0]))
    ctx.close()
    if not same:
        bad.append(seed)
        print("DIFFERS: seed", seed, "spheres", len(scene.spheres), "triangles", len(scene.triangles), "sphere lights", sl, "mode", mode,
              (width, height, spp, frames, bounces), tuning, flush=True)
    if seed % 50 == 49:
        print("through seed", seed, flush=True)
print("seeds", first, "to", last - 1, ":", len(bad), "differ", bad)
sys.exit(1 if bad else 0)
