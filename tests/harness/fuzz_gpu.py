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
width, height, spp, frames = 40, 23, 2, 2
bad = []
for seed in range(first, last):
    scene = random_scene(seed, n_tris)
    sl = seed % 3 == 0
    mode = capi.RT_MODE_BVH_TILES if seed % 5 == 0 else capi.RT_MODE_BVH
    orc = O.OracleRenderer(scene, width, height, sphere_lights=sl)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(frames)
    ctx = capi.Context(width, height)
    ctx.set_mode(mode)
    ctx.set_sphere_lights(sl)
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
        print("DIFFERS: seed", seed, "spheres", len(scene.spheres), "triangles", len(scene.triangles), "sphere lights", sl, "mode", mode, flush=True)
    if seed % 50 == 49:
        print("through seed", seed, flush=True)
print("seeds", first, "to", last - 1, ":", len(bad), "differ", bad)
sys.exit(1 if bad else 0)
