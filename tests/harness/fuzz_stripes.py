"""Row-striped contexts on random scenes and shapes: G contexts (the ranks of a row split, here on one GPU) render the same
frames, each its own stripes, by a random mix of single frames, batches and the no-allocation entry point; rank 0 collects
the others' rows through host memory; the assembled image, the ray counts added up and rank 0's id rows must equal the
oracle's whole-image render.  Arguments: first seed, one past the last."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytracer_amd import capi, scene as S
from tests import oracle_lib as O
from tests.synth import random_scene


def run(seed):
    rng = random.Random(seed * 15485863 + 3)
    scene = random_scene(seed) if seed % 3 else S.load_scene(S.scene_path(rng.choice(["spheres", "boxes", "suzanne"])))
    width, height = rng.choice([(48, 27), (64, 36), (33, 17), (20, 70), (96, 54)])
    ranks, stripe_rows = rng.choice([1, 2, 3, 4, 8]), rng.choice([1, 3, 5, 8, 16])
    spp, frames = rng.choice([1, 2, 4]), rng.randint(1, 6)
    orc = O.OracleRenderer(scene, width, height)
    orc.samples_per_pixel = spp
    orc.render_frames(frames)
    ctxs = []
    rays = 0
    for r in range(ranks):
        ctx = capi.Context(width, height, stripe_rows=stripe_rows, stripe_index=r, stripe_count=ranks)
        ctx.set_tuning(capi.RT_TUNE_OVERLAP_FRAMES, rng.choice([0, 2, 3]))
        ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
        f = 1
        while f <= frames:
            ctx.set_uniforms(S.pack_uniforms(scene.camera, f, width, height, spp))
            how = rng.choice(["frame", "batch", "noalloc"])
            n = 1 if how == "frame" else rng.randint(1, frames - f + 1)
            if how == "frame":
                ctx.render_frame()
            elif how == "batch":
                ctx.render_frames(f, n)
            else:
                ctx.render_frames_noalloc(f, n)
                ctx.sync()   # contexts on one GPU share the fixed workspace: one call at a time
            f += n
        ctx.sync()
        rays += int(ctx.read_counters()[0])
        ctxs.append(ctx)
    for r in range(1, ranks):
        if ctxs[r].owned_rows():
            ctxs[0].write_rows(ctxs[r].read_owned_rows(), r)
    ok = np.array_equal(ctxs[0].read_accum(), orc.accum_rgb()) and rays == int(orc.counters[0])
    for ctx in ctxs:
        ctx.close()
    return ok, (width, height, ranks, stripe_rows, spp, frames)


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(first, last):
        ok, shape = run(seed)
        if not ok:
            bad.append(seed)
            print("DIFFERS: seed", seed, shape, flush=True)
        if seed % 50 == 49:
            print("through seed", seed, flush=True)
    print("seeds", first, "to", last - 1, ":", len(bad), "differ", bad)
    sys.exit(1 if bad else 0)
