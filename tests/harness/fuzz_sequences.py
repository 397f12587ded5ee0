"""Random sequences of calls on one context against the oracle driven the same way: single-frame submissions that
never wait (held back and launched as batches by the library), batches, the entry point that allocates nothing, moving
cameras, changing sample counts, jumps of the frame index, reads and presents in between.  Arguments: first seed, one
past the last.  Exit status 1 if any sequence ends with a different image, id plane or ray count."""
import copy, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytracer_amd import capi, scene as S
from tests import oracle_lib as O
from tests.synth import random_scene

def run_sequence(seed):
    """One random sequence; (True, log of operations) if everything the library returned equals the oracle's."""
    rng = random.Random(seed * 104729 + 7)
    scene = random_scene(seed) if seed % 4 else S.load_scene(S.scene_path(rng.choice(["spheres", "boxes", "suzanne"])))
    width, height = rng.choice([(48, 27), (64, 36), (33, 17), (120, 40)])
    cams = [scene.camera]
    for _ in range(2):
        cam = copy.deepcopy(scene.camera)
        cam["position"] = [cam["position"][k] + rng.uniform(-0.4, 0.4) for k in range(3)]
        cam["rotation"] = [cam["rotation"][0] + rng.uniform(-5, 5), cam["rotation"][1] + rng.uniform(-5, 5), 0.0]
        cams.append(cam)
    orc = O.OracleRenderer(scene, width, height)
    ctx = capi.Context(width, height)
    ctx.set_tuning(capi.RT_TUNE_COALESCE_FRAMES, rng.choice([0, 2, 5, 16]))
    ctx.set_tuning(capi.RT_TUNE_OVERLAP_FRAMES, rng.choice([0, 2, 3]))
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    cam, spp, frame, shown = cams[0], rng.choice([1, 2, 4]), 0, 0   # shown: the last frame rendered (frame may have jumped past it)
    images = {}
    log = []

    def oracle_frames(f0, n):
        for f in range(f0, f0 + n):
            orc.scene = S.Scene(camera=cam, spheres=scene.spheres, triangles=scene.triangles)
            orc.frame_index = f - 1
            orc.samples_per_pixel = spp
            orc.render()
            img = np.full((height, width, 4), 255, np.uint8)
            img[:, :, :3] = np.floor(orc.accum_rgb().astype(np.float64) * 255 + 0.5).astype(np.uint8)
            images[f] = img

    ok = True
    for step in range(rng.randint(5, 40)):
        op = rng.choice(["frame"] * 8 + ["batch", "batch", "noalloc", "camera", "spp", "jump", "read", "present", "present_wait", "sync"])
        log.append(op)
        if op == "frame":
            frame += 1
            ctx.set_uniforms(S.pack_uniforms(cam, frame, width, height, spp))
            ctx.render_frame()
            oracle_frames(frame, 1)
            shown = frame
        elif op in ("batch", "noalloc"):
            n = rng.randint(1, 5)
            ctx.set_uniforms(S.pack_uniforms(cam, frame + 1, width, height, spp))
            (ctx.render_frames if op == "batch" else ctx.render_frames_noalloc)(frame + 1, n)
            oracle_frames(frame + 1, n)
            frame += n
            shown = frame
        elif op == "camera":
            cam = rng.choice(cams)
        elif op == "spp":
            spp = rng.choice([1, 2, 3, 4])
        elif op == "jump":
            frame += rng.choice([1, 7, 100])
        elif op == "read":
            rng.choice([ctx.read_counters, ctx.take_triangle_tests, ctx.read_primary_ids, ctx.last_render_ms if shown else ctx.sync])()
        elif op in ("present", "present_wait") and shown:
            img, at = ctx.present(wait=op == "present_wait")
            if op == "present_wait" and at != shown:
                ok = False
                print("seed", seed, "present(wait) labelled", at, "after frame", shown)
            if at and not np.array_equal(img, images[at]):
                ok = False
                print("seed", seed, "presented image of frame", at, "differs")
        elif op == "sync":
            ctx.sync()
    if shown:
        same = (np.array_equal(ctx.read_primary_ids(), orc.ids), np.array_equal(ctx.read_accum(), orc.accum_rgb()),
                int(ctx.read_counters()[0]) == int(orc.counters[0]))
        if not all(same):
            ok = False
            print("seed", seed, "ids / accumulator / ray count equal:", same)
    ctx.close()
    return ok, (width, height), log


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(first, last):
        ok, size, log = run_sequence(seed)
        if not ok:
            bad.append(seed)
            print("DIFFERS: seed", seed, size, log, flush=True)
        if seed % 50 == 49:
            print("through seed", seed, flush=True)
    print("seeds", first, "to", last - 1, ":", len(bad), "differ", bad)
    sys.exit(1 if bad else 0)
