"""Tuning sweep of the persistent kernel's scheduling knobs (one process, interleaved rounds)."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
name, W, H, spp, frames = "suzanne", 1920, 1080, 16, 16
sc = S.load_scene(S.scene_path(name))
ctx = capi.Context(W, H)
ctx.set_scene(S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, spp))
grid = list(itertools.product([1, 2, 3, 4], [8, 12, 16, 24], [1, 2, 4, 8], [1, 4, 8, 16]))
res = {g: [] for g in grid}
for rnd in range(2):
    for g in grid:
        ctx.set_tuning(0, g[0]); ctx.set_tuning(1, g[1]); ctx.set_tuning(4, g[2]); ctx.set_tuning(5, g[3])
        ctx.render_frames(1, frames); ctx.sync()
        res[g].append(ctx.last_render_ms() / frames)
for g in sorted(grid, key=lambda g: min(res[g])):
    print("shade %d leaf %2d sdone %2d refill %2d  ms/frame %s" % (g[0], g[1], g[2], g[3], " ".join("%.2f" % v for v in res[g])))
