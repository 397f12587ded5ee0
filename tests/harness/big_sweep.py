"""Scheduling-threshold sweep on a large scene (one process, one scene set-up): big_sweep.py <levels or -n>"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
from tests.synth import subdivided_suzanne_packed, heightfield_scene
levels = int(sys.argv[1])
if levels < 0:
    hf = heightfield_scene(-levels)
    cam, sp, tr = hf.camera, S.pack_spheres(hf.spheres), S.pack_triangles(hf.triangles)
else:
    cam, sp, tr = subdivided_suzanne_packed(levels)
W, H, spp, frames = 1920, 1080, 16, 4
ctx = capi.Context(W, H)
ctx.set_max_bounces(8)
ctx.set_scene(sp, tr)
ctx.set_uniforms(S.pack_uniforms(cam, 1, W, H, spp))
grid = list(itertools.product([2, 3, 4], [8, 12, 16], [2, 4, 8], [8, 12, 16]))
res = {}
for g in grid:
    for key, v in zip((0, 1, 4, 10), g):
        ctx.set_tuning(key, v)
    ctx.render_frames(1, frames); ctx.sync()
    res[g] = ctx.last_render_ms() / frames
for g in sorted(grid, key=lambda g: res[g])[:12]:
    print("shade %d leaf %2d sdone %d wait %2d  ms/frame %.2f" % (g + (res[g],)))
print("default (2, 12, 2, 12): %.2f" % res[(2, 12, 2, 12)])
