"""Subdivided-suzanne workload (968 * 4**levels model triangles) at 1080p, 8 bounces: scene set-up time and ms/frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
from raytracer_amd.scene import subdivided_suzanne_packed
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5   # negative: height field of (-levels)^2 quads instead
W, H, spp, frames = 1920, 1080, 16, int(sys.argv[2]) if len(sys.argv) > 2 else 4
if levels < 0:
    from tests.synth import heightfield_scene
    hf = heightfield_scene(-levels)
    cam, sp, tr = hf.camera, S.pack_spheres(hf.spheres), S.pack_triangles(hf.triangles)
else:
    cam, sp, tr = subdivided_suzanne_packed(levels, float(sys.argv[3])) if len(sys.argv) > 3 else subdivided_suzanne_packed(levels)
ctx = capi.Context(W, H)
ctx.set_max_bounces(8)
for k, name_ in enumerate(['SHADE', 'LEAF', 'BPC', 'BATCH', 'SDONE', 'REFILL']):
    if 'RT_' + name_ in os.environ: ctx.set_tuning(k, int(os.environ['RT_' + name_]))
if 'RT_SHADEWAIT' in os.environ: ctx.set_tuning(10, int(os.environ['RT_SHADEWAIT']))
if 'RT_MAXLEAF' in os.environ: ctx.set_tuning(6, int(os.environ['RT_MAXLEAF']))
if 'RT_TRICOST' in os.environ: ctx.set_tuning(7, int(os.environ['RT_TRICOST']))
t = time.time(); ctx.set_scene(sp, tr); ctx.sync(); t_scene = time.time() - t
ctx.set_uniforms(S.pack_uniforms(cam, 1, W, H, spp))
ctx.render_frames(1, 1); ctx.sync()
c0 = ctx.read_counters()[13]
t = time.time(); ctx.render_frames(2, frames); ctx.sync(); dt = time.time() - t
c1 = ctx.read_counters()[13]
print("levels", levels, "triangles", len(tr), "set_scene %.2f s" % t_scene, "ms/frame %.2f" % (dt * 1e3 / frames),
      "Mrays/s %.1f" % ((c1 - c0) / dt / 1e6), ctx.scene_stats())
