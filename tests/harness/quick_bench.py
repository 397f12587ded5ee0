import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from raytracer_amd import capi, scene as S
name, W, H, frames, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
mode = int(sys.argv[6]) if len(sys.argv) > 6 else 0
sc = S.load_scene(S.scene_path(name))
ctx = capi.Context(W, H)
ctx.set_mode(mode)
for k, name_ in enumerate(['SHADE', 'LEAF', 'BPC', 'BATCH', 'SDONE', 'REFILL', 'MAXLEAF', 'TRICOST', 'LDS', 'OVERLAP']):
    if 'RT_' + name_ in os.environ: ctx.set_tuning(k, int(os.environ['RT_' + name_]))
if 'RT_SHADEWAIT' in os.environ: ctx.set_tuning(10, int(os.environ['RT_SHADEWAIT']))
if 'RT_COALESCE' in os.environ: ctx.set_tuning(12, int(os.environ['RT_COALESCE']))
if 'RT_TILEORDER' in os.environ: ctx.set_tuning(14, int(os.environ['RT_TILEORDER']))
if 'RT_MIGRATE' in os.environ: ctx.set_tuning(13, int(os.environ['RT_MIGRATE']))
ctx.set_scene(S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, spp))
ctx.render_frames(1, 1); ctx.sync()
c0 = ctx.read_counters()[13]
t = time.time()
if os.environ.get('QB_SINGLE'):
    for f in range(frames):
        ctx.set_uniforms(S.pack_uniforms(sc.camera, 2 + f, W, H, spp)); ctx.render_frame()
else:
    ctx.render_frames(2, frames)
ctx.sync(); dt = time.time() - t
c1 = ctx.read_counters()[13]
print(name, W, H, frames, spp, 'mode', mode, {k: v for k, v in os.environ.items() if k.startswith('RT_')}, 'ms/frame %.3f' % (dt * 1e3 / frames), 'gpu ms %.3f' % ctx.last_render_ms(), 'Mrays/s %.1f' % ((c1 - c0) / dt / 1e6), ctx.scene_stats())
