"""Time line of one instrumented launch: when the waves find the work queues empty and when they end.
   python tests/harness/tail_trace.py scene W H frames spp [migrate]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytracer_amd import capi, scene as S
name, W, H, frames, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sc = S.load_scene(S.scene_path(name))
ctx = capi.Context(W, H)
if len(sys.argv) > 6: ctx.set_tuning(capi.RT_TUNE_MIGRATE_LANES, int(sys.argv[6]))
ctx.set_scene(S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, spp))
ctx.set_counting(True)
ctx.render_frames(1, frames); ctx.sync()
ctx.render_frames(1, frames); ctx.sync()
ms = ctx.last_render_ms()
t = ctx.wave_trace().astype(np.float64)
t0 = t[:, 0].min()
start, exh, end, live = (t[:, 0] - t0) / 1e5, (t[:, 1] - t0) / 1e5, (t[:, 2] - t0) / 1e5, t[:, 3]
has = t[:, 1] > 0
print("%s %dx%d %d frames %d spp: kernel %.2f ms (instrumented), %d waves" % (name, W, H, frames, spp, ms, len(t)))
print("start  : min %.3f max %.3f ms" % (start.min(), start.max()))
pc = [0, 1, 10, 50, 90, 99, 100]
print("queues empty seen at (ms), percentiles %s: %s" % (pc, np.round(np.percentile(exh[has], pc), 3)))
print("live lanes then: mean %.1f" % live[has].mean())
print("wave end (ms), percentiles %s: %s" % (pc, np.round(np.percentile(end, pc), 3)))
print("end - queues empty per wave (ms), percentiles: %s" % np.round(np.percentile((end - exh)[has], pc), 3))
edges = np.linspace(0, end.max(), 25)
alive = [(end > e).sum() for e in edges]
print("waves still running at t:", " ".join("%.1f:%d" % (e, a) for e, a in zip(edges, alive)))
