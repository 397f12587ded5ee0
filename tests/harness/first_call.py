"""What the first call of a fresh context costs on a scene of many triangles: rt_render_frames(1, N) straight after
rt_set_scene, then the same call again.  Arguments: subdivision levels (default 5), frames (default 8), scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
from raytracer_amd.scene import subdivided_suzanne_packed
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H, spp = 1920, 1080, 16
cam, sp, tr = subdivided_suzanne_packed(levels, float(sys.argv[3])) if len(sys.argv) > 3 else subdivided_suzanne_packed(levels)
warm = capi.Context(W, H)  # code objects loaded by another context, so that the first call below is not charged for them
warm.set_scene(sp, tr); warm.set_max_bounces(8); warm.set_uniforms(S.pack_uniforms(cam, 1, W, H, spp)); warm.render_frames(1, 2); warm.sync(); warm.close()
ctx = capi.Context(W, H)
ctx.set_max_bounces(8)
if 'RT_TILEORDER' in os.environ: ctx.set_tuning(14, int(os.environ['RT_TILEORDER']))
ctx.set_scene(sp, tr); ctx.sync()
ctx.set_uniforms(S.pack_uniforms(cam, 1, W, H, spp))
for call in range(3):
    t = time.time(); ctx.render_frames(1 + call * frames, frames); ctx.sync(); dt = time.time() - t
    print("triangles", len(tr), "call", call, "frames", frames, "ms/frame %.2f" % (dt * 1e3 / frames), "gpu ms %.2f" % ctx.last_render_ms(), flush=True)
