"""ms/frame as a function of frames per launch (RT_TUNE_BATCH_FRAMES) for scene_suzanne 1080p x 16 spp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
W, H, spp = 1920, 1080, 16
sc = S.load_scene(S.scene_path("suzanne"))
ctx = capi.Context(W, H)
ctx.set_scene(S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, spp))
import os
for k, n_ in enumerate(['SHADE', 'LEAF', 'BPC', 'BATCH', 'SDONE', 'REFILL']):
    if 'RT_' + n_ in os.environ: ctx.set_tuning(k, int(os.environ['RT_' + n_]))
ctx.render_frames(1, 4); ctx.sync()
for batch in (1, 2, 4, 16, 64):
    ctx.set_tuning(capi.RT_TUNE_BATCH_FRAMES, batch)
    n = max(batch, 16)
    ctx.render_frames(1, n); ctx.sync()
    ms = ctx.last_render_ms()
    print("batch %2d: %d frames in %.1f ms = %.2f ms/frame, %.2f ms per launch" % (batch, n, ms, ms / n, ms / (n / batch)))
