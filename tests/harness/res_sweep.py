"""Single-frame launch time vs resolution (how long one pixel takes when the GPU is not full)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
sc = S.load_scene(S.scene_path("suzanne"))
for (W, H) in [(64, 36), (240, 135), (480, 270), (960, 540), (1920, 1080), (3840, 2160)]:
    ctx = capi.Context(W, H)
    ctx.set_scene(S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
    ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, 16))
    ctx.set_tuning(capi.RT_TUNE_BATCH_FRAMES, 1)
    ctx.render_frames(1, 2); ctx.sync()
    ctx.render_frames(1, 8); ctx.sync()
    ms = ctx.last_render_ms() / 8
    print("%4dx%-4d %8d px: %.3f ms per single-frame launch, %.2f px per lane at 327680 lanes" % (W, H, W * H, ms, W * H / 327680))
    ctx.close()
