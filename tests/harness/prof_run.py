"""Small fixed workload for rocprofv3: N frames of scene_suzanne 1920x1080 x 16 spp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
name = sys.argv[2] if len(sys.argv) > 2 else "suzanne"
W, H, spp = 1920, 1080, 16
sc = S.load_scene(S.scene_path(name))
ctx = capi.Context(W, H)
ctx.set_scene(S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
import os
for k, name_ in enumerate(["SHADE", "LEAF", "BPC", "BATCH", "SDONE", "REFILL"]):
    if "RT_" + name_ in os.environ: ctx.set_tuning(k, int(os.environ["RT_" + name_]))
if "RT_MODE" in os.environ: ctx.set_mode(int(os.environ["RT_MODE"]))
ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, spp))
ctx.render_frames(1, frames)
ctx.sync()
print("ms", ctx.last_render_ms(), "rays", ctx.read_counters()[0])
