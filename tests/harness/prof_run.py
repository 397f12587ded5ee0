"""Small fixed workload for rocprofv3: N frames at 1920x1080 x 16 spp of one of
   suzanne | boxes | spheres (reference scenes, 6 bounces), hf256 (131,086-triangle height field, 8 bounces),
   x5 (suzanne split 5 times, 991,246 triangles, 8 bounces), x5s (the same with the scene scaled x32: mesh visible).
usage: prof_run.py [frames] [scene] [width height]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
name = sys.argv[2] if len(sys.argv) > 2 else "suzanne"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
spp, bounces = 16, 6
if name == "hf256":
    from tests.synth import heightfield_scene
    sc = heightfield_scene(256)
    cam, sp, tr, bounces = sc.camera, S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles), 8
elif name in ("x5", "x5s"):
    cam, sp, tr = S.subdivided_suzanne_packed(5, 32.0 if name == "x5s" else 1.0)
    bounces = 8
else:
    sc = S.load_scene(S.scene_path(name))
    cam, sp, tr = sc.camera, S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles)
ctx = capi.Context(W, H)
ctx.set_max_bounces(bounces)
ctx.set_uniforms(S.pack_uniforms(cam, 1, W, H, spp))
ctx.set_scene(sp, tr)
for k, name_ in enumerate(["SHADE", "LEAF", "BPC", "BATCH", "SDONE", "REFILL"]):
    if "RT_" + name_ in os.environ: ctx.set_tuning(k, int(os.environ["RT_" + name_]))
if "RT_MODE" in os.environ: ctx.set_mode(int(os.environ["RT_MODE"]))
ctx.render_frames(1, frames)
ctx.sync()
c = ctx.read_counters()
print("ms", ctx.last_render_ms(), "ray_all calls", c[0], "rays traversed", c[13], "frames", frames, "scene", name, W, H)
