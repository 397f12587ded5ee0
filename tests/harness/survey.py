"""Timing survey over the BASELINE configurations that fit one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
from tests.synth import heightfield_scene
from raytracer_amd.scene import subdivided_suzanne_packed
def run(label, sc, W, H, frames, spp, bounces=6):
    ctx = capi.Context(W, H); ctx.set_max_bounces(bounces)
    cam, sp, tr = sc if isinstance(sc, tuple) else (sc.camera, S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
    t = time.time(); ctx.set_scene(sp, tr); tb = time.time() - t
    ctx.set_uniforms(S.pack_uniforms(cam, 1, W, H, spp))
    ctx.render_frames(1, 2); ctx.sync()
    c0 = ctx.read_counters()
    ctx.render_frames(1, frames); ctx.sync(); ms = ctx.last_render_ms()
    c1 = ctx.read_counters()
    print("%-28s %dx%d %d frames x %d spp: %.2f ms/frame, %.2f G rays/s traversed (%.2f G ray_all calls/s), scene upload+build %.0f ms, %s"
          % (label, W, H, frames, spp, ms / frames, (c1[13] - c0[13]) / ms / 1e6, (c1[0] - c0[0]) / ms / 1e6, tb * 1e3, ctx.scene_stats()))
    ctx.close()
L = lambda n: S.load_scene(S.scene_path(n))
run("(a) spheres", L("spheres"), 640, 360, 16, 16)
run("(b) boxes 64 spp", L("boxes"), 1920, 1080, 4, 16)
run("(c) suzanne 256 spp", L("suzanne"), 1920, 1080, 16, 16)
run("(c') suzanne 256 x 1 spp", L("suzanne"), 1920, 1080, 256, 1)
run("(b') boxes 64 x 1 spp", L("boxes"), 1920, 1080, 64, 1)
run("(d) suzanne 4K 1 GPU", L("suzanne"), 3840, 2160, 16, 16)
run("heightfield 131k tris, 8 b", heightfield_scene(256), 1920, 1080, 4, 16, 8)
run("(e) suzanne split 5x, 991k", subdivided_suzanne_packed(5), 1920, 1080, 8, 16, 8)
