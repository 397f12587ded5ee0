"""Scenes of many spheres at 1080p x 16 spp: spheres as leaves of the tree against the per-ray list.
   python tests/harness/sphere_bench.py n_spheres [sphere_lights]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
from tests.synth import sphere_field_scene
n = int(sys.argv[1]); sl = len(sys.argv) > 2
sc = sphere_field_scene(n)
W, H, spp, frames = 1920, 1080, 16, 8
for linear in (1 << 30, 8):
    ctx = capi.Context(W, H)
    ctx.set_tuning(capi.RT_TUNE_LINEAR_SPHERES, linear)
    if sl: ctx.set_sphere_lights(True)
    ctx.set_scene(S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
    ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, spp))
    ctx.render_frames(1, 2); ctx.sync()
    c0 = int(ctx.read_counters()[13]); t = time.time()
    ctx.render_frames(3, frames); ctx.sync(); dt = time.time() - t
    print("%d spheres, %s%s: %.2f ms/frame, %.0f Mrays/s" % (n, "list" if linear > 8 else "tree", ", sphere lights" if sl else "", dt * 1e3 / frames, (int(ctx.read_counters()[13]) - c0) / dt / 1e6))
    ctx.close()
