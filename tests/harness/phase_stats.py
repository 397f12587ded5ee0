"""Per-phase lane utilisation of the persistent kernel (instrumented build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
name = sys.argv[1] if len(sys.argv) > 1 else "suzanne"
W, H, spp, frames = 1920, 1080, int(os.environ.get("PS_SPP", "16")), int(os.environ.get("PS_FRAMES", "16"))
if name.startswith("hf"):      # hf256: height field of 256 x 256 quads
    from tests.synth import heightfield_scene
    sc = heightfield_scene(int(name[2:]))
    packed = (S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
elif name.startswith("sub"):   # sub5: suzanne with its model split 5 times
    from tests.synth import subdivided_suzanne_packed
    lv, _, scale = name[3:].partition("x")   # sub5x32: the same with every coordinate scaled x32
    cam, sp, tr = subdivided_suzanne_packed(int(lv), float(scale)) if scale else subdivided_suzanne_packed(int(lv))
    sc = S.Scene(camera=cam)
    packed = (sp, tr)
else:
    sc = S.load_scene(S.scene_path(name))
    packed = (S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles))
ctx = capi.Context(W, H)
if "PS_BOUNCES" in os.environ: ctx.set_max_bounces(int(os.environ["PS_BOUNCES"]))
for k, n in enumerate(["SHADE", "LEAF", "BPC", "BATCH", "SDONE", "REFILL"]):
    if "RT_" + n in os.environ: ctx.set_tuning(k, int(os.environ["RT_" + n]))
ctx.set_scene(*packed)
ctx.set_counting(True)
ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, spp))
ctx.render_frames(1, frames); ctx.sync()
c = [int(x) for x in ctx.read_counters()]
rays = c[13]
print({k: v for k, v in os.environ.items() if k.startswith("RT_")}, "ms/frame %.2f" % (ctx.last_render_ms() / frames))
print("per ray: nodes %.2f tris %.2f spheres %.2f bounces %.3f" % (c[1] / rays, c[2] / rays, c[3] / rays, c[4] / rays))
for nm, it, ln in (("inner", c[5], c[6]), ("tri", c[7], c[8]), ("shade", c[9], c[10]), ("sdone", c[11], c[12])):
    print("%-6s wave-iterations %d  (%.2f per ray-of-wave)  active lanes/iter %.1f (%.0f%%)" % (nm, it, it * 64 / rays, ln / max(it, 1), 100 * ln / max(it, 1) / 64))

import ctypes as C, numpy as np
out = np.zeros(16, np.uint64)
fn = ctx.lib.rt_debug_phase_cycles; fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
assert fn(ctx.handle, out.ctypes.data_as(C.c_void_p)) == 0
tot = float(out[4])
print("wave-cycle shares: inner %.1f%% tri %.1f%% sdone %.1f%% shade %.1f%% other %.1f%%" % tuple(
    [100 * float(out[i]) / tot for i in range(4)] + [100 * (tot - float(out[:4].sum())) / tot]))
it = float(out[8])
if it:
    print("lanes per state, averaged over traversal turns: inner %.1f leaf %.1f shadow-done %.1f ready-to-shade %.1f idle %.1f dead %.1f" % tuple(float(out[i]) / it for i in range(9, 15)))
