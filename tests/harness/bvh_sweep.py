"""Builder settings sweep (leaf size, SAH triangle cost) for scene_suzanne 1080p x 16 spp."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import capi, scene as S
W, H, spp, frames = 1920, 1080, 16, 8
name = sys.argv[1] if len(sys.argv) > 1 else "suzanne"
sc = S.load_scene(S.scene_path(name))
sp, tr = S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles)
ctx = capi.Context(W, H)
res = {}
for rnd in range(2):
    for leaf, cost in itertools.product([1, 2, 3, 4, 6, 8], [8, 12, 20, 40]):
        ctx.set_tuning(capi.RT_TUNE_MAX_LEAF, leaf); ctx.set_tuning(capi.RT_TUNE_TRI_COST_X10, cost)
        ctx.set_scene(sp, tr)
        ctx.set_uniforms(S.pack_uniforms(sc.camera, 1, W, H, spp))
        ctx.render_frames(1, frames); ctx.sync()
        res.setdefault((leaf, cost), []).append(ctx.last_render_ms() / frames)
        st = ctx.scene_stats()
        res[(leaf, cost, 'st')] = st
for k in sorted([k for k in res if len(k) == 2], key=lambda k: min(res[k])):
    print("max_leaf %d tri_cost %.1f: ms/frame %s  %s" % (k[0], k[1] / 10, " ".join("%.2f" % v for v in res[k]), res[k + ('st',)]))
