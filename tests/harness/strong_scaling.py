"""Strong-scaling proxy on one GPU.

A fixed job (K frames of a W x H image at 16 spp) is split over G ranks by rows, in 8-row stripes dealt
round-robin, exactly as the multi-GPU row split does.  On a G-GPU node the ranks run side by side, each
on a GPU of its own, and the job takes as long as the slowest rank.  Here every rank's share is rendered
alone on the one GPU of the box, one after the other, and timed:

    T_1   = time of the unsplit job
    T_G   = max over ranks r of the time of rank r's share
    efficiency(G) = T_1 / (G * T_G)

which is the definition of parallel efficiency for a fixed total amount of work.  The gather at the end of
a real run (24.9 MB at 1080p, 99.5 MB at 4K over xGMI) is not part of the proxy; a one-rank RCCL rehearsal
puts it at about 1 ms.

    python tests/harness/strong_scaling.py [--configs c,d] [--ranks 2,4,8] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raytracer_amd import capi, scene as S  # noqa: E402

SPP = 16
JOBS = {  # BASELINE configurations: scene, width, height, frames
    "a": ("spheres", 640, 360, 1),
    "b": ("boxes", 1920, 1080, 4),
    "c": ("suzanne", 1920, 1080, 16),
    "d": ("suzanne", 3840, 2160, 64),
}


def timed_share(scene, width, height, frames, index, count, repeats, tuning):
    ctx = capi.Context(width, height, stripe_rows=8, stripe_index=index, stripe_count=count)
    for key, value in tuning:
        ctx.set_tuning(key, value)
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, width, height, SPP))
    ctx.set_scene(S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles))
    ctx.render_frames(1, frames)  # warm-up: buffers, code objects
    ctx.sync()
    best_wall, best_gpu = 1e30, 1e30
    rays = 0
    for _ in range(repeats):
        c0 = int(ctx.read_counters()[13])
        t = time.perf_counter()
        ctx.render_frames(1, frames)
        ctx.sync()
        wall = (time.perf_counter() - t) * 1e3
        rays = int(ctx.read_counters()[13]) - c0
        best_wall, best_gpu = min(best_wall, wall), min(best_gpu, ctx.last_render_ms())
    out = dict(rows=ctx.owned_rows(), wall_ms=best_wall, gpu_ms=best_gpu, rays=rays, frames_per_launch=ctx.frames_per_launch())
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c,d")
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--tune", default="", help="key=value,... passed to rt_set_tuning (numeric keys)")
    args = ap.parse_args()
    tuning = [tuple(int(x) for x in kv.split("=")) for kv in args.tune.split(",") if kv]
    result = {"spp_per_frame": SPP, "stripe_rows": 8, "definition": "efficiency(G) = T_1 / (G * max over ranks of T_rank), every "
              "rank's share timed alone on one MI355X (GPU time by HIP events; wall time beside it)", "jobs": {}}
    for name in args.configs.split(","):
        scene_name, width, height, frames = JOBS[name]
        scene = S.load_scene(S.scene_path(scene_name))
        one = timed_share(scene, width, height, frames, 0, 1, args.repeats, tuning)
        job = {"scene": scene_name, "width": width, "height": height, "frames": frames, "T1": one, "split": {}}
        print("config %s: T1 = %.2f ms (gpu), %.2f ms (wall), %d rays" % (name, one["gpu_ms"], one["wall_ms"], one["rays"]), flush=True)
        for g in [int(x) for x in args.ranks.split(",")]:
            shares = [timed_share(scene, width, height, frames, r, g, args.repeats, tuning) for r in range(g)]
            t_gpu = max(s["gpu_ms"] for s in shares)
            t_wall = max(s["wall_ms"] for s in shares)
            mean_gpu = sum(s["gpu_ms"] for s in shares) / g
            job["split"][str(g)] = {
                "T_G_gpu_ms": t_gpu, "T_G_wall_ms": t_wall, "mean_rank_gpu_ms": mean_gpu,
                "efficiency_gpu": one["gpu_ms"] / (g * t_gpu), "efficiency_wall": one["wall_ms"] / (g * t_wall),
                "rays_min_max": [min(s["rays"] for s in shares), max(s["rays"] for s in shares)],
                "imbalance_slowest_over_mean": t_gpu / mean_gpu,
                "fixed_cost_ms_estimate": (g * mean_gpu - one["gpu_ms"]) / max(g - 1, 1),
                "ranks": shares,
            }
            print("  G=%d: T_G = %.2f ms (gpu) %.2f (wall)  efficiency %.3f (gpu) %.3f (wall)  slowest/mean %.3f  fixed cost per launch ~%.2f ms"
                  % (g, t_gpu, t_wall, one["gpu_ms"] / (g * t_gpu), one["wall_ms"] / (g * t_wall), t_gpu / mean_gpu,
                     job["split"][str(g)]["fixed_cost_ms_estimate"]), flush=True)
        result["jobs"][name] = job
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
