#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the HIP path tracer on
scene_suzanne.json at 1920x1080, 16 spp per frame (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W

A step is one frame: one pass of the path-trace + accumulate hot path over the
image at 16 samples per pixel.  K timed steps (default 16 = 256 spp, the
configuration the metric is quoted on).

N > 1 (launched by torch.distributed.run, one rank per GPU): the image rows are
dealt to the ranks in 8-row stripes; each rank renders K*N frames of its
stripes (so the work per GPU is that of the N = 1 run: weak scaling), no
collective on the data path, and one RCCL all-gather of the accumulator rows
at the end, inside the timed region.  The assembled image is bit-identical to
what one GPU renders for the same K*N frames.  ``--split frames`` deals the K*N
frames instead (every rank renders whole images and sums them; one RCCL
all-reduce of the sums at the end): same work per GPU, not bit-identical.

Rank 0 prints one JSON line.  ``roofline`` prices the path-trace kernel against
the FP32 vector peak (this path has no matrix work); ``cpu_baseline`` times the
CPU oracle (a plain-C restatement of the reference shader) on a bounded band
of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SCENE = "suzanne"
WIDTH, HEIGHT, SPP, MAX_BOUNCES = 1920, 1080, 16, 6
PEAK_FP32_TFLOPS = 157.3  # MI355X FP32 vector peak (packed FMA), chip-level spec
# algorithmic flop prices per test / per bounce ("Roofline accounting")
F_BOX, F_TRI, F_SPH, F_SHADE = 24, 52, 31, 160


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU baseline band (0 = auto)")
    ap.add_argument("--split", choices=["rows", "frames"], default="rows",
                    help="N > 1: deal image rows (exact, default) or frames (sums + all-reduce) to the ranks")
    return ap.parse_args()


def cpu_baseline(scene, threads):
    """Times the CPU oracle on a band of frame 1 (same scene, resolution, spp)."""
    from tests import oracle_lib as O

    def run(y0, y1):
        orc = O.OracleRenderer(scene, WIDTH, HEIGHT, max_bounces=MAX_BOUNCES, threads=threads, rect=(0, y0, WIDTH, y1))
        orc.set_samples_per_pixel(SPP)
        t = time.perf_counter()
        orc.render()
        return int(orc.counters[0]), time.perf_counter() - t

    # warm up (library load, thread start), calibrate on one row per thread (the oracle splits a band
    # into rows), then size the band for about 15 s
    run(540, 541)
    cal = max(2, min(threads, 64))
    rays, dt = run(540 - cal // 2, 540 - cal // 2 + cal)
    rows = int(max(cal, min(HEIGHT // 2, round(cal * 15.0 / max(dt, 1e-3)))))
    y0 = (HEIGHT - rows) // 2
    rays, dt = run(y0, y0 + rows)
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": "rows %d..%d of frame 1 (%d of %d pixels), %d spp, %d rays in %.1f s"
                  % (y0, y0 + rows, rows * WIDTH, WIDTH * HEIGHT, SPP, rays, dt),
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from raytracer_amd import capi
    from raytracer_amd import scene as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus must equal WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the renderer)")
    # Rehearsal on a one-GPU box: RT_BENCH_REHEARSE=1 puts every rank on GPU 0 and gathers over gloo
    # (RCCL refuses two ranks on one device).  The driver's real runs use one GPU per rank and RCCL.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    device = 0 if rehearse else local_rank
    torch.cuda.set_device(device)
    # RT_BENCH_FORCE_DIST=1 runs the collective path even with one rank (smoke test of the RCCL calls)
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1:  # forced collective path outside torch.distributed.run
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))

    from raytracer_amd import sharding

    scene = S.load_scene(S.scene_path(SCENE))
    spheres, tris = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
    by_frames = args.split == "frames" and use_dist
    if by_frames:
        ctx = capi.Context(WIDTH, HEIGHT, device=device)
    else:
        ctx = capi.Context(WIDTH, HEIGHT, device=device, stripe_rows=8, stripe_index=rank, stripe_count=world)
    ctx.set_max_bounces(MAX_BOUNCES)
    ctx.set_scene(spheres, tris)
    ctx.set_uniforms(S.pack_uniforms(scene.camera, 1, WIDTH, HEIGHT, SPP))

    frames_per_step = world  # weak scaling: every rank keeps the N = 1 amount of work
    K, W = args.steps, args.warmup

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    own = ctx.owned_rows()
    frames_per_launch = ctx.frames_per_launch()
    frames_this_rank = sharding.frames_of_rank(1, K * frames_per_step, rank, world)[1] if by_frames else K * frames_per_step
    local_rows = torch.empty((3, own, WIDTH), dtype=torch.float32, device="cuda") if use_dist else None

    # warmup (frames 1..W*frames_per_step); the timed run then restarts at frame 1
    if W > 0:
        if by_frames:
            ctx.render_frame_sums(*sharding.frames_of_rank(1, W * frames_per_step, rank, world))
        else:
            ctx.render_frames(1, W * frames_per_step)
    barrier()

    c_before = [int(x) for x in ctx.read_counters()]
    barrier()
    t0 = time.perf_counter()
    if by_frames:
        # every rank: whole-image sums of its share of the frames; all-reduce; rank 0 keeps the result
        ctx.clear_accum()
        ctx.render_frame_sums(*sharding.frames_of_rank(1, K * frames_per_step, rank, world))
        ctx.export_accum(local_rows.data_ptr())
        ctx.sync()
        total = sharding.all_reduce_sums(local_rows.cpu() if rehearse else local_rows)
        if rank == 0:
            total = total.cuda().contiguous()
            torch.cuda.synchronize()
            ctx.import_accum(total.data_ptr())
            ctx.finish_frame_sums(K * frames_per_step)
            ctx.sync()
    else:
        ctx.render_frames(1, K * frames_per_step)
    t_render = None
    if use_dist and not by_frames:
        # combine: every rank contributes its rows; rank 0 assembles the image
        ctx.export_owned_rows(local_rows.data_ptr())
        ctx.sync()
        t_render = time.perf_counter() - t0
        blocks = sharding.all_gather_rows(local_rows.cpu() if rehearse else local_rows, world,
                                          counts=[len(sharding.owned_rows(HEIGHT, 8, r, world)) for r in range(world)])
        if rank == 0:
            for src, block in enumerate(blocks):
                if src != 0:
                    block = block.cuda().contiguous()
                    torch.cuda.synchronize()
                    ctx.import_rows(block.data_ptr(), src)
            ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    if t_render is not None and os.environ.get("RT_BENCH_VERBOSE") == "1":
        print("rank %d: render %.1f ms, gather + assemble %.1f ms" % (rank, t_render * 1e3, (dt - t_render) * 1e3), file=sys.stderr)
    kernel_ms = ctx.last_render_ms()
    c_after = [int(x) for x in ctx.read_counters()]
    rays = c_after[13] - c_before[13]        # rays actually traversed
    ray_calls = c_after[0] - c_before[0]     # ray_all calls as the reference counts them
    if os.environ.get("RT_BENCH_VERBOSE") == "1":
        print("rank %d: %d rays traversed, kernel %.1f ms" % (rank, rays, kernel_ms), file=sys.stderr)

    if world > 1 and rank == 0 and os.environ.get("RT_BENCH_VERIFY") == "1" and by_frames:
        ref = capi.Context(WIDTH, HEIGHT, device=device)
        ref.set_max_bounces(MAX_BOUNCES)
        ref.set_scene(spheres, tris)
        ref.set_uniforms(S.pack_uniforms(scene.camera, 1, WIDTH, HEIGHT, SPP))
        ref.render_frames(1, K * frames_per_step)
        a, b = ref.read_accum(), ctx.read_accum()
        ref.close()
        d = np.abs(a - b)
        print("verify: frame split vs single-GPU recurrence: %d of %d values differ by more than 1e-3 (clamped pixels), "
              "median abs diff %.3g" % (int((d > 1e-3).sum()), d.size, float(np.median(d))), file=sys.stderr)
    elif world > 1 and rank == 0 and os.environ.get("RT_BENCH_VERIFY") == "1":
        # the assembled image must equal what one GPU renders for the same frames (after the same warmup)
        ref = capi.Context(WIDTH, HEIGHT, device=device)
        ref.set_max_bounces(MAX_BOUNCES)
        ref.set_scene(spheres, tris)
        ref.set_uniforms(S.pack_uniforms(scene.camera, 1, WIDTH, HEIGHT, SPP))
        if W > 0:
            ref.render_frames(1, W * frames_per_step)
        ref.render_frames(1, K * frames_per_step)
        same = bool(np.array_equal(ref.read_accum(), ctx.read_accum()))
        ref.close()
        print("verify: assembled image bit-identical to the single-GPU render: %s" % same, file=sys.stderr)
        if not same:
            raise SystemExit("verification failed")

    t = torch.tensor([dt, float(rays), kernel_ms, float(ray_calls)], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, kernel_ms, rays, ray_calls = float(tmax[0]), float(tmax[2]), int(tsum[1]), int(tsum[3])
    if rank != 0:
        dist.destroy_process_group()
        return

    # roofline: algorithmic flops of the timed launches / HIP-event time of those launches
    cctx = capi.Context(WIDTH, HEIGHT, device=device)
    cctx.set_max_bounces(MAX_BOUNCES)
    cctx.set_scene(spheres, tris)
    cctx.set_counting(True)
    cctx.set_uniforms(S.pack_uniforms(scene.camera, 1, WIDTH, HEIGHT, SPP))
    cctx.render_frame()
    cnt = [int(x) for x in cctx.read_counters()]
    cctx.close()
    c_rays, c_boxpairs, c_tris, c_sph, c_bounces = cnt[13], cnt[1], cnt[2], cnt[3], cnt[4]
    lane_util = None
    if cnt[5]:
        lane_util = {"inner": cnt[6] / cnt[5] / 64, "triangle": cnt[8] / max(cnt[7], 1) / 64,
                     "shade": cnt[10] / max(cnt[9], 1) / 64, "shadow_done": cnt[12] / max(cnt[11], 1) / 64}
    flops_per_ray = (2 * c_boxpairs * F_BOX + c_tris * F_TRI + c_sph * F_SPH + c_bounces * F_SHADE) / c_rays
    brute_flops_per_ray = F_SPH * len(spheres) + F_TRI * len(tris)
    frames = K * frames_per_step
    batch = frames_per_launch       # frames per kernel launch, as chosen by the library for this context
    launches = max(1, (frames_this_rank + batch - 1) // batch)
    rays_per_launch = rays / world / launches
    launch_ms = max(kernel_ms / launches, 1e-9)
    achieved = flops_per_ray * rays_per_launch / (launch_ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "Mrays/s",
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt * 1e3 / max(K, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "reference scene file scene_suzanne.json + models/suzanne_500_tris.obj (982 triangles, 1 sphere); no random inputs",
        "config": {
            "workload": "scene_suzanne.json 1920x1080, %d frames x %d spp, max bounces %d, BVH traversal%s"
                        % (K * frames_per_step, SPP, MAX_BOUNCES,
                           "" if world == 1 else
                           (", frames dealt to %d GPUs + RCCL all-reduce of the sums" % world if by_frames else
                            ", rows dealt to %d GPUs in 8-row stripes + RCCL all-gather" % world)),
            "frames": K * frames_per_step,
            "spp_per_frame": SPP,
            "ms_per_frame": dt * 1e3 / max(K * frames_per_step, 1),
            "rays": rays,
            "ray_all_calls_of_the_reference": ray_calls,
            "note": "value counts rays actually traversed; the camera ray of a pixel is traversed once per frame and its "
                    "hit reused by the frame's 16 samples, which the shader starts on the same ray, so the reference's own "
                    "counter would read ray_all_calls_of_the_reference",
        },
        "roofline": {
            "bound": "valu_fp32",
            "achieved": achieved,
            "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_TFLOPS,
            "traffic": traffic,
            "kernel": "render_persistent_kernel (path trace of up to %d frames per launch; the ordered accumulate kernel "
                      "that follows it is included in the timing and is < 0.1%% of it)" % batch,
            "launches": launches,
            "kernel_ms_per_launch": launch_ms,
            "flops_per_ray": flops_per_ray,
            "rays_per_launch": rays_per_launch,
            "per_ray": {"box_tests": 2 * c_boxpairs / c_rays, "triangle_tests": c_tris / c_rays,
                        "sphere_tests": c_sph / c_rays, "bounces": c_bounces / c_rays},
            "lane_utilisation": lane_util,
            "brute_force_equivalent": {"flops_per_ray": brute_flops_per_ray,
                                       "tflops": brute_flops_per_ray * rays_per_launch / (launch_ms * 1e-3) / 1e12},
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        out["cpu_baseline"] = cpu_baseline(scene, threads)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
