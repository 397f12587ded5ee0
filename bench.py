#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the HIP path tracer.

    python bench.py --gpus N --steps K --warmup W [--config c]

A step is one frame: one pass of the path-trace + accumulate hot path over the
image at 16 samples per pixel.  The default configuration is the one the metric
is quoted on, BASELINE.json configs[2]: scene_suzanne.json at 1920x1080, K = 16
timed frames = 256 spp.  ``--config`` selects another BASELINE configuration:

    a  scene_spheres 640x360                      (configs[0])
    b  scene_boxes 1920x1080                      (configs[1])
    c  scene_suzanne 1920x1080                    (configs[2], default)
    d  scene_suzanne 3840x2160, rows split        (configs[3])
    e  suzanne split 5x (991,246 triangles), 8 bounces, 1920x1080, frames split   (configs[4])
    e_visible  the same mesh with the whole scene scaled x32, so that its triangles pass the
       shader's determinant threshold and are actually seen (secondary line to e)

N > 1 (launched by torch.distributed.run, one rank per GPU): by default the image
rows are dealt to the ranks in 8-row stripes; each rank renders K*N frames of
its stripes (so the work per GPU is that of the N = 1 run: weak scaling), no
collective on the data path, and one RCCL all-gather of the accumulator rows
at the end, inside the timed region.  The assembled image is bit-identical to
what one GPU renders for the same K*N frames.  ``--split frames`` (the default
of config e) deals the K*N frames instead (every rank renders whole images and
sums them; one RCCL all-reduce of the sums at the end): same work per GPU, not
bit-identical.

Rank 0 prints one JSON line.  ``roofline`` prices the path-trace kernel against
the FP32 vector peak (this path has no matrix work); the per-ray test counts and
the lane utilisation behind it come from an instrumented launch of the same
shape as the timed one (same frames per launch).  ``cpu_baseline`` times the CPU
oracle (a plain-C restatement of the reference shader) on a bounded rectangle of
the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SPP = 16
PEAK_FP32_TFLOPS = 157.3  # MI355X FP32 vector peak (packed FMA), chip-level spec
# algorithmic flop prices: one 4-wide node visit (four slab tests: 24 FMAs), one Moeller-Trumbore test, one sphere
# test, one bounce (light sample, BRDF, MIS weights, next direction), counted as the shader's own arithmetic
F_BOX, F_TRI, F_SPH, F_SHADE = 24, 52, 31, 160

CONFIGS = {
    # name: (scene, width, height, max bounces, default split, what BASELINE.json calls it)
    "a": ("spheres", 640, 360, 6, "rows", "scene_spheres.json 640x360"),
    "b": ("boxes", 1920, 1080, 6, "rows", "scene_boxes.json 1920x1080"),
    "c": ("suzanne", 1920, 1080, 6, "rows", "scene_suzanne.json 1920x1080"),
    "d": ("suzanne", 3840, 2160, 6, "rows", "scene_suzanne.json 3840x2160"),
    "e": ("suzanne_x5", 1920, 1080, 8, "frames", "suzanne model split 5x (991,246 triangles) 1920x1080"),
    "e_visible": ("suzanne_x5_scaled", 1920, 1080, 8, "frames",
                  "suzanne model split 5x (991,246 triangles), scene scaled x32 (mesh visible) 1920x1080"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every rank keeps the N = 1 amount of work (K*N frames in total); strong = the job is "
                         "fixed (K frames in total) and each rank gets 1/N of it")
    ap.add_argument("--split", choices=["rows", "frames"], default=None,
                    help="N > 1: deal image rows (exact) or frames (sums + all-reduce) to the ranks; default by config")
    return ap.parse_args()


def load_config_scene(name):
    """(camera, packed spheres, packed triangles, description of the data)"""
    from raytracer_amd import scene as S

    if name == "suzanne_x5":
        cam, sp, tr = S.subdivided_suzanne_packed(5)
        return cam, sp, tr, "scene_suzanne.json with its model's triangles split at the edge midpoints 5 times (deterministic)"
    if name == "suzanne_x5_scaled":
        cam, sp, tr = S.subdivided_suzanne_packed(5, 32.0)
        return cam, sp, tr, "scene_suzanne.json, model split 5 times, every coordinate scaled x32 (deterministic)"
    scene = S.load_scene(S.scene_path(name))
    return (scene.camera, S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles),
            "reference scene file scene_%s.json%s (%d triangles, %d spheres); no random inputs"
            % (name, " + models/suzanne_500_tris.obj" if name == "suzanne" else "", len(scene.triangles), len(scene.spheres)))


def cpu_baseline(camera, spheres, tris, width, height, max_bounces, threads):
    """Times the CPU oracle on a rectangle of frame 1 (same scene, resolution, spp), sized for
    about 15 s of CPU work."""
    from tests import oracle_lib as O

    def run(x0, y0, x1, y1):
        orc = O.OracleRenderer((camera, spheres, tris), width, height, max_bounces=max_bounces, threads=threads,
                               rect=(x0, y0, x1, y1))
        orc.set_samples_per_pixel(SPP)
        t = time.perf_counter()
        orc.render()
        return int(orc.counters[0]), time.perf_counter() - t

    cy, cx = height // 2, width // 2
    # warm up (library load, thread start), then calibrate on one pixel per thread (the oracle splits a
    # rectangle into rows, one row per thread at a time); if that is too short to time, on one row per thread
    run(cx, cy, cx + 1, cy + 1)
    cal = max(2, min(threads, 64, height))
    y0 = cy - cal // 2
    rays, dt = run(cx, y0, cx + 1, y0 + cal)
    cal_pixels = cal
    if dt < 0.2:
        rays, dt = run(0, y0, width, y0 + cal)
        cal_pixels = cal * width
    pixels = max(cal, int(round(cal_pixels * 15.0 / max(dt, 1e-3))))
    # a rectangle of about that many pixels around the centre: whole rows if it takes more than one row per thread
    if pixels >= width * cal:
        rows = max(cal, min(height // 2, pixels // width))
        x0, x1 = 0, width
    else:
        rows = cal
        w = max(1, min(width, pixels // rows))
        x0, x1 = (width - w) // 2, (width - w) // 2 + w
    y0 = (height - rows) // 2
    y1 = y0 + rows
    rays, dt = run(x0, y0, x1, y1)
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": "pixels x %d..%d, y %d..%d of frame 1 (%d of %d pixels), %d spp, %d rays in %.1f s"
                  % (x0, x1, y0, y1, (x1 - x0) * (y1 - y0), width * height, SPP, rays, dt),
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from raytracer_amd import capi
    from raytracer_amd import scene as S

    scene_name, WIDTH, HEIGHT, MAX_BOUNCES, default_split, label = CONFIGS[args.config]
    split = args.split or default_split

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

    camera, spheres, tris, data_note = load_config_scene(scene_name)
    by_frames = split == "frames" and use_dist

    def new_context(stripe=True):
        if by_frames or not stripe:
            c = capi.Context(WIDTH, HEIGHT, device=device)
        else:
            c = capi.Context(WIDTH, HEIGHT, device=device, stripe_rows=8, stripe_index=rank, stripe_count=world)
        c.set_max_bounces(MAX_BOUNCES)
        c.set_uniforms(S.pack_uniforms(camera, 1, WIDTH, HEIGHT, SPP))   # before the scene: the camera's distance sizes the boxes
        t = time.perf_counter()
        c.set_scene(spheres, tris)
        c.sync()
        return c, time.perf_counter() - t

    ctx, scene_seconds = new_context()

    # weak scaling: every rank keeps the N = 1 amount of work; strong scaling: K frames whatever N
    frames_per_step = world if args.scaling == "weak" else 1
    K, W = args.steps, args.warmup

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    own = ctx.owned_rows()
    frames_per_launch = ctx.frames_per_launch()
    frames_this_rank = sharding.frames_of_rank(1, K * frames_per_step, rank, world)[1] if by_frames else K * frames_per_step
    local_rows = torch.empty((3, HEIGHT if by_frames else own, WIDTH), dtype=torch.float32, device="cuda") if use_dist else None

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
        gathered, counts = sharding.all_gather_rows_padded(
            local_rows.cpu() if rehearse else local_rows, world,
            counts=[len(sharding.owned_rows(HEIGHT, 8, r, world)) for r in range(world)])
        if rank == 0:
            # one call scatters every other rank's block into the planes (rank 0's own rows are in place)
            gathered = gathered.cuda().contiguous()
            torch.cuda.synchronize()
            ctx.import_gathered_rows(gathered.data_ptr(), max(counts), 0)
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

    if world > 1 and rank == 0 and os.environ.get("RT_BENCH_VERIFY") == "1":
        ref, _ = new_context(stripe=False)
        if by_frames:
            ref.render_frames(1, K * frames_per_step)
            a, b = ref.read_accum(), ctx.read_accum()
            d = np.abs(a - b)
            print("verify: frame split vs single-GPU recurrence: %d of %d values differ by more than 1e-3 (clamped pixels), "
                  "median abs diff %.3g" % (int((d > 1e-3).sum()), d.size, float(np.median(d))), file=sys.stderr)
        else:
            # the assembled image must equal what one GPU renders for the same frames (after the same warmup)
            if W > 0:
                ref.render_frames(1, W * frames_per_step)
            ref.render_frames(1, K * frames_per_step)
            same = bool(np.array_equal(ref.read_accum(), ctx.read_accum()))
            print("verify: assembled image bit-identical to the single-GPU render: %s" % same, file=sys.stderr)
            if not same:
                raise SystemExit("verification failed")
        ref.close()

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

    # roofline: algorithmic flops of the timed launches / HIP-event time of those launches.  The
    # per-ray test counts and the lane utilisation come from an instrumented launch of the same
    # shape as the timed ones: the same context geometry, the same frames, the same frames per launch.
    frames = K * frames_per_step
    batch = frames_per_launch       # frames per kernel launch, as chosen by the library for this context
    launches = max(1, (frames_this_rank + batch - 1) // batch)
    counted_frames = min(frames_this_rank, batch) if frames_this_rank else 1
    cctx, _ = new_context()
    cctx.set_counting(True)
    if by_frames:
        cctx.clear_accum()
        cctx.render_frame_sums(1 + rank, counted_frames, world)
    else:
        cctx.render_frames(1, counted_frames)
    cnt = [int(x) for x in cctx.read_counters()]
    cctx.close()
    c_rays, c_boxpairs, c_tris, c_sph, c_bounces = max(cnt[13], 1), cnt[1], cnt[2], cnt[3], cnt[4]
    lane_util = None
    if cnt[5]:
        lane_util = {"inner": cnt[6] / cnt[5] / 64, "triangle": cnt[8] / max(cnt[7], 1) / 64,
                     "shade": cnt[10] / max(cnt[9], 1) / 64, "shadow_done": cnt[12] / max(cnt[11], 1) / 64,
                     "frames_in_the_instrumented_launch": counted_frames}
    flops_per_ray = (2 * c_boxpairs * F_BOX + c_tris * F_TRI + c_sph * F_SPH + c_bounces * F_SHADE) / c_rays
    brute_flops_per_ray = F_SPH * len(spheres) + F_TRI * len(tris)
    rays_per_launch = rays / world / launches
    launch_ms = max(kernel_ms / launches, 1e-9)
    achieved = flops_per_ray * rays_per_launch / (launch_ms * 1e-3) / 1e12
    # HBM traffic of the trace kernel: measured per frame by the PMC passes (profiles/, same configuration only),
    # scaled to the frames of one timed launch; algorithmic bytes: one 8-byte rgba16float colour written per
    # pixel-frame, the primary ids of the last frame (4 B per pixel), and the scene read once
    frames_in_launch = min(frames_this_rank, batch) if frames_this_rank else 0
    owned_px = (HEIGHT if by_frames else own) * WIDTH
    algorithmic_bytes = frames_in_launch * owned_px * 8 + owned_px * 4 + ctx.scene_stats()["device_bytes"]
    traffic, traffic_note = None, "no PMC measurement for this configuration in profiles/"
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and world == 1:
        try:
            rec = json.load(open(pmc)).get(args.config)
            if rec:
                traffic = rec["hbm_bytes_per_frame"] * frames_in_launch
                traffic_note = "recorded, not measured by this run: profiles/pmc_traffic.json, bytes per frame x frames per launch; " + rec.get("source", "")
        except Exception:
            traffic = None
    valu = None
    l1 = None
    calib = os.path.join(ROOT, "profiles", "valu_calibration.json")
    if os.path.exists(calib) and world == 1:
        try:
            rec = json.load(open(calib))
            k = rec.get("kernel", {}).get(args.config)
            if k:
                valu = {"valu_insts_per_simd_cycle": k["valu_insts_per_simd_cycle"], "ceiling_plain_fma": rec["ceiling_plain_fma"],
                        "frac_of_ceiling": k["valu_insts_per_simd_cycle"] / rec["ceiling_plain_fma"],
                        "lanes_active_per_valu_inst": k["lanes_active_per_valu_inst"], "source": rec.get("source", ""),
                        "recorded_from": "profiles/valu_calibration.json (rocprofv3 --pmc passes of an earlier run of this "
                                         "configuration; not measured by this run)"}
                if "l1_access_ceiling_per_cu_cycle" in rec and "l1_accesses_per_cu_cycle" in k:
                    l1 = {"accesses_per_cu_cycle": k["l1_accesses_per_cu_cycle"], "ceiling": rec["l1_access_ceiling_per_cu_cycle"],
                          "frac_of_ceiling": k["l1_accesses_per_cu_cycle"] / rec["l1_access_ceiling_per_cu_cycle"],
                          "accesses_per_ray": k.get("l1_accesses_per_ray"), "source": rec.get("l1_access_ceiling_source", ""),
                          "recorded_from": "profiles/valu_calibration.json (rocprofv3 --pmc passes of an earlier run of this "
                                           "configuration; not measured by this run)"}
        except Exception:
            valu = None

    out = {
        "metric": "Mrays/s",
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt * 1e3 / max(K, 1),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": data_note,
        "config": {
            "workload": "%s, %d frames x %d spp, max bounces %d, BVH traversal%s"
                        % (label, frames, SPP, MAX_BOUNCES,
                           "" if world == 1 else
                           (", frames dealt to %d GPUs + RCCL all-reduce of the sums" % world if by_frames else
                            ", rows dealt to %d GPUs in 8-row stripes + RCCL all-gather" % world)),
            "baseline_config": args.config,
            "frames": frames,
            "spp_per_frame": SPP,
            "ms_per_frame": dt * 1e3 / max(frames, 1),
            "rays": rays,
            "ray_all_calls_of_the_reference": ray_calls,
            "triangles": int(len(tris)),
            "scene_upload_and_build_s": scene_seconds,
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
            "algorithmic_bytes": algorithmic_bytes,
            "traffic_note": traffic_note,
            "measured_by_this_run": ["achieved", "frac", "kernel_ms_per_launch", "rays_per_launch", "flops_per_ray", "per_ray",
                                     "lane_utilisation", "brute_force_equivalent"],
            "recorded_from_profiles": ["traffic", "valu_issue", "l1_access"],
            "kernel": "render_persistent_kernel (path trace of up to %d frames per launch; the ordered accumulate kernel "
                      "that follows it is included in the timing and is < 0.1%% of it)" % batch,
            "launches": launches,
            "frames_per_launch": frames_in_launch,
            "kernel_ms_per_launch": launch_ms,
            "flops_per_ray": flops_per_ray,
            "rays_per_launch": rays_per_launch,
            "per_ray": {"box_tests": 2 * c_boxpairs / c_rays, "triangle_tests": c_tris / c_rays,
                        "sphere_tests": c_sph / c_rays, "bounces": c_bounces / c_rays},
            "lane_utilisation": lane_util,
            "valu_issue": valu,
            "l1_access": l1,
            "brute_force_equivalent": {"flops_per_ray": brute_flops_per_ray,
                                       "tflops": brute_flops_per_ray * rays_per_launch / (launch_ms * 1e-3) / 1e12},
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        out["cpu_baseline"] = cpu_baseline(camera, spheres, tris, WIDTH, HEIGHT, MAX_BOUNCES, threads)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
