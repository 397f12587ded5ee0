/*
 * rt_hip.h -- C ABI of the MI355X (gfx950) renderer backend, librt_hip.so.
 *
 * The reference (guymor4/raytracer) has no FFI layer: its renderer talks to the
 * GPU through the WebGPU objects created in src/webgpu-renderer.ts and
 * src/accumulator.ts.  Each entry point below replaces one of those WebGPU
 * interactions; the comment on it names the lines it stands in for.  The ctypes
 * binding (raytracer_amd/capi.py) binds exactly these symbols; the N-API addon
 * (raytracer_amd/js/addon) binds all but those that take device pointers.
 *
 * Conventions
 *   - every call returns 0 on success, a negative rt_status otherwise;
 *     rt_last_error() returns the message of the calling thread's last failure;
 *   - the caller owns every host buffer it passes, the library owns all device
 *     memory; buffers use the reference's byte layouts verbatim
 *     (src/gpu-schemas.ts: Sphere 64 B stride, Triangle 96 B stride,
 *     Uniforms 80 B);
 *   - one context lives on one GPU; calls on one context must be serialised by
 *     the caller (the reference drives its renderer from one JS thread);
 *   - rendering calls are asynchronous on the context's HIP stream; the read
 *     calls and rt_sync() wait for it.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument or call order            */
    RT_ERR_DEVICE = -2,    /* HIP runtime error (no GPU, OOM, ...)  */
    RT_ERR_UNSUPPORTED = -3
} rt_status;

/* Traversal used by rt_render_frame. */
typedef enum rt_mode {
    RT_MODE_BVH = 0,         /* BVH traversal, persistent waves (default)                          */
    RT_MODE_BRUTE_FORCE = 1, /* every primitive per ray, the reference's own ray_all, thread/pixel */
    RT_MODE_BVH_TILES = 2    /* BVH traversal, one thread per pixel in the shader's loop nest      */
} rt_mode;

/* Scheduling knobs of the persistent kernel (rt_set_tuning); none changes the output. */
typedef enum rt_tuning {
    RT_TUNE_SHADE_NUM = 0,     /* shade finished lanes when traversing lanes * 8 < active lanes * value (0..8);
                                  default 2, 3 for scenes of more than 4096 triangles */
    RT_TUNE_LEAF_LANES = 1,    /* visit leaves once this many lanes of a wave wait at one (1..64); default 12, 16 for such scenes */
    RT_TUNE_BLOCKS_PER_CU = 2, /* resident 256-thread blocks per CU; 0 = occupancy query                        */
    RT_TUNE_BATCH_FRAMES = 3,  /* frames traced per launch by rt_render_frames; 0 = automatic (default)        */
    RT_TUNE_SHADOW_DONE_LANES = 4, /* finish shadow rays once this many lanes of a wave wait for it (1..64)     */
    RT_TUNE_REFILL_LANES = 5,  /* fetch new pixels once this many lanes of a wave are idle (1..64); 0 = by spp   */
    RT_TUNE_MAX_LEAF = 6,      /* triangles per BVH leaf, 1..8; takes effect at the next rt_set_scene           */
    RT_TUNE_TRI_COST_X10 = 7,  /* SAH cost of a triangle test relative to a node visit, in tenths; next rt_set_scene */
    RT_TUNE_LDS_STATE = 8,     /* where lanes keep rarely touched state: 2 = in LDS as far as it fits (default), 1 = scenes
                                  of <= 4096 triangles keep only shadow-ray state there, 0 = all of it in registers    */
    RT_TUNE_OVERLAP_FRAMES = 9, /* single-frame launches: trace on this many alternating side streams so that consecutive
                                  frames overlap (2..4, default 3; 1 = 2); 0 = trace and accumulate in one kernel on the
                                  context's stream */
    RT_TUNE_SHADE_WAIT = 10,   /* shade finished lanes anyway once they have waited this many traversal turns
                                  (default 12); 0 = never */
    RT_TUNE_MAX_STACK = 11,    /* largest traversal-stack need (entries) the persistent kernel takes on, 1..64 (default
                                  64, all its stack can hold); scenes whose tree needs more are rendered by the
                                  one-thread-per-pixel kernel */
    RT_TUNE_COALESCE_FRAMES = 12, /* rt_render_frame: most frames held back and launched as one batch while the GPU is
                                  busy with earlier ones (default 16); 0 or 1 = every call launches its frame */
    RT_TUNE_MIGRATE_LANES = 13, /* once a launch's work queues are empty, a wave that is down to this many live lanes
                                  hands them to other waves of its block and exits; 0 (default) = every lane finishes in the
                                  wave that started its pixel.  Only in builds with -DRT_MIGRATE=1, ignored otherwise */
    RT_TUNE_TILE_ORDER = 14,   /* 1 (default): a launch hands out first the 8x8 tiles whose longest pixel took the most
                                  rays in the previous launch, so that what is started last is short (frames of 4 samples per pixel or more);
                                  0 = image order */
    RT_TUNE_LINEAR_SPHERES = 15, /* scenes of up to this many spheres test every sphere per ray, as the reference does;
                                  scenes of more keep their spheres in the acceleration structure (default 8); takes
                                  effect at the next rt_set_scene */
    RT_TUNE_END = 16           /* one past the last key */
} rt_tuning;

/* Message of the last failure on the calling thread ("" if none). */
const char *rt_last_error(void);

/* Number of HIP devices visible to the process (0 if none / no driver). */
int rt_device_count(void);

/*
 * Create a renderer for a width x height image on HIP device `device`.
 * Replaces WebGPURenderer.Create's adapter/device/texture/buffer set-up
 * (src/webgpu-renderer.ts:48-118,141-212) and Accumulator.create's three
 * r32float planes (src/accumulator.ts:39-46).  The accumulator planes start
 * zeroed, frame counters start at 0.
 *
 * The context renders the rows y with (y / stripe_rows) % stripe_count ==
 * stripe_index; other rows are never touched.  stripe_count = 1 renders the
 * whole image (stripe_rows is then ignored; pass 8).
 */
int rt_create(int device, int width, int height,
              int stripe_rows, int stripe_index, int stripe_count,
              rt_context **out);

/* Frees all device memory of the context; frames rt_render_frame still holds
 * back are dropped, not rendered (call rt_sync first to have them).  The
 * reference never destroys its renderer; a scene switch builds a new one
 * (src/main.ts:31-35). */
void rt_destroy(rt_context *ctx);

/*
 * Upload the scene.  Replaces the spheres/triangles storage-buffer creation and
 * writes of src/webgpu-renderer.ts:157-188.  `spheres` holds n_spheres records
 * of 64 bytes, `triangles` n_triangles records of 96 bytes; either count may
 * be 0 (then the pointer may be NULL).  Builds the acceleration structure and
 * the emissive-triangle table on the host; may be called again to replace the
 * scene.
 */
int rt_set_scene(rt_context *ctx,
                 const void *spheres, uint32_t n_spheres,
                 const void *triangles, uint32_t n_triangles);

/*
 * Replace the 80-byte Uniforms blob.  Stands in for
 * WebGPURenderer.writeUniforms (src/webgpu-renderer.ts:214-230).  resolution
 * must equal the context's width/height; samplesPerPixel must be >= 1.
 */
int rt_set_uniforms(rt_context *ctx, const void *uniforms80);

/* The shader constant maxBounceCount (public/raytracer_compute.wgsl:331);
 * default 6. */
int rt_set_max_bounces(rt_context *ctx, uint32_t max_bounces);

/*
 * Opt-in extension, off by default because it changes the image: sample the
 * emissive spheres as well as the emissive triangles at every bounce.  The
 * reference's shader carries the sampling function
 * (public/raytracer_compute.wgsl:109-171) and a TODO where it should be called
 * (:226) but never calls it; with this switch it is called right after the
 * triangle-light block (:211-225) and used the same way (one more shadow ray,
 * same MIS weight).  Scenes of up to 4096 primitives are rendered by the
 * persistent kernel, which sets the bounce's second shadow ray aside while the
 * first is traced; larger ones go through the one-thread-per-pixel kernel
 * (RT_MODE_BVH behaves as RT_MODE_BVH_TILES).
 */
int rt_set_sphere_lights(rt_context *ctx, int enable);

/*
 * Opt-in extension, off by default because it changes the image: the debug view.  The reference
 * writes a debugEnabled uniform (src/webgpu-renderer.ts:228, setDebug :279-282) that no shader
 * reads, and its shader library carries ray_triangle_normal_intersect (public/common.wgsl:235-240),
 * which nothing calls: a marker sphere of radius 0.1, 0.2 along the normal from a triangle's
 * centre, coloured by the normal.  With this switch on, frames whose uniforms have debugEnabled
 * != 0 test every triangle's marker in ray_all right after the triangle, under the same selection
 * rule; marker i reports the primary id n_spheres + n_triangles + i.  Such frames are rendered by
 * the brute-force kernel whatever the mode (the markers are in no acceleration structure).
 */
int rt_set_debug_view(rt_context *ctx, int enable);

/* Select the traversal (default RT_MODE_BVH).  All modes produce identical output. */
int rt_set_mode(rt_context *ctx, int mode);

/* Set one scheduling knob (rt_tuning).  No counterpart in the reference. */
int rt_set_tuning(rt_context *ctx, int key, int value);

/*
 * Render one frame with the current uniforms: the compute pass
 * (dispatchWorkgroups, src/webgpu-renderer.ts:253-260) followed by the
 * accumulator pass (src/accumulator.ts:63-78), as one submission
 * (src/webgpu-renderer.ts:267).  Asynchronous, like the reference's
 * fire-and-forget submit.  A frame is launched at once while fewer than two
 * launches are in flight; frames submitted while the GPU is busy with earlier
 * ones are traced concurrently with them (up to three on side streams) or, in a
 * loop that runs ahead of the GPU, held back and launched together as one batch
 * (consecutive frame indices, otherwise equal uniforms; at most
 * RT_TUNE_COALESCE_FRAMES of them, then the call waits on the host for the
 * older launch, so such a loop cannot run ahead without bound).  Frames are
 * accumulated in frame order, and every other entry point first launches what
 * is held back: the effects of the calls are those of launching each frame in
 * its own call, except for two things a caller can observe: an error in a
 * held-back launch (a device error; arguments are checked before a frame is
 * queued) is reported by the call that launches it, and rt_last_render_ms
 * covers the whole batch.  rt_destroy discards frames still held back.
 */
int rt_render_frame(rt_context *ctx);

/*
 * Render n_frames frames with frameIndex = first_frame_index,
 * first_frame_index + 1, ... (all other uniforms as last set), i.e. n_frames
 * iterations of WebGPURenderer.render's body after its frameIndex++
 * (src/webgpu-renderer.ts:248-267).  The stored uniforms' frameIndex ends at
 * the last rendered index.  Asynchronous.
 */
int rt_render_frames(rt_context *ctx, uint32_t first_frame_index, uint32_t n_frames);

/*
 * rt_render_frames for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * result as rt_render_frames, bit for bit, through the same kernels; the per-launch work counter
 * and the frame colours of a launch, which rt_render_frames keeps in buffers it allocates on
 * demand, live in a fixed 64 MiB + 512 B device array that is part of the library image (each GPU
 * the library runs on holds one until the process ends).  Calls no hipMalloc / hipFree.
 * Limits, reported as RT_ERR_UNSUPPORTED with a message:
 *   - the context may own at most 8,388,608 pixels (64 MiB / 8 B; a 3840x2160 image fits);
 *     frames per launch are limited to what fits, so very large images trace one frame per launch;
 *   - scenes whose traversal needs the stack overflow area in global memory (4-wide trees that need
 *     more than 12 stack entries over more than 4096 triangles) get that area from rt_set_scene,
 *     sized for the launch shape in force then; raising RT_TUNE_BLOCKS_PER_CU afterwards is refused;
 *   - a camera that moved so far out that the acceleration structure must be rebuilt is refused.
 * All contexts on one GPU share that GPU's array, on their own streams: calls on different
 * contexts of one GPU must not overlap in time (finish one with rt_sync before starting another),
 * or they race on the array.  The array keeps the previous call's contents; the counter is
 * cleared at the start of every launch and the colours are written before they are read.
 * What the library cannot speak for: the HIP runtime loads a kernel's code and reserves the scratch
 * memory of its register spills (16 MiB on an MI355X) the first time a process launches that kernel.
 * A caller that must see no allocation at all, the runtime's included, renders one frame of a scene
 * of the same kind with rt_render_frames first.
 */
int rt_render_frames_noalloc(rt_context *ctx, uint32_t first_frame_index, uint32_t n_frames);

/* Wait for all submitted work of the context. */
int rt_sync(rt_context *ctx);

/* Milliseconds between the start and the end on the GPU of the most recent
 * launch group (HIP events); waits for it.  For rt_render_frames that is the
 * whole call.  For rt_render_frame it is the frame's own launch, or, when the
 * frame was held back and launched with others as one batch, that whole batch
 * (up to RT_TUNE_COALESCE_FRAMES frames): rt_last_render_frames says how many
 * frames the figure covers, so ms / frames is the per-frame time.  Single
 * frames submitted back to back overlap on the GPU, so their times do not add
 * up to the elapsed time. */
int rt_last_render_ms(rt_context *ctx, float *ms);
/* Frames covered by the figure rt_last_render_ms returns (launches what is held back first). */
int rt_last_render_frames(rt_context *ctx, uint32_t *frames);

/*
 * Read back the accumulator (the three r32float planes of
 * src/accumulator.ts:39-46, which the reference only ever shows on the
 * canvas) as width*height interleaved RGB float32.  Rows the context does not
 * own read as 0.
 */
int rt_read_accum(rt_context *ctx, float *rgb);

/*
 * The image the accumulator pass shows on the canvas (`return vec4(accumulatedColor, 1.0)`,
 * public/accumulator.wgsl:67, src/accumulator.ts:63-78): width*height RGBA8, value * 255 rounded to nearest,
 * alpha 255, no gamma; rows the context does not own are black.  The reference presents every frame and never
 * waits for it; neither does this call with wait = 0: it asks for the image of the frames launched so far
 * (converted on the GPU behind them, copied to pinned host memory on a stream of its own), copies the newest image
 * that has already arrived into `rgba8` and returns at once; *frame_index (may be NULL) receives the frameIndex
 * that image ends with, 0 if none has arrived yet (`rgba8` is then left as it is).  Frames rt_render_frame still
 * holds back are not launched for it: a render loop that presents every frame keeps its batching, and the canvas
 * runs a launch or two behind, as a swap chain does.  wait != 0: launches what is held back, waits, and returns
 * the image of everything submitted.
 */
int rt_present(rt_context *ctx, uint8_t *rgba8, int wait, uint32_t *frame_index);

/* Inverse of rt_read_accum: replace the accumulator planes by width*height
 * interleaved RGB float32 from host memory, and wait for the copy.  No
 * counterpart in the reference.  For hosts without device buffers of their own
 * (the Node addon) that combine frame-split sums (rt_render_frame_sums) of
 * several GPUs in host memory before rt_finish_frame_sums. */
int rt_write_accum(rt_context *ctx, const float *rgb);

/* Read back the last frame's rgba16float intermediate texture
 * (src/webgpu-renderer.ts:108-112): width*height*4 half-float bit patterns. */
int rt_read_frame_f16(rt_context *ctx, uint16_t *rgba);

/*
 * Read back the last frame's primary-hit ids, width*height int32: the index of
 * the object the first ray_all call of the pixel returned, in ray_all's
 * iteration order (public/raytracer_compute.wgsl:15-39): spheres 0..S-1,
 * triangles S..S+T-1, miss -1.  Rows not owned read as -2.  The reference's
 * HitInfo carries no id; this output exists for parity checks.
 */
int rt_read_primary_ids(rt_context *ctx, int32_t *ids);

/*
 * Cumulative 64-bit counters since creation:
 *   [0] ray_all calls as the reference makes them (closest-hit + shadow); this is
 *       the count behind rt_take_triangle_tests
 *   [13] rays actually traversed: the persistent kernel traces the camera ray
 *       of a pixel once per frame and reuses its hit for the frame's other
 *       samples, which the shader starts on the same ray
 *       (public/raytracer_compute.wgsl:329-336)
 *   [1] pairs of box tests (a visit of a 4-wide node counts 2, of a binary node 1)
 *   [2] triangle tests   [3] sphere tests   [4] bounces (shaded hits)
 *   persistent kernel only, per wave: [5] inner-step iterations, [6] lanes active
 *   in them, [7] leaf-step iterations, [8] lanes active in them, [9] shade
 *   passes, [10] lanes active in them, [11] shadow hand-over steps, [12] lanes
 *   active in them; [14..15] reserved, 0
 * [1..12] advance only while counting is enabled (rt_set_counting), which
 * selects an instrumented build of the same kernel; timings are taken with
 * counting off.  [0] always advances.
 */
int rt_read_counters(rt_context *ctx, uint64_t counters[16]);
int rt_set_counting(rt_context *ctx, int enable);

/*
 * Diagnostic, no counterpart in the reference: wave-cycle totals per phase of the instrumented
 * persistent kernel after a run with counting enabled.  out[0..4]: cycles in node steps, triangle
 * steps, shadow hand-over, shading, whole kernel (summed over waves); out[8]: traversal turns;
 * out[9..14]: lanes per state summed over those turns (at an inner node, at a leaf, shadow ray
 * finished, waiting to be shaded, idle, out of work); the rest 0.
 */
int rt_debug_phase_cycles(rt_context *ctx, uint64_t out[16]);

/*
 * Diagnostic, no counterpart in the reference: the time line of the most recent instrumented launch of
 * the persistent kernel (counting enabled), four values per wave of its grid: wall-clock ticks (100 MHz)
 * at which the wave started, found the last work queue empty (0 if it never did) and ended, and the
 * number of its lanes that still had a pixel in hand at that second moment.  Copies at most max_values
 * values into `out` (may be NULL) and stores the number available in *n_values.
 */
int rt_debug_wave_trace(rt_context *ctx, uint64_t *out, size_t max_values, size_t *n_values);

/*
 * The reference's performance counter: performance_counters[0] accumulates
 * arrayLength(&triangles) per ray_all call in a u32
 * (public/raytracer_compute.wgsl:27) and
 * WebGPURenderer.getTriangleTestsSinceLastCheck returns it and zeroes it
 * (src/webgpu-renderer.ts:301-321).  Returns the same value, wrapped modulo
 * 2^32 as the reference's u32 does, since the previous call.
 */
int rt_take_triangle_tests(rt_context *ctx, uint32_t *triangle_tests);

/*
 * Multi-GPU support (no counterpart in the reference, which uses one GPU).
 * rt_owned_rows: number of image rows this context renders.
 * rt_export_owned_rows: copy the accumulator rows the context owns, in
 *   increasing y, into device memory `dst` as three planes of
 *   owned_rows*width float32 (R plane, G plane, B plane), on the context's
 *   stream.  `dst` must be device memory on the context's GPU (for instance a
 *   torch tensor's data_ptr) of 3*owned_rows*width*4 bytes.
 * rt_import_rows: inverse for a gathered image: scatter such a block that
 *   came from the context with stripe_index `from_index` into this context's
 *   accumulator planes (used on the rank that assembles the full image).
 */
int rt_owned_rows(rt_context *ctx, int *rows);
int rt_export_owned_rows(rt_context *ctx, void *dst_device);
int rt_import_rows(rt_context *ctx, const void *src_device, int from_index);
/* rt_import_gathered_rows: the result of an all-gather in one call: `src_device` holds one block per
 * stripe index, in index order, each padded to rows_per_block rows (3 planes of rows_per_block*width
 * float32; the rows beyond a stripe's own count are ignored); every block except that of
 * `skip_index` (pass -1 for none; usually the caller's own index, whose rows are in place) is
 * scattered into this context's accumulator planes, on the context's stream. */
int rt_import_gathered_rows(rt_context *ctx, const void *src_device, int rows_per_block, int skip_index);
/* The same exchange through host memory, for hosts that have no device buffers of their own (the
 * Node addon): rt_read_owned_rows waits for the context and fills `planes` with 3*owned_rows*width
 * float32 in the layout above; rt_write_rows scatters such a block that came from the context with
 * stripe_index `from_index` into this context's accumulator planes and waits for it. */
int rt_read_owned_rows(rt_context *ctx, float *planes);
/* rt_rows_of_stripe: number of rows the context with stripe_index `stripe_index` (same image, same
 * stripe_rows and stripe_count as this one) owns, i.e. the height of the block it exports. */
int rt_rows_of_stripe(rt_context *ctx, int stripe_index, int *rows);
int rt_write_rows(rt_context *ctx, const float *planes, int from_index);

/*
 * Frame-split rendering across GPUs (no counterpart in the reference).  Frames
 * are dealt to ranks instead of rows: rank r of G renders frames first + r,
 * first + r + G, ... of the whole image.  The accumulator recurrence
 * (public/accumulator.wgsl:53-60) cannot run across ranks, so in this mode the
 * planes collect the plain f32 sum of the frames' rgba16float colours; the
 * ranks' sums are added (one all-reduce on the exported planes) and
 * rt_finish_frame_sums turns the total into the accumulator value
 * saturate(sum / (n_frames_total + 1)), which is what the recurrence yields
 * for frames 1..n when no intermediate value is clamped.  Unlike the row
 * split this is not bit-identical to a single-GPU run: the order of the f32
 * additions differs and the per-frame saturate of pixels whose running mean
 * exceeds 1 is skipped.
 * rt_clear_accum: zero the planes (before the first rt_render_frame_sums).
 * rt_render_frame_sums: trace n_frames frames first, first + stride, ... and
 *   add their colours to the planes.
 * rt_export_accum / rt_import_accum: copy the three full planes
 *   (3*width*height float32, R plane, G plane, B plane) to / from device
 *   memory on the context's GPU, on the context's stream.
 */
int rt_clear_accum(rt_context *ctx);
int rt_render_frame_sums(rt_context *ctx, uint32_t first_frame_index, uint32_t n_frames, uint32_t stride);
int rt_export_accum(rt_context *ctx, void *dst_device);
int rt_import_accum(rt_context *ctx, const void *src_device);
int rt_finish_frame_sums(rt_context *ctx, uint32_t n_frames_total);

/* How many frames rt_render_frames puts into one kernel launch for this context
 * (RT_TUNE_BATCH_FRAMES, or the automatic choice), for reports. */
int rt_frames_per_launch(rt_context *ctx, uint32_t *frames);

/* Acceleration-structure statistics of the current scene, for reports:
 * out[0] nodes, out[1] leaves, out[2] max depth, out[3] bytes on device. */
int rt_scene_stats(rt_context *ctx, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
