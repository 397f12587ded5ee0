/*
 * rt_napi.c -- thin N-API addon over the C ABI of librt_hip.so (include/rt_hip.h).
 * One JS function per entry point; no logic of its own.  Typed arrays go in and
 * out without conversion: Float32Array records in the reference's byte layouts
 * (src/gpu-schemas.ts) and read-backs as Float32Array / Uint16Array / Int32Array.
 * Every failure throws an Error carrying rt_last_error().
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/rt_hip.h"

#define NAPI_OK(call)                                                     \
    do {                                                                  \
        if ((call) != napi_ok) {                                          \
            napi_throw_error(env, NULL, "N-API call failed: " #call);     \
            return NULL;                                                  \
        }                                                                 \
    } while (0)

static napi_value throw_rt(napi_env env, const char *what)
{
    char msg[512];
    const char *e = rt_last_error();
    strncpy(msg, what, sizeof msg - 1);
    msg[sizeof msg - 1] = 0;
    strncat(msg, ": ", sizeof msg - strlen(msg) - 1);
    strncat(msg, e ? e : "", sizeof msg - strlen(msg) - 1);
    napi_throw_error(env, NULL, msg);
    return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}

/* What a JS renderer handle points at: the context (NULL after destroy) and the image size it was
 * created with, which sizes every read-back buffer. */
typedef struct handle_box {
    rt_context *ctx;
    uint32_t width, height;
} handle_box;

static handle_box *get_box(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a renderer handle");
        return NULL;
    }
    return (handle_box *)p;
}

static rt_context *get_ctx(napi_env env, napi_value v)
{
    handle_box *box = get_box(env, v);
    return box ? box->ctx : NULL; /* NULL after destroy */
}

static int get_i32(napi_env env, napi_value v, int32_t *out)
{
    if (napi_get_value_int32(env, v, out) != napi_ok) {
        napi_throw_type_error(env, NULL, "expected an integer");
        return 0;
    }
    return 1;
}

static int get_u32(napi_env env, napi_value v, uint32_t *out)
{
    if (napi_get_value_uint32(env, v, out) != napi_ok) {
        napi_throw_type_error(env, NULL, "expected an unsigned integer");
        return 0;
    }
    return 1;
}

/* Float32Array -> pointer + element count */
static int get_f32_array(napi_env env, napi_value v, float **data, size_t *len)
{
    napi_typedarray_type type;
    napi_value ab;
    size_t off;
    void *p;
    if (napi_get_typedarray_info(env, v, &type, len, &p, &ab, &off) != napi_ok || type != napi_float32_array) {
        napi_throw_type_error(env, NULL, "expected a Float32Array");
        return 0;
    }
    *data = (float *)p;
    return 1;
}

static void finalize_handle(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    handle_box *box = (handle_box *)data;
    if (box->ctx) rt_destroy(box->ctx);
    free(box);
}

static napi_value js_device_count(napi_env env, napi_callback_info info)
{
    (void)info;
    napi_value out;
    NAPI_OK(napi_create_int32(env, rt_device_count(), &out));
    return out;
}

static napi_value js_create(napi_env env, napi_callback_info info)
{
    napi_value a[6];
    int32_t v[6];
    if (!get_args(env, info, 6, a)) return NULL;
    for (int i = 0; i < 6; i++)
        if (!get_i32(env, a[i], &v[i])) return NULL;
    rt_context *ctx = NULL;
    if (rt_create(v[0], v[1], v[2], v[3], v[4], v[5], &ctx) != RT_OK) return throw_rt(env, "rt_create");
    handle_box *box = (handle_box *)malloc(sizeof *box);
    if (!box) { rt_destroy(ctx); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    box->ctx = ctx;
    box->width = (uint32_t)v[1];
    box->height = (uint32_t)v[2];
    napi_value out;
    if (napi_create_external(env, box, finalize_handle, NULL, &out) != napi_ok) {
        rt_destroy(ctx); free(box);
        napi_throw_error(env, NULL, "napi_create_external failed");
        return NULL;
    }
    return out;
}

static napi_value js_destroy(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    void *p = NULL;
    if (!get_args(env, info, 1, a)) return NULL;
    if (napi_get_value_external(env, a[0], &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a renderer handle");
        return NULL;
    }
    handle_box *box = (handle_box *)p;
    if (box->ctx) { rt_destroy(box->ctx); box->ctx = NULL; }
    return NULL;
}

#define NEED_CTX(ctx, v)                                               \
    rt_context *ctx = get_ctx(env, v);                                 \
    if (!ctx) {                                                        \
        bool pending = false;                                          \
        napi_is_exception_pending(env, &pending);                      \
        if (!pending) napi_throw_error(env, NULL, "renderer destroyed"); \
        return NULL;                                                   \
    }

static napi_value js_set_scene(napi_env env, napi_callback_info info)
{
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    float *sp, *tr;
    size_t nsp, ntr;
    if (!get_f32_array(env, a[1], &sp, &nsp) || !get_f32_array(env, a[2], &tr, &ntr)) return NULL;
    if (nsp % 16 != 0 || ntr % 24 != 0) {
        napi_throw_type_error(env, NULL, "spheres need 16 floats per record, triangles 24");
        return NULL;
    }
    if (rt_set_scene(ctx, nsp ? sp : NULL, (uint32_t)(nsp / 16), ntr ? tr : NULL, (uint32_t)(ntr / 24)) != RT_OK)
        return throw_rt(env, "rt_set_scene");
    return NULL;
}

static napi_value js_set_uniforms(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    float *u;
    size_t n;
    if (!get_f32_array(env, a[1], &u, &n)) return NULL;
    if (n != 20) { napi_throw_type_error(env, NULL, "uniforms must be 20 floats (80 bytes)"); return NULL; }
    if (rt_set_uniforms(ctx, u) != RT_OK) return throw_rt(env, "rt_set_uniforms");
    return NULL;
}

/* (handle, int) setters */
#define INT_SETTER(jsname, call)                                              \
    static napi_value jsname(napi_env env, napi_callback_info info)           \
    {                                                                         \
        napi_value a[2];                                                      \
        if (!get_args(env, info, 2, a)) return NULL;                          \
        NEED_CTX(ctx, a[0]);                                                  \
        int32_t v;                                                            \
        if (!get_i32(env, a[1], &v)) return NULL;                             \
        if (call != RT_OK) return throw_rt(env, #call);                       \
        return NULL;                                                          \
    }
INT_SETTER(js_set_max_bounces, rt_set_max_bounces(ctx, (uint32_t)v))
INT_SETTER(js_set_mode, rt_set_mode(ctx, v))
INT_SETTER(js_set_counting, rt_set_counting(ctx, v))
INT_SETTER(js_set_sphere_lights, rt_set_sphere_lights(ctx, v))
INT_SETTER(js_set_debug_view, rt_set_debug_view(ctx, v))

static napi_value js_set_tuning(napi_env env, napi_callback_info info)
{
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    int32_t k, v;
    if (!get_i32(env, a[1], &k) || !get_i32(env, a[2], &v)) return NULL;
    if (rt_set_tuning(ctx, k, v) != RT_OK) return throw_rt(env, "rt_set_tuning");
    return NULL;
}

static napi_value js_render_frame(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    if (rt_render_frame(ctx) != RT_OK) return throw_rt(env, "rt_render_frame");
    return NULL;
}

static napi_value js_render_frames(napi_env env, napi_callback_info info)
{
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    uint32_t first, n;
    if (!get_u32(env, a[1], &first) || !get_u32(env, a[2], &n)) return NULL;
    if (rt_render_frames(ctx, first, n) != RT_OK) return throw_rt(env, "rt_render_frames");
    return NULL;
}

static napi_value js_sync(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    if (rt_sync(ctx) != RT_OK) return throw_rt(env, "rt_sync");
    return NULL;
}

static napi_value js_last_render_ms(napi_env env, napi_callback_info info)
{
    napi_value a[1], out;
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    float ms = 0;
    if (rt_last_render_ms(ctx, &ms) != RT_OK) return throw_rt(env, "rt_last_render_ms");
    NAPI_OK(napi_create_double(env, ms, &out));
    return out;
}

/* (handle, Uint8Array or Uint8ClampedArray of width*height*4, wait) -> frameIndex the image ends with (0: none yet) */
static napi_value js_present(napi_env env, napi_callback_info info)
{
    napi_value a[3], out;
    if (!get_args(env, info, 3, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    handle_box *box = get_box(env, a[0]);
    napi_typedarray_type type;
    napi_value ab;
    size_t len, off;
    void *p;
    if (napi_get_typedarray_info(env, a[1], &type, &len, &p, &ab, &off) != napi_ok ||
        (type != napi_uint8_array && type != napi_uint8_clamped_array)) {
        napi_throw_type_error(env, NULL, "expected a Uint8Array or Uint8ClampedArray");
        return NULL;
    }
    if (len != (size_t)box->width * box->height * 4) {
        napi_throw_range_error(env, NULL, "expected width * height * 4 bytes");
        return NULL;
    }
    int32_t wait;
    if (!get_i32(env, a[2], &wait)) return NULL;
    uint32_t frame = 0;
    if (rt_present(ctx, (uint8_t *)p, wait, &frame) != RT_OK) return throw_rt(env, "rt_present");
    NAPI_OK(napi_create_uint32(env, frame, &out));
    return out;
}

static napi_value js_last_render_frames(napi_env env, napi_callback_info info)
{
    napi_value a[1], out;
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    uint32_t n = 0;
    if (rt_last_render_frames(ctx, &n) != RT_OK) return throw_rt(env, "rt_last_render_frames");
    NAPI_OK(napi_create_uint32(env, n, &out));
    return out;
}

/* Frame-split rendering (rt_render_frame_sums and friends); the sums travel between renderers through
 * readAccum / writeAccum, i.e. host memory. */
static napi_value js_clear_accum(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    if (rt_clear_accum(ctx) != RT_OK) return throw_rt(env, "rt_clear_accum");
    return NULL;
}

static napi_value js_render_frame_sums(napi_env env, napi_callback_info info)
{
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    uint32_t first, n, stride;
    if (!get_u32(env, a[1], &first) || !get_u32(env, a[2], &n) || !get_u32(env, a[3], &stride)) return NULL;
    if (rt_render_frame_sums(ctx, first, n, stride) != RT_OK) return throw_rt(env, "rt_render_frame_sums");
    return NULL;
}

static napi_value js_finish_frame_sums(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    uint32_t total;
    if (!get_u32(env, a[1], &total)) return NULL;
    if (rt_finish_frame_sums(ctx, total) != RT_OK) return throw_rt(env, "rt_finish_frame_sums");
    return NULL;
}

/* (handle, Float32Array of width*height*3 interleaved RGB) */
static napi_value js_write_accum(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    handle_box *box = get_box(env, a[0]);
    float *data;
    size_t n;
    if (!get_f32_array(env, a[1], &data, &n)) return NULL;
    if (n != (size_t)box->width * box->height * 3) {
        napi_throw_range_error(env, NULL, "expected width * height * 3 floats");
        return NULL;
    }
    if (rt_write_accum(ctx, data) != RT_OK) return throw_rt(env, "rt_write_accum");
    return NULL;
}

/* Read-backs into a fresh typed array.  The library always writes the context's own width x height
 * elements, so the buffer is sized from the size the handle was created with; the (width, height)
 * the caller passes must agree with it. */
static napi_value read_into(napi_env env, napi_callback_info info, napi_typedarray_type type, size_t elem_bytes,
                            size_t per_pixel, int which)
{
    napi_value a[3], ab, out;
    if (!get_args(env, info, 3, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    handle_box *box = get_box(env, a[0]);
    uint32_t w, h;
    if (!get_u32(env, a[1], &w) || !get_u32(env, a[2], &h)) return NULL;
    if (w != box->width || h != box->height) {
        napi_throw_range_error(env, NULL, "read-back size differs from the size the renderer was created with");
        return NULL;
    }
    size_t n = (size_t)box->width * box->height * per_pixel;
    void *data = NULL;
    NAPI_OK(napi_create_arraybuffer(env, n * elem_bytes, &data, &ab));
    int rc = which == 0 ? rt_read_accum(ctx, (float *)data)
           : which == 1 ? rt_read_frame_f16(ctx, (uint16_t *)data)
                        : rt_read_primary_ids(ctx, (int32_t *)data);
    if (rc != RT_OK) return throw_rt(env, "read-back");
    NAPI_OK(napi_create_typedarray(env, type, n, ab, 0, &out));
    return out;
}

static napi_value js_read_accum(napi_env env, napi_callback_info info) { return read_into(env, info, napi_float32_array, 4, 3, 0); }
static napi_value js_read_frame_f16(napi_env env, napi_callback_info info) { return read_into(env, info, napi_uint16_array, 2, 4, 1); }
static napi_value js_read_primary_ids(napi_env env, napi_callback_info info) { return read_into(env, info, napi_int32_array, 4, 1, 2); }

static napi_value u64_array(napi_env env, const uint64_t *v, size_t n)
{
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, n, &arr));
    for (size_t i = 0; i < n; i++) {
        napi_value x;
        NAPI_OK(napi_create_double(env, (double)v[i], &x)); /* exact below 2^53 */
        NAPI_OK(napi_set_element(env, arr, (uint32_t)i, x));
    }
    return arr;
}

static napi_value js_read_counters(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    uint64_t c[16];
    if (rt_read_counters(ctx, c) != RT_OK) return throw_rt(env, "rt_read_counters");
    return u64_array(env, c, 16);
}

static napi_value js_scene_stats(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    uint64_t c[4];
    if (rt_scene_stats(ctx, c) != RT_OK) return throw_rt(env, "rt_scene_stats");
    return u64_array(env, c, 4);
}

static napi_value js_take_triangle_tests(napi_env env, napi_callback_info info)
{
    napi_value a[1], out;
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    uint32_t v = 0;
    if (rt_take_triangle_tests(ctx, &v) != RT_OK) return throw_rt(env, "rt_take_triangle_tests");
    NAPI_OK(napi_create_uint32(env, v, &out));
    return out;
}

static napi_value js_owned_rows(napi_env env, napi_callback_info info)
{
    napi_value a[1], out;
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    int rows = 0;
    if (rt_owned_rows(ctx, &rows) != RT_OK) return throw_rt(env, "rt_owned_rows");
    NAPI_OK(napi_create_int32(env, rows, &out));
    return out;
}

/* Accumulator rows the context owns, as three planes (R, G, B) of ownedRows x width floats. */
static napi_value js_read_owned_rows(napi_env env, napi_callback_info info)
{
    napi_value a[1], ab, out;
    if (!get_args(env, info, 1, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    handle_box *box = get_box(env, a[0]);
    int rows = 0;
    if (rt_owned_rows(ctx, &rows) != RT_OK) return throw_rt(env, "rt_owned_rows");
    size_t n = (size_t)rows * box->width * 3;
    void *data = NULL;
    NAPI_OK(napi_create_arraybuffer(env, n * sizeof(float), &data, &ab));
    if (n && rt_read_owned_rows(ctx, (float *)data) != RT_OK) return throw_rt(env, "rt_read_owned_rows");
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, n, ab, 0, &out));
    return out;
}

/* (handle, Float32Array block of the context with stripe index fromIndex, fromIndex) */
static napi_value js_write_rows(napi_env env, napi_callback_info info)
{
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return NULL;
    NEED_CTX(ctx, a[0]);
    handle_box *box = get_box(env, a[0]);
    float *data;
    size_t n;
    int32_t from;
    if (!get_f32_array(env, a[1], &data, &n) || !get_i32(env, a[2], &from)) return NULL;
    int rows = 0;
    if (rt_rows_of_stripe(ctx, from, &rows) != RT_OK) return throw_rt(env, "rt_rows_of_stripe");
    if (n != (size_t)rows * box->width * 3) {
        napi_throw_range_error(env, NULL, "block size differs from 3 * rows of that stripe * width");
        return NULL;
    }
    if (n && rt_write_rows(ctx, data, from) != RT_OK) return throw_rt(env, "rt_write_rows");
    return NULL;
}

static napi_value init(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"deviceCount", js_device_count}, {"create", js_create}, {"destroy", js_destroy},
        {"setScene", js_set_scene}, {"setUniforms", js_set_uniforms}, {"setMaxBounces", js_set_max_bounces},
        {"setMode", js_set_mode}, {"setCounting", js_set_counting}, {"setTuning", js_set_tuning},
        {"setSphereLights", js_set_sphere_lights}, {"setDebugView", js_set_debug_view},
        {"readOwnedRows", js_read_owned_rows}, {"writeRows", js_write_rows},
        {"renderFrame", js_render_frame}, {"renderFrames", js_render_frames}, {"sync", js_sync},
        {"lastRenderMs", js_last_render_ms}, {"readAccum", js_read_accum}, {"readFrameF16", js_read_frame_f16},
        {"readPrimaryIds", js_read_primary_ids}, {"readCounters", js_read_counters},
        {"present", js_present}, {"lastRenderFrames", js_last_render_frames}, {"clearAccum", js_clear_accum}, {"renderFrameSums", js_render_frame_sums},
        {"finishFrameSums", js_finish_frame_sums}, {"writeAccum", js_write_accum},
        {"takeTriangleTests", js_take_triangle_tests}, {"ownedRows", js_owned_rows}, {"sceneStats", js_scene_stats},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
            napi_throw_error(env, NULL, "addon initialisation failed");
            return NULL;
        }
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
