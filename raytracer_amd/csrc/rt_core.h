// rt_core.h -- path-tracer core shared by the HIP kernels (rt_hip.hip) and the
// host-side scene preparation (rt_scene.cpp).  Everything here compiles both
// as device code (hipcc, gfx950) and as plain host C++.
//
// Numeric contract: every f32 operation that feeds a result (intersection
// tests, shading, sampling, accumulation) is a single IEEE binary32 operation
// in the order written, with correctly rounded division and square root and no
// FMA contraction; the translation unit must be built with -ffp-contract=off.
// The only fused / approximate arithmetic is inside the bounding-box tests,
// which decide what is visited, never what is returned.
//
// Reference restated: public/common.wgsl, public/raytracer_compute.wgsl,
// public/accumulator.wgsl of guymor4/raytracer (line numbers in the comments).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

#ifndef RT_SCALAR_READS
#define RT_SCALAR_READS 1
#endif
struct f3 { float x, y, z; };
struct alignas(16) f4 { float x, y, z, w; };

// A 16-byte read whose address is the same in every lane of a wave (a sphere or light record indexed
// by a loop counter).  Scene tables are never written while a kernel runs, so on the GPU the read goes
// through the constant address space: the compiler then fetches it once per wave with a scalar load
// instead of occupying the vector-memory path with 64 copies of one address.
RT_HD f4 read_uniform(const f4 *p)
{
#if defined(__HIP_DEVICE_COMPILE__) && RT_SCALAR_READS
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    v4f_ v = *(const __attribute__((address_space(4))) v4f_ *)(const void *)p;
    f4 r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
#else
    return *p;
#endif
}

// A 16-byte read that must stay one load instruction: read as four separate floats, a record whose
// fourth component is used later than the first three is fetched as 12 + 4 bytes, and every vector
// load instruction occupies the memory path for at least 16 clocks.
RT_HD f4 read_whole(const f4 *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    v4f_ v = *(const v4f_ *)(const void *)p;
    f4 r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
#else
    return *p;
#endif
}

RT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD f3 add(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 sub(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 mul(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 scale(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_HD f3 divs(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_HD f3 neg(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_HD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD f3 cross(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD float length(f3 a) { return sqrtf(dot(a, a)); }
RT_HD f3 normalize(f3 a) { return divs(a, length(a)); }
RT_HD f3 xyz(f4 a) { return mk3(a.x, a.y, a.z); }
RT_HD f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
RT_HD f4 mk4(f3 a, float w) { return mk4(a.x, a.y, a.z, w); }

RT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
RT_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

constexpr float kPi = 3.14159265358979323846f; // common.wgsl:4

// common.wgsl:66-73
RT_HD uint32_t wang_hash(uint32_t s)
{
    uint32_t ns = (s ^ 61u) ^ (s >> 16);
    ns = ns * 9u;
    ns = ns ^ (ns >> 4);
    ns = ns * 0x27d4eb2du;
    ns = ns ^ (ns >> 15);
    return ns;
}

// common.wgsl:60-64
RT_HD float rand_f(uint32_t &state)
{
    state = state * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (float)((word >> 22) ^ word) * u2f(0x2f800004u);
}

// sin and cos as the renderer defines them wherever the shader calls sin(),
// cos() or tan(): quadrant k = rint(x * 2/pi), Cody-Waite reduction with the
// three-part pi/2 below, cephes single-precision polynomials on [-pi/4, pi/4].
RT_HD void sincos_f(float x, float &s, float &c)
{
    float k = rintf(x * 0.636619772367581343f);
    int q = (int)k;
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    float sn = r + r * (z * ps);
    float pc = 2.443315711809948e-5f;
    pc = pc * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    float cs = (1.0f - 0.5f * z) + (z * z) * pc;
    int m = q & 3;
    float s0 = (m & 1) ? cs : sn;
    float c0 = (m & 1) ? sn : cs;
    s = (m & 2) ? -s0 : s0;
    c = (m == 1 || m == 2) ? -c0 : c0;
}

// raytracer_compute.wgsl:11-13
RT_HD float luminance(f3 c) { return dot(c, mk3(0.2126f, 0.7152f, 0.0722f)); }

// ---------------------------------------------------------------------------
// Device-resident scene (all pointers are device pointers in the kernels, host
// pointers in host-side checks).
//
// nodes      4 x f4 per node: child boxes (lo0,hi0,lo1,hi1 packed) + two child
//            links.  A link >= 0 is a node index; a link < 0 encodes a leaf as
//            ~((first << 3) | (count - 1)) over tri_isect.
// nodes4     8 x f4 per 4-wide node (persistent kernel): lo.x[4] lo.y[4] lo.z[4]
//            hi.x[4] hi.y[4] hi.z[4] links[4] pad; same link encoding; unused
//            slots (2 and 3 only) carry the link kNoRoot.
// nodes4q    the same tree, same node numbers, in 4 x f4 = 64 bytes per node (persistent kernel, large scenes):
//            the node's own box origin and a power-of-two step per axis, the four child boxes as 8-bit
//            multiples of the steps from the origin (rounded outwards), the four links:
//            (origin.xyz, step exponents) (lo.x[4] lo.y[4] lo.z[4] hi.x[4] as bytes) (hi.y[4] hi.z[4] as
//            bytes, link0, link1) (link2, link3, 0, 0); see lane_visit4q in rt_lane.h
// tri_isect  3 x f4 per triangle in leaf order:
//            (v0.xyz, e01.x) (e01.yz, e02.xy) (e02.z, unit normal.xyz) -- the normal is
//            tri_shade's, so that the back-face test needs no further fetch
// tri_orig   original index of the triangle at each leaf position.  The persistent kernel
//            numbers triangles by leaf position while it traverses (the record has no room
//            for the index) and asks this table only where original order matters: ties in t
//            and the primary-hit id it reports.
// tri_shade  3 x f4 per triangle in original order:
//            (normal.xyz, smoothness) (color.xyz, specularProbability)
//            (emissionColor*emissionStrength, 0)
// tri_shade_leaf  the same records in leaf order
// tri_raw    the reference's 96-byte records (brute-force mode only)
// spheres    the reference's 64-byte records = 4 x f4
// sph_leaf   scenes of many spheres (BuildOptions::linear_spheres): the spheres are leaves of the tree too.  A leaf
//            holds triangles or spheres, never both; leaf positions n_tris and up stand for spheres, position
//            n_tris + k for sph_leaf[k] = (centre, radius) of the sphere with original index sph_orig[k]
// lights     5 x f4 per emissive triangle in original order, plus one trailing
//            all-zero "nothing selected" entry:
//            (v0.xyz, area) (v1.xyz, power/total_power) (v2.xyz, running power sum)
//            (light normal.xyz, 0) (emission.xyz, 0)
// ---------------------------------------------------------------------------
struct SceneView {
    const f4 *nodes;
    const f4 *nodes4;
    const f4 *nodes4q;
    const f4 *tri_isect;
    const f4 *tri_shade;
    const f4 *tri_shade_leaf;
    const uint32_t *tri_orig;
    const f4 *tri_raw;
    const f4 *spheres;
    const f4 *sph_leaf;        // (centre, radius) of the spheres that are leaves of the tree, in leaf order
    const uint32_t *sph_orig;  // their original indices
    const f4 *lights;
    uint32_t n_linear_spheres; // spheres every ray tests one by one: n_spheres, or 0 when they are in the tree
    uint32_t n_spheres;
    uint32_t n_tris;
    uint32_t n_lights;       // entries with emissionStrength > 0
    uint32_t n_lights_pass1; // entries counted by the first scan (strength not <= 0)
    float total_power;
    int32_t root;            // root link (leaf-encoded when the tree is one leaf); 0x7fffffff if no triangles
    int32_t root4;           // the same for the 4-wide tree
    uint32_t debug_view;     // opt-in extension (brute-force ray_all only): every triangle's normal marker is tested too
};

constexpr int32_t kNoRoot = 0x7fffffff;
constexpr int kTriRecord = 3; // f4 per tri_isect record

// Per-frame constants derived from the 80-byte uniforms on the host.
// raytracer_compute.wgsl:290-316 computes these per pixel from the same inputs.
struct FrameParams {
    f3 cam_pos, right, up, forward;
    float focal_length;
    float aspect;
    float res_x, res_y;
    float spp_f;       // uniforms.samplesPerPixel
    float frame_f;     // uniforms.frameIndex
    uint32_t spp;      // u32(uniforms.samplesPerPixel)
    uint32_t seed_add; // u32(uniforms.frameIndex) * 12345
    uint32_t res_x_u;  // u32(uniforms.resolution.x)
    uint32_t max_bounces;
    int32_t width, height;
    uint32_t sphere_lights; // opt-in extension: every bounce also samples the emissive spheres (see rt_pixel.h)
};

struct Hit { // common.wgsl:50-57
    float t;
    f3 color, normal, emission;
    float smoothness, specularProbability;
};

struct TestCounts {
    uint32_t boxes, tris, spheres, bounces;
};

// ---------------------------------------------------------------------------
// Primitive tests
// ---------------------------------------------------------------------------

// common.wgsl:137-168.  Returns the reported t (which is t1 even when the t2
// branch is taken) or -1; t_normal receives the parameter the normal is taken at.
RT_HD float sphere_t(f3 ro, f3 rd, f4 s0, float &t_normal)
{
    f3 oc = sub(ro, xyz(s0));
    float a = dot(rd, rd);
    float b = 2.0f * dot(oc, rd);
    float c = dot(oc, oc) - s0.w * s0.w;
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant < 0.0f) return -1.0f;
    float sq = sqrtf(discriminant);
    float t1 = (-b - sq) / (2.0f * a);
    float t2 = (-b + sq) / (2.0f * a);
    if (t1 > 0.01f) { t_normal = t1; return t1; }
    if (t2 > 0.01f) { t_normal = t2; return t1; }
    return -1.0f;
}

// common.wgsl:170-200 up to the t test, on precomputed edges.  Returns t or -1.
RT_HD float triangle_t(f3 ro, f3 rd, f3 v0, f3 e01, f3 e02)
{
    f3 h = cross(rd, e02);
    float a = dot(e01, h);
    if (fabsf(a) < 0.001f) return -1.0f;
    float f = 1.0f / a;
    f3 s = sub(ro, v0);
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    f3 q = cross(s, e01);
    float v = f * dot(rd, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = f * dot(e02, q);
    if (t < 0.01f) return -1.0f;
    return t;
}

// ---------------------------------------------------------------------------
// Traversal.  `Stack` provides reset(), push(int32_t), pop() -> int32_t, empty().
// ---------------------------------------------------------------------------

struct RayBox {
    f3 inv, ood; // 1/d and -(o/d): the slab distance of a plane at x is fma(x, inv, ood)
};

RT_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

RT_HD RayBox make_raybox(f3 o, f3 d)
{
    RayBox r;
    // A zero component would give inf and then inf - inf = NaN in the fused slab
    // distances; a huge finite reciprocal keeps them finite with the right signs.
    r.inv = mk3(clampf(fast_rcp(d.x), -1e30f, 1e30f), clampf(fast_rcp(d.y), -1e30f, 1e30f),
                clampf(fast_rcp(d.z), -1e30f, 1e30f));
    r.ood = mk3(-(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z));
    return r;
}

// Entry distance of the ray into the box clipped to [0, tmax]; hit iff the
// returned near <= far.  The reciprocals are finite (make_raybox), so for
// coordinates below 1e8 in magnitude every slab distance is finite.
RT_HD bool slab(const RayBox &rb, float lox, float loy, float loz, float hix, float hiy, float hiz,
                float tmax, float &tnear)
{
    float ax = __builtin_fmaf(lox, rb.inv.x, rb.ood.x), bx = __builtin_fmaf(hix, rb.inv.x, rb.ood.x);
    float ay = __builtin_fmaf(loy, rb.inv.y, rb.ood.y), by = __builtin_fmaf(hiy, rb.inv.y, rb.ood.y);
    float az = __builtin_fmaf(loz, rb.inv.z, rb.ood.z), bz = __builtin_fmaf(hiz, rb.inv.z, rb.ood.z);
    float n = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    float f = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = n;
    return n <= f;
}

RT_HD bool all_nan(f3 d) { return (d.x != d.x) && (d.y != d.y) && (d.z != d.z); }

// Whether leaf position `first` stands for a sphere (SceneView::sph_leaf).
RT_HD bool leaf_holds_spheres(const SceneView &sc, uint32_t first) { return first >= sc.n_tris; }

// Closest hit with the reference's selection rule (raytracer_compute.wgsl:15-39):
// among hits with t > 0 the smallest t wins, ties go to the lowest index in
// ray_all order (spheres first).  Returns the id, -1 on a miss.
template <class Stack>
RT_HD int32_t trace_closest(const SceneView &sc, f3 ro, f3 rd, Stack &stack, float &out_t, TestCounts *cnt)
{
    float best_t = INFINITY;
    int32_t best_id = 0x7fffffff;
    // A direction that is NaN in every component fails every test of the
    // reference (all comparisons false, t = NaN is not > 0): a miss.
    if (all_nan(rd)) { out_t = -1.0f; return -1; }

    for (uint32_t i = 0; i < sc.n_linear_spheres; i++) {
        float tn;
        float t = sphere_t(ro, rd, sc.spheres[4 * i], tn);
        if (cnt) cnt->spheres++;
        if (t > 0.0f && (t < best_t || (t == best_t && (int32_t)i < best_id))) { best_t = t; best_id = (int32_t)i; }
    }

    if (sc.root != kNoRoot) {
        RayBox rb = make_raybox(ro, rd);
        int32_t cur = sc.root;
        stack.reset();
        for (;;) {
            if (cur >= 0) {
                const f4 *n = sc.nodes + 4 * (size_t)cur;
                f4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
                float tn0, tn1;
                bool h0 = slab(rb, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best_t, tn0);
                bool h1 = slab(rb, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best_t, tn1);
                if (cnt) cnt->boxes++;
                int32_t c0 = (int32_t)f2u(q3.x), c1 = (int32_t)f2u(q3.y);
                if (h0 && h1) {
                    bool swap = tn1 < tn0;
                    stack.push(swap ? c0 : c1);
                    cur = swap ? c1 : c0;
                    continue;
                }
                if (h0) { cur = c0; continue; }
                if (h1) { cur = c1; continue; }
            } else {
                uint32_t code = (uint32_t)~cur;
                uint32_t first = code >> 3, count = (code & 7u) + 1u;
                for (uint32_t k = 0; k < count; k++) {
                    if (leaf_holds_spheres(sc, first)) {
                        float tn;
                        float t = sphere_t(ro, rd, sc.sph_leaf[first + k - sc.n_tris], tn);
                        int32_t id = (int32_t)sc.sph_orig[first + k - sc.n_tris];
                        if (cnt) cnt->spheres++;
                        if (t > 0.0f && (t < best_t || (t == best_t && id < best_id))) { best_t = t; best_id = id; }
                        continue;
                    }
                    const f4 *tr = sc.tri_isect + kTriRecord * (size_t)(first + k);
                    f4 a = tr[0], b = tr[1], c = tr[2];
                    float t = triangle_t(ro, rd, xyz(a), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x));
                    if (cnt) cnt->tris++;
                    if (!(t > 0.0f)) continue;
                    int32_t id = (int32_t)(sc.tri_orig[first + k] + sc.n_spheres);
                    if (t < best_t || (t == best_t && id < best_id)) {
                        // common.wgsl:202-207 back-face cull on the stored unit normal
                        f3 nrm = mk3(c.y, c.z, c.w);
                        if (dot(nrm, rd) > 0.0f) continue;
                        best_t = t;
                        best_id = id;
                    }
                }
            }
            if (stack.empty()) break;
            cur = stack.pop();
        }
    }
    if (best_id == 0x7fffffff) { out_t = -1.0f; return -1; }
    out_t = best_t;
    return best_id;
}

// Shadow query of raytracer_compute.wgsl:216-219: the light is reached iff
// ray_all misses or its closest t > limit, i.e. iff no valid hit has !(t > limit).
template <class Stack>
RT_HD bool trace_occluded(const SceneView &sc, f3 ro, f3 rd, float limit, Stack &stack, TestCounts *cnt)
{
    if (all_nan(rd)) return false;
    for (uint32_t i = 0; i < sc.n_linear_spheres; i++) {
        float tn;
        float t = sphere_t(ro, rd, sc.spheres[4 * i], tn);
        if (cnt) cnt->spheres++;
        if (t > 0.0f && !(t > limit)) return true;
    }
    if (sc.root == kNoRoot) return false;
    RayBox rb = make_raybox(ro, rd);
    // a NaN limit makes every valid hit an occluder; fminf drops it from the box clip
    int32_t cur = sc.root;
    stack.reset();
    for (;;) {
        if (cur >= 0) {
            const f4 *n = sc.nodes + 4 * (size_t)cur;
            f4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
            float tn0, tn1;
            bool h0 = slab(rb, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, limit, tn0);
            bool h1 = slab(rb, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, limit, tn1);
            if (cnt) cnt->boxes++;
            int32_t c0 = (int32_t)f2u(q3.x), c1 = (int32_t)f2u(q3.y);
            if (h0 && h1) { stack.push(c1); cur = c0; continue; }
            if (h0) { cur = c0; continue; }
            if (h1) { cur = c1; continue; }
        } else {
            uint32_t code = (uint32_t)~cur;
            uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                if (leaf_holds_spheres(sc, first)) {
                    float tn;
                    float t = sphere_t(ro, rd, sc.sph_leaf[first + k - sc.n_tris], tn);
                    if (cnt) cnt->spheres++;
                    if (t > 0.0f && !(t > limit)) return true;
                    continue;
                }
                const f4 *tr = sc.tri_isect + kTriRecord * (size_t)(first + k);
                f4 a = tr[0], b = tr[1], c = tr[2];
                float t = triangle_t(ro, rd, xyz(a), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x));
                if (cnt) cnt->tris++;
                if (!(t > 0.0f) || t > limit) continue;
                f3 nrm = mk3(c.y, c.z, c.w);
                if (dot(nrm, rd) > 0.0f) continue;
                return true;
            }
        }
        if (stack.empty()) break;
        cur = stack.pop();
    }
    return false;
}

// The marker of a triangle in the debug view (common.wgsl:235-240, ray_triangle_normal_intersect,
// which the shader defines and never calls): a sphere of radius 0.1 placed 0.2 along the normal
// from the triangle's centre.  r: the triangle's 96-byte record.  Returns (centre, radius).
RT_HD f4 marker_sphere(const f4 *r)
{
    f3 v0 = xyz(r[0]), v1 = xyz(r[1]), v2 = xyz(r[2]);
    f3 center = divs(add(add(v0, v1), v2), 3.0f);
    f3 normal = normalize(cross(sub(v1, v0), sub(v2, v0)));
    return mk4(add(center, scale(normal, 0.2f)), 0.1f);
}

// Brute-force ray_all exactly as raytracer_compute.wgsl:15-39 over the raw
// 96-byte triangle records (edges and normal recomputed per test).
RT_HD int32_t brute_closest(const SceneView &sc, f3 ro, f3 rd, float &out_t)
{
    float closest_t = -1.0f;
    int32_t closest_id = -1;
    for (uint32_t i = 0; i < sc.n_spheres; i++) {
        float tn;
        float t = sphere_t(ro, rd, sc.spheres[4 * i], tn);
        if (t > 0.0f && (closest_t < 0.0f || t < closest_t)) { closest_t = t; closest_id = (int32_t)i; }
    }
    for (uint32_t i = 0; i < sc.n_tris; i++) {
        const f4 *r = sc.tri_raw + 6 * (size_t)i;
        f3 v0 = xyz(r[0]), v1 = xyz(r[1]), v2 = xyz(r[2]);
        f3 e01 = sub(v1, v0), e02 = sub(v2, v0);
        float t = triangle_t(ro, rd, v0, e01, e02);
        if (!(t < 0.0f)) { // otherwise the miss record of common.wgsl:180-199
            f3 nrm = normalize(cross(e01, e02));
            if (!(dot(nrm, rd) > 0.0f) && t > 0.0f && (closest_t < 0.0f || t < closest_t)) {
                closest_t = t;
                closest_id = (int32_t)(sc.n_spheres + i);
            }
        }
        if (sc.debug_view) { // the triangle's marker is the next candidate, same rule; marker ids follow the triangles'
            float tn;
            float tm = sphere_t(ro, rd, marker_sphere(r), tn);
            if (tm > 0.0f && (closest_t < 0.0f || tm < closest_t)) { closest_t = tm; closest_id = (int32_t)(sc.n_spheres + sc.n_tris + i); }
        }
    }
    out_t = closest_t;
    return closest_id;
}

// HitInfo of a traced id (the fields ray_sphere_intersect / ray_triangle_intersect return).
// LEAF_IDS: triangle ids count leaf positions (the persistent kernel's numbering) instead of
// original indices.
template <bool LEAF_IDS = false>
RT_HD Hit fetch_hit(const SceneView &sc, f3 ro, f3 rd, int32_t id, float t)
{
    Hit h;
    h.t = t;
    if ((uint32_t)id < sc.n_spheres) {
        const f4 *s = sc.spheres + 4 * (size_t)id;
        f4 s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
        // The normal is taken at the parameter of the branch that reported the hit (common.wgsl:155-165):
        // a reported t above 0.01 can only come from the t1 branch, whose parameter is t itself; the
        // t2 branch reports t1 <= 0.01 and needs the roots again.
        float tn = t;
        if (!(t > 0.01f)) (void)sphere_t(ro, rd, s0, tn);
        f3 hit_point = add(ro, scale(rd, tn));
        h.normal = normalize(sub(hit_point, xyz(s0)));
        h.color = xyz(s1);
        h.smoothness = s1.w;
        h.emission = scale(xyz(s2), s2.w);
        h.specularProbability = s3.x;
    } else if ((uint32_t)id >= sc.n_spheres + sc.n_tris) {
        // a marker of the debug view: the Sphere(centre, 0.1, normal, 0, vec3(), 0, 0) of common.wgsl:239
        const f4 *r = sc.tri_raw + 6 * (size_t)((uint32_t)id - sc.n_spheres - sc.n_tris);
        f4 s0 = marker_sphere(r);
        float tn = t;
        if (!(t > 0.01f)) (void)sphere_t(ro, rd, s0, tn);
        f3 hit_point = add(ro, scale(rd, tn));
        h.normal = normalize(sub(hit_point, xyz(s0)));
        h.color = normalize(cross(sub(xyz(r[1]), xyz(r[0])), sub(xyz(r[2]), xyz(r[0]))));
        h.smoothness = 0.0f;
        h.emission = scale(mk3(0.0f, 0.0f, 0.0f), 0.0f);
        h.specularProbability = 0.0f;
    } else {
        const f4 *s = (LEAF_IDS ? sc.tri_shade_leaf : sc.tri_shade) + 3 * (size_t)((uint32_t)id - sc.n_spheres);
        f4 a = s[0], b = s[1], c = s[2];
        h.normal = xyz(a);
        h.smoothness = a.w;
        h.color = xyz(b);
        h.specularProbability = b.w;
        h.emission = xyz(c);
    }
    return h;
}

// ---------------------------------------------------------------------------
// Sampling (raytracer_compute.wgsl:42-107, common.wgsl:107-134,214-225)
// ---------------------------------------------------------------------------

struct LightSample {
    f3 direction, emission;
    float distance, pdf;
};

// The two O(T) scans of the shader collapse to a walk over the emissive
// triangles' running power sums, which the host accumulated in the same order
// with the same f32 additions.
RT_HD LightSample sample_light_triangles(const SceneView &sc, f3 hit_point, uint32_t &state)
{
    LightSample out;
    out.direction = mk3(0, 0, 0); out.emission = mk3(0, 0, 0); out.distance = 0.0f; out.pdf = 0.0f;
    if (sc.n_lights_pass1 == 0u || sc.total_power <= 0.0f) return out;

    float random_power = rand_f(state) * sc.total_power;
    uint32_t sel = sc.n_lights; // trailing zero entry = "var selected_triangle" left at its zero value
    for (uint32_t i = 0; i < sc.n_lights; i++) {
        if (read_uniform(sc.lights + 5 * i + 2).w >= random_power) { sel = i; break; }
    }
    const f4 *L = sc.lights + 5 * (size_t)sel;
    f4 l0 = read_whole(L + 0), l1 = read_whole(L + 1), l2 = L[2], l3 = L[3], l4 = L[4];

    // common.wgsl:214-225
    float r1 = rand_f(state);
    float r2 = rand_f(state);
    float sqrt_r1 = sqrtf(r1);
    float u = 1.0f - sqrt_r1;
    float v = r2 * sqrt_r1;
    float w = 1.0f - u - v;
    f3 light_point = add(add(scale(xyz(l0), u), scale(xyz(l1), v)), scale(xyz(l2), w));

    f3 light_direction = sub(light_point, hit_point);
    float distance = length(light_direction);
    f3 normalized_direction = divs(light_direction, distance);
    float cos_light = -dot(xyz(l3), normalized_direction);
    if (cos_light <= 0.0f) return out;

    float area = l0.w;
    out.pdf = (distance * distance) / (area * cos_light) * l1.w;
    out.direction = normalized_direction;
    out.emission = xyz(l4);
    out.distance = distance;
    return out;
}

// raytracer_compute.wgsl:109-171: defined in the shader and never called there; used by the
// opt-in sphere-light extension (FrameParams::sphere_lights).  The all-zero sphere stands for
// WGSL's zero-initialised `selected_sphere` when u32(rand * count) lands on count.
RT_HD LightSample sample_light_spheres(const SceneView &sc, f3 hit_point, uint32_t &state)
{
    LightSample out;
    out.direction = mk3(0, 0, 0); out.emission = mk3(0, 0, 0); out.distance = 0.0f; out.pdf = 0.0f;
    uint32_t light_count = 0u;
    for (uint32_t i = 0; i < sc.n_spheres; i++)
        if (read_uniform(sc.spheres + 4 * i + 2).w > 0.0f) light_count++;
    if (light_count == 0u) return out;

    uint32_t random_sphere_index = (uint32_t)(rand_f(state) * (float)light_count);
    uint32_t current_index = 0u;
    f4 s0 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    for (uint32_t i = 0; i < sc.n_spheres; i++) {
        if (read_uniform(sc.spheres + 4 * i + 2).w > 0.0f) {
            if (current_index == random_sphere_index) { s0 = sc.spheres[4 * i]; s2 = sc.spheres[4 * i + 2]; break; }
            current_index++;
        }
    }
    f3 to_center = sub(xyz(s0), hit_point);
    float distance_to_center = length(to_center);
    float radius = s0.w;
    if (distance_to_center <= radius) return out;

    f3 center_direction = divs(to_center, distance_to_center);
    bool steep = fabsf(center_direction.y) > 0.99f;
    f3 up = mk3(steep ? 1.0f : 0.0f, steep ? 0.0f : 1.0f, 0.0f);
    f3 right = normalize(cross(up, center_direction));
    f3 forward = cross(center_direction, right);

    float cos_theta_max = sqrtf(1.0f - (radius * radius) / (distance_to_center * distance_to_center));
    float cos_theta = 1.0f - rand_f(state) * (1.0f - cos_theta_max);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * kPi * rand_f(state);
    float sphi, cphi;
    sincos_f(phi, sphi, cphi);
    out.direction = normalize(add(add(scale(center_direction, cos_theta), scale(right, sin_theta * cphi)),
                                  scale(forward, sin_theta * sphi)));
    out.pdf = 1.0f / (2.0f * kPi * (1.0f - cos_theta_max) * (float)light_count);
    out.emission = scale(xyz(s2), s2.w);
    out.distance = distance_to_center - radius;
    return out;
}

RT_HD f3 sample_cosine_hemisphere(f3 normal, uint32_t &state) // common.wgsl:107-134
{
    float r1 = rand_f(state);
    float r2 = rand_f(state);
    float phi = 2.0f * kPi * r1;
    float cos_theta = sqrtf(r2);
    float sin_theta = sqrtf(1.0f - r2);
    float sphi, cphi;
    sincos_f(phi, sphi, cphi);
    f3 local_dir = mk3(cphi * sin_theta, cos_theta, sphi * sin_theta);
    bool steep = fabsf(normal.y) > 0.99f;
    f3 up = mk3(steep ? 1.0f : 0.0f, steep ? 0.0f : 1.0f, 0.0f);
    f3 tangent = normalize(cross(up, normal));
    f3 bitangent = cross(normal, tangent);
    return normalize(add(add(scale(tangent, local_dir.x), scale(normal, local_dir.y)),
                         scale(bitangent, local_dir.z)));
}

RT_HD float power_heuristic(float pdf_a, float pdf_b) // raytracer_compute.wgsl:174-178
{
    float a = pdf_a * pdf_a;
    float b = pdf_b * pdf_b;
    return a / (a + b);
}

RT_HD float brdf_pdf(f3 hit_normal, f3 direction) // raytracer_compute.wgsl:181-187
{
    float cos_theta = dot(hit_normal, direction);
    if (cos_theta <= 0.0f) return 0.0f;
    return cos_theta / kPi;
}

// ---------------------------------------------------------------------------
// Per-pixel pieces of raytracer_compute.wgsl:280-341
// ---------------------------------------------------------------------------

// Seed, jitter and camera ray (raytracer_compute.wgsl:293-329).
RT_HD f3 primary_direction(const FrameParams &fp, int px, int py, uint32_t &state)
{
    float uvx = ((float)px + 0.5f) / fp.res_x;
    float uvy = ((float)py + 0.5f) / fp.res_y;
    uint32_t seed = (uint32_t)py * fp.res_x_u + (uint32_t)px + fp.seed_add;
    state = wang_hash(seed);
    float jx = rand_f(state) - 0.5f;
    float jy = rand_f(state) - 0.5f;
    float offx = jx / fp.res_x;
    float offy = jy / fp.res_y;
    float cx = ((uvx + offx) * 2.0f - 1.0f) * fp.aspect;
    float cy = 1.0f - (uvy + offy) * 2.0f;
    return normalize(add(add(scale(fp.right, cx), scale(fp.up, cy)), scale(fp.forward, fp.focal_length)));
}

// NEE contribution of raytracer_compute.wgsl:220-223 once the light is known to be reached.
RT_HD f3 nee_contribution(const LightSample &ts, f3 hit_normal, float cos_theta, f3 color)
{
    float brdf_pdf_val = brdf_pdf(hit_normal, ts.direction);
    float mis_weight = power_heuristic(ts.pdf, brdf_pdf_val);
    float brdf = cos_theta / kPi;
    f3 c = scale(ts.emission, brdf);
    c = mul(c, color);
    c = scale(c, mis_weight);
    return divs(c, ts.pdf);
}

// Everything of one bounce after the shadow test (raytracer_compute.wgsl:229-274)
// except the addition to `light`: the emission term of :242 is returned in em
// (has_em says whether the shader adds it) so that the caller can add it after
// the light-sample term of :223, in the shader's order.  Returns false when the
// path ends; otherwise ro/rd hold the next ray.
RT_HD bool finish_bounce(const Hit &hit, uint32_t bounce, f3 &ro, f3 &rd, f3 &color, f3 &em, bool &has_em,
                         bool &is_specular_bounce, uint32_t &state)
{
    bool is_primary_ray = (bounce == 0u);
    bool is_emissive = (hit.emission.x + hit.emission.y + hit.emission.z) > 0.01f;
    has_em = (is_primary_ray && is_emissive) || is_specular_bounce;
    em = mk3(0.0f, 0.0f, 0.0f);
    if (has_em) {
        f3 incoming_dir = neg(normalize(rd));
        float brdf_pdf_val = brdf_pdf(hit.normal, incoming_dir);
        float mis_weight = power_heuristic(brdf_pdf_val, 0.001f);
        em = scale(mul(hit.emission, color), mis_weight);
    }

    color = mul(color, hit.color);
    if (color.x + color.y + color.z < 0.01f) return false;

    if (bounce >= 3u) {
        float lum = luminance(color);
        float p = clampf(lum, 0.05f, 0.95f);
        if (rand_f(state) > p) return false;
        color = divs(color, p);
    }

    is_specular_bounce = hit.specularProbability >= rand_f(state);

    f3 diffuseDir = sample_cosine_hemisphere(hit.normal, state);
    f3 specularDir = sub(rd, scale(hit.normal, 2.0f * dot(hit.normal, rd))); // reflect()
    float mt = is_specular_bounce ? hit.smoothness : 0.0f;
    f3 mixed = add(scale(diffuseDir, 1.0f - mt), scale(specularDir, mt)); // mix()
    f3 new_direction = normalize(mixed);
    f3 new_origin = add(add(ro, scale(rd, hit.t)), scale(hit.normal, 0.01f));
    ro = new_origin;
    rd = new_direction;
    return true;
}

// f32 -> f16 -> f32 as the rgba16float store and the sampled read do
// (src/webgpu-renderer.ts:108-112); round to nearest even.
RT_HD uint16_t f32_to_f16_bits(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
#else
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(ax == 0x7f800000u ? (sign | 0x7c00u) : (sign | 0x7e00u | ((ax >> 13) & 0x1ffu)));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x33000001u) return (uint16_t)sign;
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u, shift = 13, hb = 0;
    if (e < -14) shift = (uint32_t)(13 + (-14 - e));
    else { hb = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (hb + q));
#endif
}

RT_HD float f16_bits_to_f32(uint16_t h)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, h);
#else
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; sh++; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    return u2f(x);
#endif
}

// accumulator.wgsl:51-61 for one channel.
RT_HD float accumulate_channel(float stored, float cur, float frame_f)
{
    float weight = 1.0f / (frame_f + 1.0f);
    float s = (frame_f > 0.0f) ? stored : 0.0f;
    return clampf(s * (1.0f - weight) + cur * weight, 0.0f, 1.0f);
}

} // namespace rt
