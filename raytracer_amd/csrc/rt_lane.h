// rt_lane.h -- the path tracer as a per-lane state machine.
//
// The shader's loop nest (raytracer_compute.wgsl:189-278 inside :333-337) is
// unrolled into resumable pieces so that the 64 lanes of a wave need not be at
// the same sample, bounce or kind of ray:
//
//   lane_begin_pixel  seed + camera ray, first path            (:293-336)
//   lane_step_inner   one BVH node visit  } ray_all, for both the closest-hit
//   lane_step_leaf    one BVH leaf visit  } query and the shadow query (:15-39)
//   lane_shade        everything between two ray_all calls: hit record, light
//                     sample and shadow-ray launch (:204-216), or shadow result,
//                     emission, throughput, roulette, next direction (:219-274),
//                     path end / next sample / pixel end (:198-201, :334-337)
//
// Each lane always owns exactly one ray in flight (closest or shadow).  The
// order of the random draws and of every f32 operation per lane is that of the
// shader, so results do not depend on how lanes are scheduled.
#pragma once

#include "rt_core.h"

namespace rt {

enum : int32_t {
    kLaneNeedUnit = 0, // no pixel assigned
    kLaneTraverse = 1, // ray in flight
    kLaneShade = 2,    // ray finished, waiting for lane_shade
    kLaneDead = 3      // no work left
};

constexpr int32_t kNoHit = 0x7fffffff;

struct Lane {
    int32_t mode;
    // pixel / frame
    int32_t px, py;
    uint32_t frame_k;
    uint32_t rng, sample, bounce;
    f3 prim_d, total;
    int32_t primary_id;
    // path
    f3 color, light;
    bool spec;
    // ray in flight
    f3 o, d;
    RayBox rb;
    float tmax;      // closest: best t so far; shadow: the limit distance - 0.1
    int32_t best_id; // closest: best id so far; shadow: kNoHit until an occluder is found
    int32_t cur;     // node link being visited; kNoRoot when the traversal is over
    bool anyhit;
    // bounce suspended while its shadow ray is in flight
    f3 in_o, in_d, nee;
    float hit_t;
    int32_t hit_id;
};

RT_HD bool link_is_inner(int32_t cur) { return cur >= 0 && cur != kNoRoot; }
RT_HD bool link_is_leaf(int32_t cur) { return cur < 0; }

template <class Stack>
RT_HD void lane_pop(Lane &L, Stack &stack)
{
    if (stack.empty()) { L.cur = kNoRoot; L.mode = kLaneShade; }
    else L.cur = stack.pop();
}

// Starts ray_all for the lane's ray (L.o, L.d): spheres are tested here
// (raytracer_compute.wgsl:19-24), triangles by the step functions.
template <class Stack>
RT_HD void lane_launch_ray(const SceneView &sc, Lane &L, Stack &stack, bool anyhit, float limit, TestCounts *cnt)
{
    L.anyhit = anyhit;
    L.tmax = anyhit ? limit : INFINITY;
    L.best_id = kNoHit;
    L.mode = kLaneTraverse;
    stack.reset();
    L.cur = sc.root;
    if (all_nan(L.d)) L.cur = kNoRoot; // every test of the reference fails on a NaN direction
    else {
        for (uint32_t i = 0; i < sc.n_spheres; i++) {
            float tn;
            float t = sphere_t(L.o, L.d, sc.spheres[4 * i], tn);
            if (cnt) cnt->spheres++;
            if (!(t > 0.0f)) continue;
            if (anyhit) {
                if (!(t > limit)) { L.best_id = (int32_t)i; L.cur = kNoRoot; }
            } else if (t < L.tmax || (t == L.tmax && (int32_t)i < L.best_id)) {
                L.tmax = t;
                L.best_id = (int32_t)i;
            }
        }
        L.rb = make_raybox(L.o, L.d);
    }
    if (L.cur == kNoRoot) L.mode = kLaneShade;
}

template <class Stack>
RT_HD void lane_step_inner(const SceneView &sc, Lane &L, Stack &stack, TestCounts *cnt)
{
    const f4 *n = sc.nodes + 4 * (size_t)L.cur;
    f4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
    float tn0, tn1;
    bool h0 = slab(L.rb, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, L.tmax, tn0);
    bool h1 = slab(L.rb, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, L.tmax, tn1);
    if (cnt) cnt->boxes++;
    int32_t c0 = (int32_t)f2u(q3.x), c1 = (int32_t)f2u(q3.y);
    if (h0 && h1) {
        bool swap = tn1 < tn0;
        stack.push(swap ? c0 : c1);
        L.cur = swap ? c1 : c0;
    } else if (h0) L.cur = c0;
    else if (h1) L.cur = c1;
    else lane_pop(L, stack);
}

template <class Stack>
RT_HD void lane_step_leaf(const SceneView &sc, Lane &L, Stack &stack, TestCounts *cnt)
{
    uint32_t code = (uint32_t)~L.cur;
    uint32_t first = code >> 3, count = (code & 7u) + 1u;
    bool stop = false;
    for (uint32_t k = 0; k < count && !stop; k++) {
        const f4 *tr = sc.tri_isect + 3 * (size_t)(first + k);
        f4 a = tr[0], b = tr[1], c = tr[2];
        float t = triangle_t(L.o, L.d, xyz(a), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x));
        if (cnt) cnt->tris++;
        if (!(t > 0.0f)) continue;
        uint32_t orig = f2u(c.y);
        int32_t id = (int32_t)(orig + sc.n_spheres);
        bool candidate = L.anyhit ? !(t > L.tmax) : (t < L.tmax || (t == L.tmax && id < L.best_id));
        if (!candidate) continue;
        f3 nrm = xyz(sc.tri_shade[3 * (size_t)orig]); // common.wgsl:202-207
        if (dot(nrm, L.d) > 0.0f) continue;
        L.best_id = id;
        if (L.anyhit) stop = true;
        else L.tmax = t;
    }
    if (stop) { L.cur = kNoRoot; L.mode = kLaneShade; }
    else lane_pop(L, stack);
}

// Seed, jitter, camera ray and the first path of a pixel (raytracer_compute.wgsl:293-336).
template <class Stack>
RT_HD void lane_begin_pixel(const SceneView &sc, const FrameParams &fp, Lane &L, Stack &stack, uint32_t &n_rays,
                            TestCounts *cnt)
{
    L.prim_d = primary_direction(fp, L.px, L.py, L.rng);
    L.total = mk3(0.0f, 0.0f, 0.0f);
    L.sample = 0;
    L.primary_id = -1;
    L.color = mk3(1.0f, 1.0f, 1.0f);
    L.light = mk3(0.0f, 0.0f, 0.0f);
    L.spec = false;
    L.bounce = 0;
    L.o = fp.cam_pos;
    L.d = L.prim_d;
    n_rays++;
    lane_launch_ray(sc, L, stack, false, 0.0f, cnt);
}

// Consumes the finished ray of a lane in kLaneShade and either launches the
// lane's next ray or finishes the pixel.  Returns true when the pixel is
// finished; out_color is then its frame colour (raytracer_compute.wgsl:337).
template <class Stack>
RT_HD bool lane_shade(const SceneView &sc, const FrameParams &fp, Lane &L, Stack &stack, uint32_t &n_rays,
                      f3 &out_color, TestCounts *cnt)
{
    bool path_end = false, launch = false, launch_anyhit = false, finish = false;
    float launch_limit = 0.0f;
    bool shadow_result = L.anyhit;
    int32_t id;
    float t;
    if (shadow_result) {
        if (L.best_id == kNoHit) L.light = add(L.light, L.nee); // raytracer_compute.wgsl:219-224
        L.o = L.in_o;
        L.d = L.in_d;
        id = L.hit_id;
        t = L.hit_t;
        finish = true;
    } else {
        id = (L.best_id == kNoHit) ? -1 : L.best_id;
        t = L.tmax;
        if (L.bounce == 0u && L.sample == 0u) L.primary_id = id;
        if (id < 0) { // raytracer_compute.wgsl:198-202
            L.light = add(L.light, mul(scale(mk3(1.0f, 1.0f, 1.0f), 0.4f), L.color));
            path_end = true;
        }
    }
    if (!path_end) {
        Hit hit = fetch_hit(sc, L.o, L.d, id, t);
        if (!shadow_result) {
            if (cnt) cnt->bounces++;
            f3 hit_point = add(L.o, scale(L.d, hit.t));
            LightSample ts = sample_light_triangles(sc, hit_point, L.rng);
            finish = true;
            if (ts.pdf > 0.0f) {
                float cos_theta = dot(hit.normal, ts.direction);
                if (cos_theta > 0.0f) {
                    L.nee = nee_contribution(ts, hit.normal, cos_theta, L.color);
                    L.in_o = L.o;
                    L.in_d = L.d;
                    L.hit_t = t;
                    L.hit_id = id;
                    L.o = add(hit_point, scale(hit.normal, 0.01f));
                    L.d = ts.direction;
                    launch = true;
                    launch_anyhit = true;
                    launch_limit = ts.distance - 0.1f;
                    finish = false;
                }
            }
        }
        if (finish) {
            bool alive = finish_bounce(hit, L.bounce, L.o, L.d, L.color, L.light, L.spec, L.rng);
            L.bounce++;
            if (alive && L.bounce < fp.max_bounces) launch = true;
            else path_end = true;
        }
    }
    bool pixel_done = false;
    if (path_end) {
        L.total = add(L.total, L.light);
        L.sample++;
        if (L.sample < fp.spp) { // raytracer_compute.wgsl:334-336: same camera ray, rng continues
            L.color = mk3(1.0f, 1.0f, 1.0f);
            L.light = mk3(0.0f, 0.0f, 0.0f);
            L.spec = false;
            L.bounce = 0;
            L.o = fp.cam_pos;
            L.d = L.prim_d;
            launch = true;
        } else {
            out_color = divs(L.total, fp.spp_f);
            L.mode = kLaneNeedUnit;
            pixel_done = true;
        }
    }
    if (launch) {
        n_rays++;
        lane_launch_ray(sc, L, stack, launch_anyhit, launch_limit, cnt);
    }
    return pixel_done;
}

} // namespace rt
