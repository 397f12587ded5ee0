// rt_lane.h -- the path tracer as a per-lane state machine.
//
// The shader's loop nest (raytracer_compute.wgsl:189-278 inside :333-337) is
// cut into resumable pieces so that the 64 lanes of a wave need not be at the
// same sample, bounce or kind of ray:
//
//   lane_begin_pixel   seed + camera ray, first path                     (:293-336)
//   lane_step_node     one BVH node visit      } ray_all for the closest-hit
//   lane_step_triangle one triangle of a leaf  } and the shadow query   (:15-39)
//   lane_shade_closest after a closest-hit ray: sky on a miss (:198-202); on a
//                      hit the whole bounce at once -- light sample (:211-216)
//                      and emission, throughput, roulette, next direction
//                      (:229-274) -- which launches the shadow ray and parks the
//                      next bounce ray, or launches the next ray directly
//   lane_shadow_done   after a shadow ray: the two additions to `light` in the
//                      shader's order (:223 then :242), then the parked ray,
//                      the next sample or the end of the pixel
//
// Doing the bounce before its shadow ray is traced is legal because nothing in
// :229-274 reads the shadow result; only the additions to `light` depend on it,
// and those are replayed in order by lane_shadow_done.  The random draws per
// lane stay in the shader's order (light pick, triangle point, roulette,
// specular coin, hemisphere).  Each lane always owns exactly one ray in
// flight, so results do not depend on how lanes are scheduled.
#pragma once

#include <type_traits>
#include "rt_core.h"

#ifndef RT_WIDE_BVH
#define RT_WIDE_BVH 1 // the lane state machine traverses the 4-wide tree (0: the binary tree)
#endif

namespace rt {

enum : int32_t {
    kLaneNeedUnit = 0,     // no pixel assigned
    kLaneTraverse = 1,     // ray in flight
    kLaneShadeClosest = 2, // closest-hit ray finished
    kLaneShadowDone = 3,   // shadow ray finished
    kLanePixelDone = 4,    // all samples done; colour waits to be stored
    kLaneDead = 5          // no work left
};

constexpr int32_t kNoHit = 0x7fffffff;
constexpr bool kWideBvh = RT_WIDE_BVH != 0;

// ray_all calls as the reference makes them, and those actually traversed (the
// camera ray of samples 1.. of a pixel-frame is answered from the lane's cache).
struct RayCounts {
    uint32_t calls, traversed;
};

// Per-lane state.  Fields that are touched once per ray or less are packed so that the
// traversal loop, which only needs the ray, keeps them in few registers.
struct Lane {
    int32_t mode;
    // pixel / frame
    uint32_t unit;        // (frame, pixel) unit number handed out by the scheduler
    uint32_t rng;
    uint32_t bits;        // sample (bits 0-11), bounce (12-19), flags (20-24), cost so far (25-31), see below
    // path
    f3 color, light;
    // ray in flight
    f3 o, d;
    RayBox rb;
    float tmax;      // closest: best t so far; shadow: the limit distance - 0.1
    int32_t best_id; // closest: best id so far; shadow: kNoHit until an occluder is found
    int32_t cur;     // node link being visited; kNoRoot when the traversal is over
};

// State a lane touches rarely, kept behind a `Park` policy (put(slot, value) / get(slot) on
// kParkFloats floats) so that the kernel can hold it in LDS instead of registers:
//  - what is set aside while a shadow ray is in flight (slots 0-13): the next bounce ray, the
//    result of its sphere tests (done while the bounce is shaded, when most lanes of the wave are
//    active, rather than when the shadow ray ends, when few are) and the two pending additions to
//    `light`, written by lane_shade_closest, read back by lane_shadow_done;
//  - what lives for a whole pixel-frame (slots 14-21): the camera ray direction, its
//    closest hit (every sample of a frame starts on the same ray, raytracer_compute.wgsl:329-336,
//    so it is traced once) and the sum of the samples' radiance.
// RegPark keeps everything in registers.
//  - with the sphere-light extension, the second shadow ray of a bounce while the first one flies (slots
//    22-28): its direction, its limit distance and its pending addition to `light` (its origin is the first
//    one's).
enum {
    kParkNextO = 0, kParkNextD = 3, kParkNee = 6, kParkEm = 9, kParkNextT = 12, kParkNextId = 13, kParkShadowFloats = 14,
    kParkTotal = 14, kParkPrimT = 17, kParkPrimId = 18, kParkPrimD = 19,
    kParkS2D = 22, kParkS2Limit = 25, kParkS2Nee = 26, kParkSecondShadowFloats = 7, kParkFloats = 29
};

struct RegPark {
    float v[kParkFloats];
    RT_HD void put(int slot, float x) { v[slot] = x; }
    RT_HD float get(int slot) const { return v[slot]; }
};

template <class Park>
RT_HD void park_put3(Park &park, int slot, f3 a)
{
    park.put(slot, a.x);
    park.put(slot + 1, a.y);
    park.put(slot + 2, a.z);
}

template <class Park>
RT_HD f3 park_get3(const Park &park, int slot)
{
    return mk3(park.get(slot), park.get(slot + 1), park.get(slot + 2));
}

constexpr uint32_t kSampleMask = 0xfffu;    // samples per pixel-frame the lane state machine can count: up to kMaxLaneSamples
constexpr uint32_t kMaxLaneSamples = kSampleMask + 1u;
constexpr uint32_t kBounceShift = 12;       // 8 bits
constexpr uint32_t kBitSpec = 1u << 20;     // is_specular_bounce of the path
constexpr uint32_t kBitAnyhit = 1u << 21;   // the ray in flight is a shadow ray
constexpr uint32_t kBitHasEm = 1u << 22;    // L.em is to be added to light
constexpr uint32_t kBitContinue = 1u << 23; // the path goes on after the shadow ray
constexpr uint32_t kBitShadow2 = 1u << 24;  // a second shadow ray of this bounce (sphere-light extension) is set aside
// Bits 25-31: the bounces of the pixel-frame's finished samples, added up and saturating at 127: what the
// unit has cost so far (two rays per bounce), for the scheduler (lane_unit_cost).
constexpr uint32_t kCostShift = 25, kCostMask = 0x7fu << kCostShift;

RT_HD uint32_t lane_sample(const Lane &L) { return L.bits & kSampleMask; }
RT_HD uint32_t lane_bounce(const Lane &L) { return (L.bits >> kBounceShift) & 0xffu; }
RT_HD bool lane_flag(const Lane &L, uint32_t bit) { return (L.bits & bit) != 0u; }
RT_HD uint32_t lane_unit_cost(const Lane &L) { return L.bits >> kCostShift; }
RT_HD void lane_set_flag(Lane &L, uint32_t bit, bool on) { L.bits = on ? (L.bits | bit) : (L.bits & ~bit); }
// primary-hit id of the pixel (-1 on a miss), valid once the camera ray has been shaded
template <class Park>
RT_HD int32_t lane_primary_id(const SceneView &sc, const Park &park)
{
    int32_t id = (int32_t)f2u(park.get(kParkPrimId));
    if (id == kNoHit) return -1;
    if ((uint32_t)id < sc.n_spheres) return id;
    return (int32_t)(sc.tri_orig[(uint32_t)id - sc.n_spheres] + sc.n_spheres); // reported by original index
}

RT_HD bool link_is_inner(int32_t cur) { return cur >= 0 && cur != kNoRoot; }
RT_HD bool link_is_leaf(int32_t cur) { return cur < 0; }

// Whether a stack type asks lane_step_triangle to touch the record its top entry points at
// (Stack::kTouchesTop, with top_in_lds() and top()).
template <class T, class = void>
struct stack_touches_top : std::false_type {};
template <class T>
struct stack_touches_top<T, std::void_t<decltype(T::kTouchesTop)>> : std::integral_constant<bool, T::kTouchesTop> {};

// Whether a stack type says that the lanes traverse the 64-byte nodes (Stack::kQuantNodes).
template <class T, class = void>
struct stack_quant_nodes : std::false_type {};
template <class T>
struct stack_quant_nodes<T, std::void_t<decltype(T::kQuantNodes)>> : std::integral_constant<bool, T::kQuantNodes> {};

// Address of the record a link points at: a 4-wide node (128- or 64-byte form) or the first triangle of a leaf.
// Whether a stack type says that leaves may hold spheres (Stack::kSphereLeaves; yes unless it says otherwise).
template <class T, class = void>
struct stack_sphere_leaves : std::true_type {};
template <class T>
struct stack_sphere_leaves<T, std::void_t<decltype(T::kSphereLeaves)>> : std::integral_constant<bool, T::kSphereLeaves> {};

template <bool QUANT, bool SPHERES = true>
RT_HD const char *link_address(const SceneView &sc, int32_t link)
{
    if (link < 0) {
        uint32_t first = (uint32_t)~link >> 3;
        return SPHERES && leaf_holds_spheres(sc, first) ? (const char *)(sc.sph_leaf + (first - sc.n_tris))
                                                        : (const char *)(sc.tri_isect + kTriRecord * (size_t)first);
    }
    return QUANT ? (const char *)sc.nodes4q + ((size_t)link << 6) : (const char *)sc.nodes4 + ((size_t)link << 7);
}

template <class Stack>
RT_HD void lane_pop(Lane &L, Stack &stack)
{
    if (stack.empty()) { L.cur = kNoRoot; L.mode = lane_flag(L, kBitAnyhit) ? kLaneShadowDone : kLaneShadeClosest; }
    else L.cur = stack.pop();
}

// Sphere part of ray_all (raytracer_compute.wgsl:19-24) for the lane's ray,
// merged into L.tmax / L.best_id under the reference's selection rule.
// TREE_SPHERES: the scene may keep spheres in the tree, the rest (n_linear_spheres) are tested here; without it
// all n_spheres are, and the kernel needs one value less.
template <bool TREE_SPHERES>
RT_HD void lane_test_spheres(const SceneView &sc, Lane &L, TestCounts *cnt)
{
    const uint32_t n_linear = TREE_SPHERES ? sc.n_linear_spheres : sc.n_spheres;
    for (uint32_t i = 0; i < n_linear; i++) {
        f4 s0 = read_uniform(sc.spheres + 4 * i);
        float tn;
        float t = sphere_t(L.o, L.d, s0, tn);
        if (cnt) cnt->spheres++;
        if (!(t > 0.0f)) continue;
        if (lane_flag(L, kBitAnyhit)) {
            if (!(t > L.tmax)) { L.best_id = (int32_t)i; L.cur = kNoRoot; }
        } else if (t < L.tmax || (t == L.tmax && (int32_t)i < L.best_id)) {
            L.tmax = t;
            L.best_id = (int32_t)i;
        }
    }
}

// Starts ray_all for the lane's ray (L.o, L.d).
template <class Stack>
RT_HD void lane_launch_ray(const SceneView &sc, Lane &L, Stack &stack, bool anyhit, float limit, RayCounts &n_rays,
                           TestCounts *cnt)
{
    n_rays.calls++;
    n_rays.traversed++;
    lane_set_flag(L, kBitAnyhit, anyhit);
    L.tmax = anyhit ? limit : INFINITY;
    L.best_id = kNoHit;
    L.mode = kLaneTraverse;
    stack.reset();
    L.cur = kWideBvh ? sc.root4 : sc.root;
    if (all_nan(L.d)) L.cur = kNoRoot; // every test of the reference fails on a NaN direction
    else {
        L.rb = make_raybox(L.o, L.d);
        lane_test_spheres<stack_sphere_leaves<Stack>::value>(sc, L, cnt);
    }
    if (L.cur == kNoRoot) L.mode = anyhit ? kLaneShadowDone : kLaneShadeClosest;
}

// The sphere part of a closest-hit ray_all for a ray that is launched later (the bounce ray set
// aside behind a shadow ray): best t and id over the spheres, (INFINITY, kNoHit) if none.
template <bool TREE_SPHERES>
RT_HD void spheres_closest(const SceneView &sc, f3 o, f3 d, float &best_t, int32_t &best_id, TestCounts *cnt)
{
    best_t = INFINITY;
    best_id = kNoHit;
    if (all_nan(d)) return;
    const uint32_t n_linear = TREE_SPHERES ? sc.n_linear_spheres : sc.n_spheres;
    for (uint32_t i = 0; i < n_linear; i++) {
        float tn;
        float t = sphere_t(o, d, read_uniform(sc.spheres + 4 * i), tn);
        if (cnt) cnt->spheres++;
        if (t > 0.0f && (t < best_t || (t == best_t && (int32_t)i < best_id))) { best_t = t; best_id = (int32_t)i; }
    }
}

// Starts the closest-hit ray_all of the lane's ray (L.o, L.d) whose sphere tests were done before
// (spheres_closest): only the tree is left to traverse.
template <class Stack>
RT_HD void lane_launch_pretested(const SceneView &sc, Lane &L, Stack &stack, float best_t, int32_t best_id, RayCounts &n_rays)
{
    n_rays.calls++;
    n_rays.traversed++;
    lane_set_flag(L, kBitAnyhit, false);
    L.tmax = best_t;
    L.best_id = best_id;
    L.mode = kLaneTraverse;
    stack.reset();
    L.cur = kWideBvh ? sc.root4 : sc.root;
    if (all_nan(L.d)) L.cur = kNoRoot;
    else L.rb = make_raybox(L.o, L.d);
    if (L.cur == kNoRoot) L.mode = kLaneShadeClosest;
}

// One visit of a node of the binary tree (used when RT_WIDE_BVH is 0): the nearer hit child is
// visited next, the other goes on the stack.
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

// a[i] * m + s for the four children of a node.  (Packed v_pk_fma_f32 was measured slower here.)
RT_HD void fma4(const f4 &a, float m, float s, float out[4])
{
    out[0] = __builtin_fmaf(a.x, m, s); out[1] = __builtin_fmaf(a.y, m, s);
    out[2] = __builtin_fmaf(a.z, m, s); out[3] = __builtin_fmaf(a.w, m, s);
}

// The second half of a 4-wide node visit: given the entry distances tn[] and hit flags h[] of the
// four children and their links, the nearest hit child is visited next and the other hit children
// go on the stack.
// LINKS16: the links are 16-bit codes in the low halves of lk (0x7fff marks an unused slot); the
// stack takes the low 16 bits of what is pushed, the link visited next is sign-extended.
template <int LINKS = 0, class Stack>
RT_HD void lane_descend4(Lane &L, Stack &stack, const float tn[4], const bool h[4], f4 lk, TestCounts *cnt)
{
    float t0 = tn[0], t1 = tn[1], t2 = tn[2], t3 = tn[3];
    int32_t c0 = (int32_t)f2u(lk.x), c1 = (int32_t)f2u(lk.y), c2 = (int32_t)f2u(lk.z), c3 = (int32_t)f2u(lk.w);
    constexpr int32_t unused = LINKS == 16 ? 0x7fff : kNoRoot;
    // children 0 and 1 always exist; an unused slot is never a hit whatever its box says
    bool h0 = h[0], h1 = h[1], h2 = h[2] & (c2 != unused), h3 = h[3] & (c3 != unused);
    if (cnt) cnt->boxes += 2; // counted in box pairs
    // nearest hit child
    float tb = INFINITY;
    int32_t next = kNoRoot;
    int which = -1;
    if (h0) { tb = t0; next = c0; which = 0; }
    if (h1 && (which < 0 || t1 < tb)) { tb = t1; next = c1; which = 1; }
    if (h2 && (which < 0 || t2 < tb)) { tb = t2; next = c2; which = 2; }
    if (h3 && (which < 0 || t3 < tb)) { tb = t3; next = c3; which = 3; }
    if (which < 0) { lane_pop(L, stack); return; }
    if (h0 && which != 0) stack.push(c0);
    if (h1 && which != 1) stack.push(c1);
    if (h2 && which != 2) stack.push(c2);
    if (h3 && which != 3) stack.push(c3);
    L.cur = LINKS == 16 ? (int32_t)(int16_t)next : next;
}

// One visit of a 4-wide node: four slab tests, the nearest hit child is visited
// next and the other hit children go on the stack (unsorted: the order only
// affects how much is culled later, never the result).
// The sign of the ray direction says which plane of each slab is entered first, so
// the near and far planes are fetched as such (a node stores lo.x[4], lo.y[4],
// lo.z[4], hi.x[4], hi.y[4], hi.z[4]) and no per-slab min/max is needed; for lo <= hi
// this gives the same distances as slab().
// bits 0-7 of lo, bits 0-7 of hi as bits 8-15: a 16-bit link from the low bytes of two plane values
RT_HD float link16_of(float lo, float hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return u2f(__builtin_amdgcn_perm(f2u(hi), f2u(lo), 0x0c0c0400u));
#else
    return u2f(((f2u(hi) & 0xffu) << 8) | (f2u(lo) & 0xffu));
#endif
}

// PACKED: the node's links are read from the low bytes of its x and y planes (BuiltScene::links_packed)
// instead of from the link vector: six loads per visit instead of seven.
template <int PACKED = 0, class Stack>
RT_HD void lane_visit4(const char *node, uint32_t at, Lane &L, Stack &stack, TestCounts *cnt)
{
    // byte offset of the near plane vector: lo (0/16/32) for a non-negative direction, hi (48/64/80) otherwise
    uint32_t nx = at | ((uint32_t)((int32_t)f2u(L.rb.inv.x) >> 31) & 48u);
    uint32_t ny = at | (16u ^ ((uint32_t)((int32_t)f2u(L.rb.inv.y) >> 31) & 80u));
    uint32_t nz = at | (32u ^ ((uint32_t)((int32_t)f2u(L.rb.inv.z) >> 31) & 112u));
    f4 nearx = *(const f4 *)(node + nx), farx = *(const f4 *)(node + (nx ^ 48u));
    f4 neary = *(const f4 *)(node + ny), fary = *(const f4 *)(node + (ny ^ 80u));
    f4 nearz = *(const f4 *)(node + nz), farz = *(const f4 *)(node + (nz ^ 112u));
    f4 lk;
    if (PACKED == 16) lk = mk4(link16_of(nearx.x, neary.x), link16_of(nearx.y, neary.y), link16_of(nearx.z, neary.z), link16_of(nearx.w, neary.w));
    else lk = *(const f4 *)(node + (at | 96u));
    float ax[4], ay[4], az[4], bx[4], by[4], bz[4], tn[4];
    fma4(nearx, L.rb.inv.x, L.rb.ood.x, ax); fma4(farx, L.rb.inv.x, L.rb.ood.x, bx);
    fma4(neary, L.rb.inv.y, L.rb.ood.y, ay); fma4(fary, L.rb.inv.y, L.rb.ood.y, by);
    fma4(nearz, L.rb.inv.z, L.rb.ood.z, az); fma4(farz, L.rb.inv.z, L.rb.ood.z, bz);
    bool h[4];
    for (int i = 0; i < 4; i++) {
        tn[i] = fmaxf(fmaxf(ax[i], ay[i]), fmaxf(az[i], 0.0f));
        float tf = fminf(fminf(bx[i], by[i]), fminf(bz[i], L.tmax));
        h[i] = tn[i] <= tf;
    }
    lane_descend4<PACKED>(L, stack, tn, h, lk, cnt);
}

// Byte i of u as a float (one v_cvt_f32_ubyte instruction each).
RT_HD void bytes4(uint32_t u, float out[4])
{
    out[0] = (float)(u & 0xffu); out[1] = (float)((u >> 8) & 0xffu);
    out[2] = (float)((u >> 16) & 0xffu); out[3] = (float)(u >> 24);
}

// One visit of a 4-wide node in its 64-byte form (SceneView::nodes4q): a child plane is origin + q * step with
// an 8-bit q, so its slab distance (origin + q * step) * inv + ood is one FMA on q: q * (step * inv) +
// (origin * inv + ood), the two constants made once per axis and visit.  Four loads (three of 16 bytes, one of
// 8) instead of seven, half the bytes; the boxes are the 128-byte form's moved outwards by less than a step
// (1/255 of the node's extent), and the folded arithmetic moves a distance by a few 1e-7 of the magnitudes
// involved, which the padding of the triangle boxes covers like the rounding of the unfolded form.
template <class Stack>
RT_HD void lane_visit4q(const char *node, Lane &L, Stack &stack, TestCounts *cnt)
{
    f4 h0 = read_whole((const f4 *)node), h1 = read_whole((const f4 *)(node + 16)), h2 = read_whole((const f4 *)(node + 32));
    float l2 = *(const float *)(node + 48), l3 = *(const float *)(node + 52);
    uint32_t ex = f2u(h0.w);
    float sx = u2f((ex & 0xffu) << 23), sy = u2f(((ex >> 8) & 0xffu) << 23), sz = u2f(((ex >> 16) & 0xffu) << 23);
    float ax = sx * L.rb.inv.x, ay = sy * L.rb.inv.y, az = sz * L.rb.inv.z;
    float bx = __builtin_fmaf(h0.x, L.rb.inv.x, L.rb.ood.x), by = __builtin_fmaf(h0.y, L.rb.inv.y, L.rb.ood.y),
          bz = __builtin_fmaf(h0.z, L.rb.inv.z, L.rb.ood.z);
    // the plane entered first is the low one for a non-negative direction, the high one otherwise
    bool nx = (int32_t)f2u(L.rb.inv.x) < 0, ny = (int32_t)f2u(L.rb.inv.y) < 0, nz = (int32_t)f2u(L.rb.inv.z) < 0;
    uint32_t lox = f2u(h1.x), loy = f2u(h1.y), loz = f2u(h1.z), hix = f2u(h1.w), hiy = f2u(h2.x), hiz = f2u(h2.y);
    float qnx[4], qny[4], qnz[4], qfx[4], qfy[4], qfz[4];
    bytes4(nx ? hix : lox, qnx); bytes4(nx ? lox : hix, qfx);
    bytes4(ny ? hiy : loy, qny); bytes4(ny ? loy : hiy, qfy);
    bytes4(nz ? hiz : loz, qnz); bytes4(nz ? loz : hiz, qfz);
    float tn[4];
    bool h[4];
    for (int i = 0; i < 4; i++) {
        float nearx = __builtin_fmaf(qnx[i], ax, bx), neary = __builtin_fmaf(qny[i], ay, by), nearz = __builtin_fmaf(qnz[i], az, bz);
        float farx = __builtin_fmaf(qfx[i], ax, bx), fary = __builtin_fmaf(qfy[i], ay, by), farz = __builtin_fmaf(qfz[i], az, bz);
        tn[i] = fmaxf(fmaxf(nearx, neary), fmaxf(nearz, 0.0f));
        float tf = fminf(fminf(farx, fary), fminf(farz, L.tmax));
        h[i] = tn[i] <= tf;
    }
    lane_descend4<0>(L, stack, tn, h, mk4(h2.z, h2.w, l2, l3), cnt);
}

// One visit of the 4-wide node L.cur.  PACKED 64 selects the 64-byte nodes.
template <int PACKED = 0, class Stack>
RT_HD void lane_step_inner4(const SceneView &sc, Lane &L, Stack &stack, TestCounts *cnt)
{
    if (PACKED == 64) lane_visit4q((const char *)sc.nodes4q + ((size_t)(uint32_t)L.cur << 6), L, stack, cnt);
    else lane_visit4<PACKED>((const char *)sc.nodes4, (uint32_t)L.cur << 7, L, stack, cnt);
}

// The node visit of whichever tree the lanes traverse.
template <int PACKED = 0, class Stack>
RT_HD void lane_step_node(const SceneView &sc, Lane &L, Stack &stack, TestCounts *cnt)
{
    if (kWideBvh) lane_step_inner4<PACKED>(sc, L, stack, cnt);
    else lane_step_inner(sc, L, stack, cnt);
}

// The reference's scan keeps the first of several hits at the same t, i.e. the one with the lowest
// id in its order: spheres, then triangles by original index (raytracer_compute.wgsl:19-36).  id is
// a triangle by leaf position; best is the lane's best id so far (kNoHit, a sphere, or a triangle by
// leaf position).  Only reached on an exact tie in t.
RT_HD bool leaf_id_precedes(const SceneView &sc, int32_t id, int32_t best)
{
    if (best == kNoHit) return true;
    if ((uint32_t)best < sc.n_spheres) return false;
    return sc.tri_orig[(uint32_t)id - sc.n_spheres] < sc.tri_orig[(uint32_t)best - sc.n_spheres];
}

// Tests the first triangle of the leaf L.cur and shrinks the leaf by it.
template <class Stack>
RT_HD void lane_step_triangle(const SceneView &sc, Lane &L, Stack &stack, TestCounts *cnt)
{
    uint32_t code = (uint32_t)~L.cur;
    uint32_t first = code >> 3, left = code & 7u; // triangles after this one
    if (stack_sphere_leaves<Stack>::value && leaf_holds_spheres(sc, first)) {
        // a sphere of a leaf of spheres: the sphere part of ray_all (raytracer_compute.wgsl:19-24) for this one,
        // as lane_test_spheres does it for spheres outside the tree; ids are original indices, all below any
        // triangle's, so on a tie in t the lower number wins whatever the other is
        uint32_t k = first - sc.n_tris;
        float tn;
        float t = sphere_t(L.o, L.d, read_whole(sc.sph_leaf + k), tn);
        int32_t id = (int32_t)sc.sph_orig[k];
        if (cnt) cnt->spheres++;
        bool occluder = false;
        if (t > 0.0f) {
            if (lane_flag(L, kBitAnyhit)) {
                if (!(t > L.tmax)) { L.best_id = id; occluder = true; }
            } else if (t < L.tmax || (t == L.tmax && id < L.best_id)) {
                L.tmax = t;
                L.best_id = id;
            }
        }
        if (occluder) { L.cur = kNoRoot; L.mode = kLaneShadowDone; }
        else if (left != 0u) L.cur = (int32_t)~(((first + 1u) << 3) | (left - 1u));
        else lane_pop(L, stack);
        return;
    }
    const f4 *tr = sc.tri_isect + kTriRecord * (size_t)first;
    f4 a = read_whole(tr + 0), b = read_whole(tr + 1), c = read_whole(tr + 2); // three 16-byte loads, not 12 + 16 + 16 + 4
    float touched = 0.0f;
    if constexpr (stack_touches_top<Stack>::value) {
        // what the lane visits after this leaf is known already: start bringing it into the caches
        if (left == 0u && stack.top_in_lds()) touched = *(const float *)link_address<stack_quant_nodes<Stack>::value, stack_sphere_leaves<Stack>::value>(sc, stack.top());
    }
    float t = triangle_t(L.o, L.d, xyz(a), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x));
    if (cnt) cnt->tris++;
    bool stop = false;
    if (t > 0.0f) {
        int32_t id = (int32_t)(first + sc.n_spheres); // by leaf position, see SceneView::tri_orig
        bool anyhit = lane_flag(L, kBitAnyhit);
        bool candidate = anyhit ? !(t > L.tmax) : (t < L.tmax || (t == L.tmax && leaf_id_precedes(sc, id, L.best_id)));
        if (candidate) {
            f3 nrm = mk3(c.y, c.z, c.w); // common.wgsl:202-207
            if (!(dot(nrm, L.d) > 0.0f)) {
                L.best_id = id;
                if (anyhit) stop = true;
                else L.tmax = t;
            }
        }
    }
    (void)touched;
    if constexpr (stack_touches_top<Stack>::value) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(touched)); // keeps the load
#else
        (void)touched;
#endif
    }
    if (stop) { L.cur = kNoRoot; L.mode = kLaneShadowDone; }
    else if (left != 0u) L.cur = (int32_t)~(((first + 1u) << 3) | (left - 1u));
    else lane_pop(L, stack);
}

// A path has ended with radiance L.light (raytracer_compute.wgsl:335-336): the
// next sample starts on the same camera ray, whose ray_all result is already
// known, so the lane goes straight back to kLaneShadeClosest; after the last
// sample the pixel is finished.
template <class Park>
RT_HD void lane_end_path(const FrameParams &fp, Lane &L, Park &park, RayCounts &n_rays)
{
    f3 total = add(park_get3(park, kParkTotal), L.light);
    uint32_t sample = lane_sample(L) + 1u;
    uint32_t cost = lane_unit_cost(L) + lane_bounce(L);
    cost = (cost > 127u ? 127u : cost) << kCostShift;
    if (sample < fp.spp && (int32_t)f2u(park.get(kParkPrimId)) == kNoHit) {
        // The camera ray missed everything, and every remaining sample starts on that same ray: each
        // is ray_all -> miss -> light = 0.4 * color with color = 1 (raytracer_compute.wgsl:198-202),
        // no random draw involved.  Add them here, one by one as the shader's sample loop does,
        // instead of sending the lane through the shading phase once per sample.
        f3 sky = add(mk3(0.0f, 0.0f, 0.0f), mul(scale(mk3(1.0f, 1.0f, 1.0f), 0.4f), mk3(1.0f, 1.0f, 1.0f)));
        for (; sample < fp.spp; sample++) {
            total = add(total, sky);
            n_rays.calls++;
        }
    }
    park_put3(park, kParkTotal, total);
    if (sample < fp.spp) {
        L.bits = sample | cost; // bounce 0, no flags: not specular, closest-hit result
        L.color = mk3(1.0f, 1.0f, 1.0f);
        L.light = mk3(0.0f, 0.0f, 0.0f);
        L.o = fp.cam_pos;
        L.d = park_get3(park, kParkPrimD);
        L.tmax = park.get(kParkPrimT);
        L.best_id = (int32_t)f2u(park.get(kParkPrimId));
        L.cur = kNoRoot;
        L.mode = kLaneShadeClosest;
        n_rays.calls++;
    } else {
        L.bits = (L.bits & ~kCostMask) | cost;
        L.mode = kLanePixelDone;
    }
}

// Seed, jitter, camera ray and the first path of a pixel (raytracer_compute.wgsl:293-336).
template <class Stack, class Park>
RT_HD void lane_begin_pixel(const SceneView &sc, const FrameParams &fp, Lane &L, int px, int py, Stack &stack,
                            Park &park, RayCounts &n_rays, TestCounts *cnt)
{
    L.d = primary_direction(fp, px, py, L.rng);
    park_put3(park, kParkPrimD, L.d);
    park_put3(park, kParkTotal, mk3(0.0f, 0.0f, 0.0f));
    park.put(kParkPrimId, u2f((uint32_t)kNoHit));
    L.bits = 0u;
    L.color = mk3(1.0f, 1.0f, 1.0f);
    L.light = mk3(0.0f, 0.0f, 0.0f);
    L.o = fp.cam_pos;
    lane_launch_ray(sc, L, stack, false, 0.0f, n_rays, cnt);
}

// Frame colour of a lane in kLanePixelDone (raytracer_compute.wgsl:337).
template <class Park>
RT_HD f3 lane_pixel_color(const FrameParams &fp, const Park &park) { return divs(park_get3(park, kParkTotal), fp.spp_f); }

// A pixel of a frame whose bounce loop never runs (max_bounces 0): zero radiance, no rays
// (raytracer_compute.wgsl:196).
template <class Park>
RT_HD void lane_empty_pixel(Lane &L, Park &park)
{
    park_put3(park, kParkTotal, mk3(0.0f, 0.0f, 0.0f));
    park.put(kParkPrimId, u2f((uint32_t)kNoHit));
    L.bits = 0u;
    L.mode = kLanePixelDone;
}

// Lane in kLaneShadeClosest: consumes the closest-hit result.
// SL: the opt-in sphere-light extension (FrameParams::sphere_lights, rt_pixel.h): after the triangle-light
// sample the emissive spheres are sampled too, which can give the bounce a second shadow ray; it is set
// aside while the first one is traced.
template <bool SL = false, class Stack, class Park>
RT_HD void lane_shade_closest(const SceneView &sc, const FrameParams &fp, Lane &L, Stack &stack, Park &park,
                              RayCounts &n_rays, TestCounts *cnt)
{
    int32_t id = (L.best_id == kNoHit) ? -1 : L.best_id;
    float t = L.tmax;
    uint32_t bounce = lane_bounce(L);
    if ((L.bits & ~kCostMask) == 0u) { // first bounce of the first sample: remember the camera ray's hit
        park.put(kParkPrimT, L.tmax);
        park.put(kParkPrimId, u2f((uint32_t)L.best_id));
    }
    bool end_path = false, launch = false, launch_anyhit = false;
    float launch_limit = 0.0f;
    if (id < 0) { // raytracer_compute.wgsl:198-202
        L.light = add(L.light, mul(scale(mk3(1.0f, 1.0f, 1.0f), 0.4f), L.color));
        end_path = true;
    } else {
        if (cnt) cnt->bounces++;
        Hit hit = fetch_hit<true>(sc, L.o, L.d, id, t);
        f3 hit_point = add(L.o, scale(L.d, hit.t));
        LightSample ts = sample_light_triangles(sc, hit_point, L.rng);
        bool shadow = false, shadow2 = false;
        if (ts.pdf > 0.0f) {
            float cos_theta = dot(hit.normal, ts.direction);
            if (cos_theta > 0.0f) {
                park_put3(park, kParkNee, nee_contribution(ts, hit.normal, cos_theta, L.color));
                shadow = true;
            }
        }
        LightSample ss;
        if (SL) { // raytracer_compute.wgsl:226, wired like :211-225
            ss = sample_light_spheres(sc, hit_point, L.rng);
            if (ss.pdf > 0.0f) {
                float cos_theta = dot(hit.normal, ss.direction);
                if (cos_theta > 0.0f) {
                    // the only shadow ray of the bounce takes the first one's place; otherwise it waits its turn
                    park_put3(park, shadow ? kParkS2Nee : kParkNee, nee_contribution(ss, hit.normal, cos_theta, L.color));
                    shadow2 = true;
                }
            }
        }
        f3 no = L.o, nd = L.d, em;
        bool has_em, spec = lane_flag(L, kBitSpec);
        bool alive = finish_bounce(hit, bounce, no, nd, L.color, em, has_em, spec, L.rng);
        bounce++;
        bool cont = alive && bounce < fp.max_bounces;
        L.bits = (L.bits & (kSampleMask | kCostMask)) | (bounce << kBounceShift) | (spec ? kBitSpec : 0u) | (has_em ? kBitHasEm : 0u) |
                 (cont ? kBitContinue : 0u) | (SL && shadow && shadow2 ? kBitShadow2 : 0u);
        if (shadow || (SL && shadow2)) {
            park_put3(park, kParkNextO, no);
            park_put3(park, kParkNextD, nd);
            park_put3(park, kParkEm, em);
            if (cont) {
                float nt;
                int32_t nid;
                spheres_closest<stack_sphere_leaves<Stack>::value>(sc, no, nd, nt, nid, cnt);
                park.put(kParkNextT, nt);
                park.put(kParkNextId, u2f((uint32_t)nid));
            }
            L.o = add(hit_point, scale(hit.normal, 0.01f));
            L.d = ts.direction;
            launch = true;
            launch_anyhit = true;
            launch_limit = ts.distance - 0.1f;
            if (SL && shadow2) {
                if (shadow) {
                    park_put3(park, kParkS2D, ss.direction);
                    park.put(kParkS2Limit, ss.distance - 0.1f);
                } else {
                    L.d = ss.direction;
                    launch_limit = ss.distance - 0.1f;
                }
            }
        } else {
            if (has_em) L.light = add(L.light, em);
            if (cont) { L.o = no; L.d = nd; launch = true; }
            else end_path = true;
        }
    }
    if (launch) lane_launch_ray(sc, L, stack, launch_anyhit, launch_limit, n_rays, cnt);
    if (end_path) lane_end_path(fp, L, park, n_rays);
}

// Lane in kLaneShadowDone: consumes the shadow result (raytracer_compute.wgsl:219-224, :242).
template <bool SL = false, class Stack, class Park>
RT_HD void lane_shadow_done(const SceneView &sc, const FrameParams &fp, Lane &L, Stack &stack, Park &park,
                            RayCounts &n_rays, TestCounts *cnt)
{
    if (L.best_id == kNoHit) L.light = add(L.light, park_get3(park, kParkNee));
    if (SL && lane_flag(L, kBitShadow2)) {
        // the bounce's second shadow ray, from the same origin (L.o is untouched by the traversal)
        lane_set_flag(L, kBitShadow2, false);
        L.d = park_get3(park, kParkS2D);
        park_put3(park, kParkNee, park_get3(park, kParkS2Nee));
        lane_launch_ray(sc, L, stack, true, park.get(kParkS2Limit), n_rays, cnt);
        return;
    }
    if (lane_flag(L, kBitHasEm)) L.light = add(L.light, park_get3(park, kParkEm));
    if (lane_flag(L, kBitContinue)) {
        L.o = park_get3(park, kParkNextO);
        L.d = park_get3(park, kParkNextD);
        lane_launch_pretested(sc, L, stack, park.get(kParkNextT), (int32_t)f2u(park.get(kParkNextId)), n_rays);
    } else {
        lane_end_path(fp, L, park, n_rays);
    }
}

} // namespace rt
