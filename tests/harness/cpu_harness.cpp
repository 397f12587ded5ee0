// Host build of the renderer's shared core (rt_core.h / rt_pixel.h /
// rt_scene.cpp) so that traversal, shading and scene preparation can be checked
// against the oracle in a container without a GPU.  Test infrastructure only:
// the shipped library has no CPU rendering path.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../raytracer_amd/csrc/rt_pixel.h"
#include "../../raytracer_amd/csrc/rt_scene.h"

namespace {
static int g_link_mode = 0; // har_set_link_mode
struct HostStack {
    int32_t data[128];
    int sp = 0;
    // 16-bit mode: entries are two bytes wide as in the kernels for small scenes (a pop sign-extends)
    void push(int32_t v) { data[sp++] = g_link_mode == 16 ? (int32_t)(int16_t)v : v; if (sp > high) high = sp; }
    int32_t pop() { return data[--sp]; }
    bool empty() const { return sp == 0; }
    void reset() { sp = 0; }
    int high = 0;
};
} // namespace

static int g_sphere_lights = 0;
// the opt-in sphere-light extension for the following har_render_frame calls
extern "C" void har_set_sphere_lights(int enable) { g_sphere_lights = enable; }

// Checks the link packing of the 4-wide nodes against an unpacked build of the same scene: every
// plane of a packed node lies outwards of (or on) the unpacked one by at most 512 ulp, the low bytes
// of the x and y planes spell the 16-bit code of the link vector's entry, and the link vector itself
// is unchanged.  Returns the number of violations, or -1 if the scene's links do not fit 16 bits.
extern "C" int har_check_link_packing(const void *spheres, uint32_t n_spheres, const void *tris, uint32_t n_tris)
{
    rt::BuiltScene a, b;
    rt::BuildOptions opt;
    rt::build_scene(spheres, n_spheres, tris, n_tris, opt, a);
    opt.pack_links = false;
    rt::build_scene(spheres, n_spheres, tris, n_tris, opt, b);
    if (a.links_packed != 16) return -1;
    if (a.nodes4.size() != b.nodes4.size()) return 1 << 30;
    auto ordered = [](float f) { // integer that orders like the float
        int32_t i;
        std::memcpy(&i, &f, 4);
        return i < 0 ? (int64_t)INT32_MIN - i : (int64_t)i;
    };
    int bad = 0;
    for (size_t n = 0; n < a.nodes4.size() / 8; n++) {
        const float *pa = &a.nodes4[8 * n].x, *pb = &b.nodes4[8 * n].x;
        for (int k = 0; k < 4; k++) {
            uint32_t la, lb;
            std::memcpy(&la, pa + 24 + k, 4);
            std::memcpy(&lb, pb + 24 + k, 4);
            if (la != lb) bad++;
            uint32_t code = (int32_t)lb == rt::kNoRoot ? 0x7fffu : (lb & 0xffffu);
            for (int plane = 0; plane < 6; plane++) {
                float va = pa[4 * plane + k], vb = pb[4 * plane + k];
                int64_t d = ordered(va) - ordered(vb); // > 0: moved up
                bool lo = plane < 3;
                if (plane == 2 || plane == 5) { if (d != 0) bad++; continue; } // z planes carry nothing
                if (lo ? (d > 0 || d < -512) : (d < 0 || d > 512)) bad++;
                uint32_t bits;
                std::memcpy(&bits, &va, 4);
                uint32_t want = (plane % 3 == 0) ? (code & 0xffu) : (code >> 8);
                if ((bits & 0xffu) != want) bad++;
            }
        }
    }
    return bad;
}

// Checks the 64-byte nodes against the 128-byte ones of the same scene: same links, and every child plane,
// decoded exactly (origin + q * 2^e in double precision), lies outwards of (or on) the 128-byte node's plane by
// less than one step.  Returns the number of violations.
extern "C" int har_check_quant_nodes(const void *spheres, uint32_t n_spheres, const void *tris, uint32_t n_tris)
{
    rt::BuiltScene b;
    rt::BuildOptions opt;
    opt.pack_links = false;
    opt.quant_nodes = true;
    rt::build_scene(spheres, n_spheres, tris, n_tris, opt, b);
    if (b.nodes4q.size() * 2 != b.nodes4.size()) return 1 << 30;
    int bad = 0;
    for (size_t n = 0; n < b.nodes4.size() / 8; n++) {
        const float *full = &b.nodes4[8 * n].x;
        const rt::f4 *q = &b.nodes4q[4 * n];
        uint32_t ex = rt::f2u(q[0].w);
        const float origin[3] = {q[0].x, q[0].y, q[0].z};
        const uint32_t planes[6] = {rt::f2u(q[1].x), rt::f2u(q[1].y), rt::f2u(q[1].z), rt::f2u(q[1].w), rt::f2u(q[2].x), rt::f2u(q[2].y)};
        const uint32_t links[4] = {rt::f2u(q[2].z), rt::f2u(q[2].w), rt::f2u(q[3].x), rt::f2u(q[3].y)};
        for (int k = 0; k < 4; k++) {
            uint32_t lf;
            std::memcpy(&lf, full + 24 + k, 4);
            if (lf != links[k]) bad++;
            if ((int32_t)lf == rt::kNoRoot) continue;
            for (int r = 0; r < 6; r++) {
                int a = r % 3;
                int e = (int)((ex >> (8 * a)) & 0xffu) - 127;
                double step = std::ldexp(1.0, e), plane = (double)origin[a] + (double)((planes[r] >> (8 * k)) & 0xffu) * step;
                double want = full[4 * r + k];
                if (r < 3 ? (plane > want || plane < want - step) : (plane < want || plane > want + step)) bad++;
            }
        }
    }
    return bad;
}

// How the lane emulation reads a node's child links: 0 = from the link vector; -1 = as the kernels do
// for the scene at hand (16-bit links packed into the low bytes of the node planes when the builder
// says they fit, otherwise the link vector); 64 = every scene through the 64-byte nodes (the kernels use them
// for scenes whose links do not fit 16 bits).
extern "C" void har_set_link_mode(int mode) { g_link_mode = mode; }

extern "C" int har_render_frame(const void *uniforms80, const void *spheres, uint32_t n_spheres,
                                const void *tris, uint32_t n_tris, uint32_t max_bounces, int brute,
                                uint32_t max_leaf, float *accum, uint16_t *frame_f16, int32_t *ids,
                                uint64_t *counters, int n_threads)
{
    rt::BuiltScene bs;
    rt::BuildOptions opt;
    if (max_leaf) opt.max_leaf = max_leaf;
    rt::build_scene(spheres, n_spheres, tris, n_tris, opt, bs);
    rt::FrameParams fp;
    std::string err;
    if (!rt::make_frame_params(uniforms80, max_bounces, fp, err)) return -1;
    fp.sphere_lights = g_sphere_lights ? 1u : 0u;
    rt::SceneView sc{};
    sc.nodes = bs.nodes.data(); sc.nodes4 = bs.nodes4.data(); sc.nodes4q = bs.nodes4q.data(); sc.root4 = bs.root4; sc.tri_isect = bs.tri_isect.data(); sc.tri_shade = bs.tri_shade.data(); sc.tri_shade_leaf = bs.tri_shade_leaf.data(); sc.tri_orig = bs.tri_orig.data();
    sc.tri_raw = bs.tri_raw.data(); sc.spheres = bs.spheres.data(); sc.lights = bs.lights.data(); sc.sph_leaf = bs.sph_leaf.data(); sc.sph_orig = bs.sph_orig.data(); sc.n_linear_spheres = bs.n_spheres - bs.n_tree_spheres;
    sc.n_spheres = bs.n_spheres; sc.n_tris = bs.n_tris; sc.n_lights = bs.n_lights;
    sc.n_lights_pass1 = bs.n_lights_pass1; sc.total_power = bs.total_power; sc.root = bs.root;

    const int W = fp.width, H = fp.height;
    std::atomic<int> next_row{0};
    std::atomic<uint64_t> rays{0}, boxes{0}, tris_n{0}, sph{0}, high{0};
    auto worker = [&]() {
        HostStack st;
        uint64_t local = 0;
        rt::TestCounts cnt = {0, 0, 0, 0};
        uint64_t b = 0, t = 0, s = 0;
        for (;;) {
            int y = next_row.fetch_add(1);
            if (y >= H) break;
            for (int x = 0; x < W; x++) {
                uint32_t n = 0;
                int32_t pid;
                rt::f3 c = brute ? rt::render_pixel<true>(sc, fp, x, y, st, n, pid, &cnt)
                                 : rt::render_pixel<false>(sc, fp, x, y, st, n, pid, &cnt);
                local += n;
                b += cnt.boxes; t += cnt.tris; s += cnt.spheres;
                cnt = {0, 0, 0, 0};
                size_t p = (size_t)y * W + x, plane = (size_t)W * H;
                uint16_t h[4] = {rt::f32_to_f16_bits(c.x), rt::f32_to_f16_bits(c.y), rt::f32_to_f16_bits(c.z), 0x3c00};
                if (frame_f16) std::memcpy(frame_f16 + 4 * p, h, 8);
                if (ids) ids[p] = pid;
                for (int ch = 0; ch < 3; ch++)
                    accum[ch * plane + p] = rt::accumulate_channel(accum[ch * plane + p], rt::f16_bits_to_f32(h[ch]), fp.frame_f);
            }
        }
        rays += local; boxes += b; tris_n += t; sph += s;
        uint64_t hv = (uint64_t)st.high, cur = high.load();
        while (hv > cur && !high.compare_exchange_weak(cur, hv)) {}
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; i++) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    if (counters) {
        counters[0] += rays; counters[1] += boxes; counters[2] += tris_n; counters[3] += sph;
        counters[4] = bs.nodes.size() / 4; counters[5] = bs.n_leaves; counters[6] = bs.max_depth; counters[7] = high;
    }
    return 0;
}

// Debug aid: follows the brute-force path of one pixel and reports every ray for
// which the BVH queries disagree with the brute-force ones.
#include <cstdio>
extern "C" int har_debug_pixel(const void *uniforms80, const void *spheres, uint32_t n_spheres, const void *tris,
                               uint32_t n_tris, uint32_t max_bounces, int px, int py)
{
    rt::BuiltScene bs;
    rt::BuildOptions opt;
    rt::build_scene(spheres, n_spheres, tris, n_tris, opt, bs);
    rt::FrameParams fp;
    std::string err;
    if (!rt::make_frame_params(uniforms80, max_bounces, fp, err)) return -1;
    rt::SceneView sc{};
    sc.nodes = bs.nodes.data(); sc.nodes4 = bs.nodes4.data(); sc.nodes4q = bs.nodes4q.data(); sc.root4 = bs.root4; sc.tri_isect = bs.tri_isect.data(); sc.tri_shade = bs.tri_shade.data(); sc.tri_shade_leaf = bs.tri_shade_leaf.data(); sc.tri_orig = bs.tri_orig.data();
    sc.tri_raw = bs.tri_raw.data(); sc.spheres = bs.spheres.data(); sc.lights = bs.lights.data(); sc.sph_leaf = bs.sph_leaf.data(); sc.sph_orig = bs.sph_orig.data(); sc.n_linear_spheres = bs.n_spheres - bs.n_tree_spheres;
    sc.n_spheres = bs.n_spheres; sc.n_tris = bs.n_tris; sc.n_lights = bs.n_lights;
    sc.n_lights_pass1 = bs.n_lights_pass1; sc.total_power = bs.total_power; sc.root = bs.root;
    HostStack st;
    uint32_t state;
    rt::f3 rd0 = rt::primary_direction(fp, px, py, state);
    int mismatches = 0;
    for (uint32_t s = 0; s < fp.spp; s++) {
        rt::f3 ro = fp.cam_pos, rd = rd0, color = rt::mk3(1, 1, 1), light = rt::mk3(0, 0, 0);
        bool spec = false;
        for (uint32_t i = 0; i < max_bounces; i++) {
            float tb, tv;
            int32_t idb = rt::brute_closest(sc, ro, rd, tb);
            int32_t idv = rt::trace_closest(sc, ro, rd, st, tv, nullptr);
            if (idb != idv || (idb >= 0 && tb != tv)) {
                std::printf("closest mismatch s=%u b=%u o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) brute id=%d t=%.9g bvh id=%d t=%.9g\n",
                            s, i, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, idb, tb, idv, tv);
                mismatches++;
            }
            if (idb < 0) break;
            rt::Hit hit = rt::fetch_hit(sc, ro, rd, idb, tb);
            rt::f3 hp = rt::add(ro, rt::scale(rd, hit.t));
            rt::LightSample ts = rt::sample_light_triangles(sc, hp, state);
            if (ts.pdf > 0.0f) {
                float ct = rt::dot(hit.normal, ts.direction);
                if (ct > 0.0f) {
                    rt::f3 so = rt::add(hp, rt::scale(hit.normal, 0.01f));
                    float limit = ts.distance - 0.1f, stt;
                    int32_t sid = rt::brute_closest(sc, so, ts.direction, stt);
                    bool reached_b = stt < 0.0f || stt > limit;
                    bool reached_v = !rt::trace_occluded(sc, so, ts.direction, limit, st, nullptr);
                    if (reached_b != reached_v) {
                        std::printf("shadow mismatch s=%u b=%u o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) limit=%.9g brute id=%d t=%.9g reached=%d bvh reached=%d\n",
                                    s, i, so.x, so.y, so.z, ts.direction.x, ts.direction.y, ts.direction.z, limit, sid, stt, reached_b, reached_v);
                        mismatches++;
                    }
                    if (reached_b) light = rt::add(light, rt::nee_contribution(ts, hit.normal, ct, color));
                }
            }
            rt::f3 em; bool has_em;
            bool alive = rt::finish_bounce(hit, i, ro, rd, color, em, has_em, spec, state);
            if (has_em) light = rt::add(light, em);
            if (!alive) break;
        }
    }
    return mismatches;
}

// ---------------------------------------------------------------------------
// Emulation of the persistent kernel's wave: 64 rt::Lane state machines driven
// by the same phases (hand out pixels, traversal steps with a shading
// threshold and postponed leaves, shade).  Checks rt_lane.h on the host.
// ---------------------------------------------------------------------------
#include "../../raytracer_amd/csrc/rt_lane.h"

extern "C" int har_render_frames_lanes(const void *uniforms80, const void *spheres, uint32_t n_spheres,
                                       const void *tris, uint32_t n_tris, uint32_t max_bounces,
                                       uint32_t first_frame, uint32_t n_frames, int shade_num, int leaf_lanes,
                                       float *accum, uint16_t *frame_f16, int32_t *ids, uint64_t *counters,
                                       int n_threads)
{
    rt::BuiltScene bs;
    rt::BuildOptions opt;
    if (getenv("RT_GREEDY_COLLAPSE")) opt.greedy_collapse = true; // for tests/harness/tree_stats.py
    opt.quant_nodes = g_link_mode == 64;
    rt::build_scene(spheres, n_spheres, tris, n_tris, opt, bs);
    float uni[20];
    std::memcpy(uni, uniforms80, 80);
    uni[12] = (float)first_frame;
    rt::FrameParams fp;
    std::string err;
    if (!rt::make_frame_params(uni, max_bounces, fp, err)) return -1;
    rt::SceneView sc{};
    sc.nodes = bs.nodes.data(); sc.nodes4 = bs.nodes4.data(); sc.nodes4q = bs.nodes4q.data(); sc.root4 = bs.root4; sc.tri_isect = bs.tri_isect.data(); sc.tri_shade = bs.tri_shade.data(); sc.tri_shade_leaf = bs.tri_shade_leaf.data(); sc.tri_orig = bs.tri_orig.data();
    sc.tri_raw = bs.tri_raw.data(); sc.spheres = bs.spheres.data(); sc.lights = bs.lights.data(); sc.sph_leaf = bs.sph_leaf.data(); sc.sph_orig = bs.sph_orig.data(); sc.n_linear_spheres = bs.n_spheres - bs.n_tree_spheres;
    sc.n_spheres = bs.n_spheres; sc.n_tris = bs.n_tris; sc.n_lights = bs.n_lights;
    sc.n_lights_pass1 = bs.n_lights_pass1; sc.total_power = bs.total_power; sc.root = bs.root;

    const bool sl = g_sphere_lights != 0; // the sphere-light extension, as set by har_set_sphere_lights
    const int requested_mode = g_link_mode;
    const int link_mode = requested_mode < 0 ? bs.links_packed : 0;
    g_link_mode = link_mode; // what HostStack sees during this call
    const int32_t n_nodes4 = (int32_t)(bs.nodes4.size() / 8);
    std::atomic<int> bad_links{0};
    const int W = fp.width, H = fp.height;
    const uint32_t tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8, tiles_per_frame = tiles_x * tiles_y;
    const uint32_t total_tiles = tiles_per_frame * n_frames;
    std::vector<uint16_t> batch((size_t)n_frames * W * H * 4);
    std::atomic<uint32_t> work{0};
    std::atomic<uint64_t> rays{0}, traversed{0}, high{0}, sum_boxes{0}, sum_tris{0};

    auto wave = [&]() {
        rt::Lane L[64];
        HostStack st[64];
        rt::RegPark park[64];
        for (auto &l : L) l.mode = rt::kLaneNeedUnit;
        rt::RayCounts n_rays = {0, 0};
        rt::TestCounts cnt = {0, 0, 0, 0};
        uint32_t pool_tile = 0, pool_next = 64;
        bool exhausted = false;
        for (;;) {
            for (;;) {
                int n_need = 0;
                for (auto &l : L) n_need += l.mode == rt::kLaneNeedUnit;
                if (!n_need) break;
                if (pool_next >= 64) {
                    if (exhausted) { for (auto &l : L) if (l.mode == rt::kLaneNeedUnit) l.mode = rt::kLaneDead; break; }
                    uint32_t tkt = work.fetch_add(1);
                    if (tkt >= total_tiles) { exhausted = true; continue; }
                    pool_tile = tkt; pool_next = 0;
                }
                uint32_t rank = 0;
                for (int i = 0; i < 64; i++) {
                    if (L[i].mode != rt::kLaneNeedUnit) continue;
                    uint32_t slot = pool_next + rank++;
                    if (slot >= 64) continue;
                    uint32_t frame_k = pool_tile / tiles_per_frame, tile = pool_tile % tiles_per_frame;
                    int px = (tile % tiles_x) * 8 + (slot & 7), py = (tile / tiles_x) * 8 + (slot >> 3);
                    if (px < W && py < H) {
                        L[i].unit = (frame_k * H + py) * W + px; // this emulation numbers units row-major
                        rt::FrameParams f = fp;
                        f.seed_add = (uint32_t)(float)(first_frame + frame_k) * 12345u;
                        rt::lane_begin_pixel(sc, f, L[i], px, py, st[i], park[i], n_rays, &cnt);
                    }
                }
                pool_next += std::min<uint32_t>(n_need, 64 - pool_next);
            }
            bool any_alive = false;
            for (auto &l : L) any_alive |= l.mode != rt::kLaneDead;
            if (!any_alive) break;
            for (;;) {
                int n_trav = 0, n_shade = 0;
                for (auto &l : L) { n_trav += l.mode == rt::kLaneTraverse; n_shade += l.mode == rt::kLaneShadeClosest; }
                if (!n_trav) break;
                if (n_shade && n_trav * 8 < (n_trav + n_shade) * shade_num) break;
                for (int i = 0; i < 64; i++)
                    if (L[i].mode == rt::kLaneTraverse && rt::link_is_inner(L[i].cur)) {
                        if (L[i].cur >= n_nodes4) { bad_links++; L[i].cur = rt::kNoRoot; L[i].mode = rt::kLaneShadeClosest; continue; }
                        if (link_mode == 16) rt::lane_step_node<16>(sc, L[i], st[i], &cnt);
                        else if (requested_mode == 64) rt::lane_step_node<64>(sc, L[i], st[i], &cnt);
                        else rt::lane_step_node(sc, L[i], st[i], &cnt);
                    }
                int n_leaf = 0, n_inner = 0;
                for (auto &l : L) {
                    n_leaf += l.mode == rt::kLaneTraverse && rt::link_is_leaf(l.cur);
                    n_inner += l.mode == rt::kLaneTraverse && rt::link_is_inner(l.cur);
                }
                if (n_leaf && (n_leaf >= leaf_lanes || !n_inner))
                    for (int i = 0; i < 64; i++)
                        if (L[i].mode == rt::kLaneTraverse && rt::link_is_leaf(L[i].cur)) {
                            uint32_t code = (uint32_t)~L[i].cur;
                            if ((code >> 3) + (code & 7u) >= bs.n_tris + bs.n_tree_spheres) { bad_links++; L[i].cur = rt::kNoRoot; L[i].mode = rt::kLaneShadeClosest; continue; }
                            rt::lane_step_triangle(sc, L[i], st[i], &cnt);
                        }
                for (int i = 0; i < 64; i++)
                    if (L[i].mode == rt::kLaneShadowDone) {
                        if (sl) rt::lane_shadow_done<true>(sc, fp, L[i], st[i], park[i], n_rays, &cnt);
                        else rt::lane_shadow_done(sc, fp, L[i], st[i], park[i], n_rays, &cnt);
                    }
            }
            for (int i = 0; i < 64; i++) {
                // (a shadow hand-over can start the bounce's second shadow ray, which a lane may finish at once)
                while (L[i].mode == rt::kLaneShadowDone) {
                    if (sl) rt::lane_shadow_done<true>(sc, fp, L[i], st[i], park[i], n_rays, &cnt);
                    else rt::lane_shadow_done(sc, fp, L[i], st[i], park[i], n_rays, &cnt);
                }
                if (L[i].mode == rt::kLaneShadeClosest) {
                    if (sl) rt::lane_shade_closest<true>(sc, fp, L[i], st[i], park[i], n_rays, &cnt);
                    else rt::lane_shade_closest(sc, fp, L[i], st[i], park[i], n_rays, &cnt);
                }
            }
            for (int i = 0; i < 64; i++) {
                if (L[i].mode != rt::kLanePixelDone) continue;
                rt::f3 c = rt::lane_pixel_color(fp, park[i]);
                size_t p = L[i].unit % ((size_t)W * H);
                uint32_t frame_k = L[i].unit / ((uint32_t)W * H);
                uint16_t *dst = &batch[((size_t)frame_k * W * H + p) * 4];
                dst[0] = rt::f32_to_f16_bits(c.x); dst[1] = rt::f32_to_f16_bits(c.y);
                dst[2] = rt::f32_to_f16_bits(c.z); dst[3] = 0x3c00;
                if (frame_k + 1 == n_frames && ids) ids[p] = rt::lane_primary_id(sc, park[i]);
                L[i].mode = rt::kLaneNeedUnit;
            }
        }
        rays += n_rays.calls; traversed += n_rays.traversed; sum_boxes += cnt.boxes; sum_tris += cnt.tris;
        for (auto &x : st) { uint64_t hv = (uint64_t)x.high, cur = high.load(); while (hv > cur && !high.compare_exchange_weak(cur, hv)) {} }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; i++) th.emplace_back(wave);
    wave();
    for (auto &t : th) t.join();
    size_t plane = (size_t)W * H;
    for (size_t p = 0; p < plane; p++) {
        for (uint32_t k = 0; k < n_frames; k++) {
            const uint16_t *src = &batch[((size_t)k * plane + p) * 4];
            float ff = (float)(first_frame + k);
            for (int ch = 0; ch < 3; ch++)
                accum[ch * plane + p] = rt::accumulate_channel(accum[ch * plane + p], rt::f16_bits_to_f32(src[ch]), ff);
            if (k + 1 == n_frames && frame_f16) std::memcpy(frame_f16 + 4 * p, src, 8);
        }
    }
    if (counters) { counters[0] += rays; counters[13] += traversed; counters[1] += sum_boxes; counters[2] += sum_tris; counters[14] = bs.max_stack4; counters[15] = bs.nodes4.size() / 8; counters[7] = high; }
    g_link_mode = requested_mode;
    if (bad_links) return -2;      // a link led outside the tables
    return requested_mode < 0 ? link_mode : 0;
}
