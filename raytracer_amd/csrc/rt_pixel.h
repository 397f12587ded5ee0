// rt_pixel.h -- one pixel of one frame in the loop structure of the reference
// shader (raytracer_compute.wgsl:189-278 and :319-337): samples outside,
// bounces inside.  Used by the one-thread-per-pixel kernel; the persistent
// kernel runs the same bounce pieces from rt_core.h as a per-lane state machine.
#pragma once

#include "rt_core.h"

namespace rt {

// raytracer_compute.wgsl:189-278.  BRUTE selects the reference's own ray_all.
template <bool BRUTE, class Stack>
RT_HD f3 ray_trace(const SceneView &sc, f3 ro, f3 rd, uint32_t max_bounces, bool sphere_lights, uint32_t &state,
                   Stack &stack, uint32_t &n_rays, int32_t &first_id, TestCounts *cnt)
{
    f3 sky_color = scale(mk3(1.0f, 1.0f, 1.0f), 0.4f);
    f3 color = mk3(1.0f, 1.0f, 1.0f);
    f3 light = mk3(0.0f, 0.0f, 0.0f);
    bool is_specular_bounce = false;

    for (uint32_t i = 0; i < max_bounces; i++) {
        float t;
        int32_t id = BRUTE ? brute_closest(sc, ro, rd, t) : trace_closest(sc, ro, rd, stack, t, cnt);
        n_rays++;
        if (i == 0u) first_id = id;
        if (id < 0) {
            light = add(light, mul(sky_color, color));
            break;
        }
        if (cnt) cnt->bounces++;
        Hit hit = fetch_hit(sc, ro, rd, id, t);
        f3 hit_point = add(ro, scale(rd, hit.t));

        LightSample ts = sample_light_triangles(sc, hit_point, state);
        if (ts.pdf > 0.0f) {
            float cos_theta = dot(hit.normal, ts.direction);
            if (cos_theta > 0.0f) {
                f3 so = add(hit_point, scale(hit.normal, 0.01f));
                float limit = ts.distance - 0.1f;
                bool reached;
                if (BRUTE) {
                    float st;
                    (void)brute_closest(sc, so, ts.direction, st);
                    reached = st < 0.0f || st > limit;
                } else {
                    reached = !trace_occluded(sc, so, ts.direction, limit, stack, cnt);
                }
                n_rays++;
                if (reached) light = add(light, nee_contribution(ts, hit.normal, cos_theta, color));
            }
        }
        // Extension (off by default): the sphere-light sampling that the shader's TODO at
        // raytracer_compute.wgsl:226 asks for, with sample_light_spheres (:109-171) wired exactly
        // like the triangle-light block above: a second sample, a second shadow ray.
        if (sphere_lights) {
            LightSample ss = sample_light_spheres(sc, hit_point, state);
            if (ss.pdf > 0.0f) {
                float cos_theta = dot(hit.normal, ss.direction);
                if (cos_theta > 0.0f) {
                    f3 so = add(hit_point, scale(hit.normal, 0.01f));
                    float limit = ss.distance - 0.1f;
                    bool reached;
                    if (BRUTE) {
                        float st;
                        (void)brute_closest(sc, so, ss.direction, st);
                        reached = st < 0.0f || st > limit;
                    } else {
                        reached = !trace_occluded(sc, so, ss.direction, limit, stack, cnt);
                    }
                    n_rays++;
                    if (reached) light = add(light, nee_contribution(ss, hit.normal, cos_theta, color));
                }
            }
        }
        f3 em;
        bool has_em;
        bool alive = finish_bounce(hit, i, ro, rd, color, em, has_em, is_specular_bounce, state);
        if (has_em) light = add(light, em);
        if (!alive) break;
    }
    return light;
}

// raytracer_compute.wgsl:293-337: returns the frame colour of the pixel
// (before the rgba16float store).
template <bool BRUTE, class Stack>
RT_HD f3 render_pixel(const SceneView &sc, const FrameParams &fp, int px, int py, Stack &stack,
                      uint32_t &n_rays, int32_t &primary_id, TestCounts *cnt)
{
    uint32_t state;
    f3 rd = primary_direction(fp, px, py, state);
    f3 total = mk3(0.0f, 0.0f, 0.0f);
    primary_id = -1;
    for (uint32_t s = 0; s < fp.spp; s++) {
        int32_t fid = -1;
        total = add(total, ray_trace<BRUTE>(sc, fp.cam_pos, rd, fp.max_bounces, fp.sphere_lights != 0u, state, stack, n_rays, fid, cnt));
        if (s == 0u) primary_id = fid;
    }
    return divs(total, fp.spp_f);
}

} // namespace rt
