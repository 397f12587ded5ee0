/*
 * CPU oracle for the guymor4/raytracer hot path -- TEST INFRASTRUCTURE ONLY.
 * See wgsl_oracle.h for scope, the floating-point contract and the
 * "parity unpinned" statement.  Build: -O2 -ffp-contract=off, no -ffast-math.
 *
 * Every function cites the reference lines it restates.  The statement order,
 * operand order and branch structure follow the WGSL text so that the file can
 * be read side by side with it.
 */
#include "wgsl_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { return vdivs(a, vlength(a)); }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

static const float ORC_PI = 3.14159265358979323846f; /* common.wgsl:4 */

/* ---- record views over the reference byte layouts (src/gpu-schemas.ts) ---- */

typedef struct { /* common.wgsl:12-20, stride 64 */
    v3 center; float radius;
    v3 color; float smoothness;
    v3 emissionColor; float emissionStrength;
    float specularProbability;
} SphereV;

typedef struct { /* common.wgsl:22-32, stride 96 */
    v3 v0, v1, v2;
    v3 color;
    v3 emissionColor; float emissionStrength;
    float smoothness; float specularProbability;
} TriV;

static SphereV load_sphere(const void *p)
{
    const float *f = (const float *)p;
    SphereV s;
    s.center = ld3(f + 0); s.radius = f[3];
    s.color = ld3(f + 4); s.smoothness = f[7];
    s.emissionColor = ld3(f + 8); s.emissionStrength = f[11];
    s.specularProbability = f[12];
    return s;
}

static TriV load_tri(const void *p)
{
    const float *f = (const float *)p;
    TriV t;
    t.v0 = ld3(f + 0); t.v1 = ld3(f + 4); t.v2 = ld3(f + 8);
    t.color = ld3(f + 12);
    t.emissionColor = ld3(f + 16); t.emissionStrength = f[19];
    t.smoothness = f[20]; t.specularProbability = f[21];
    return t;
}

typedef struct { /* common.wgsl:34-48, size 80 */
    v3 cam_position, cam_rotation;
    float cam_fov;
    float frameIndex;
    float res_x, res_y;
    float samplesPerPixel;
    float debugEnabled;
} UniV;

static UniV load_uniforms(const void *p)
{
    const float *f = (const float *)p;
    UniV u;
    u.cam_position = ld3(f + 0);
    u.cam_rotation = ld3(f + 4);
    u.cam_fov = f[7];
    u.frameIndex = f[12];
    u.res_x = f[14]; u.res_y = f[15];
    u.samplesPerPixel = f[16];
    u.debugEnabled = f[17];
    return u;
}

/* ---- integer RNG ---- */

uint32_t orc_wang_hash(uint32_t s) /* common.wgsl:66-73 */
{
    uint32_t ns = (s ^ 61u) ^ (s >> 16);
    ns = ns * 9u;
    ns = ns ^ (ns >> 4);
    ns = ns * 0x27d4eb2du;
    ns = ns ^ (ns >> 15);
    return ns;
}

float orc_rand_f(uint32_t *state) /* common.wgsl:60-64 */
{
    *state = *state * 747796405u + 2891336453u;
    uint32_t word = ((*state >> ((*state >> 28) + 4u)) ^ *state) * 277803737u;
    uint32_t bits = 0x2f800004u;
    float scale;
    memcpy(&scale, &bits, 4);
    return (float)((word >> 22) ^ word) * scale;
}

/* ---- transcendental contract ---- */

/*
 * sin and cos of x for |x| < 2^15: quadrant k = rint(x * 2/pi), three-constant
 * Cody-Waite reduction r = ((x - k*A) - k*B) - k*C, then the cephes single
 * precision minimax polynomials on [-pi/4, pi/4].  Every step is one f32
 * operation in the order written.
 */
void orc_sincosf(float x, float *s, float *c)
{
    const float two_over_pi = 0.636619772367581343f;
    const float A = 1.5703125f;
    const float B = 4.837512969970703125e-4f;
    const float C = 7.54978995489188216e-8f;
    float k = rintf(x * two_over_pi);
    int q = (int)k;
    float r = x - k * A;
    r = r - k * B;
    r = r - k * C;
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    float sn = r + r * (z * ps);
    float pc = 2.443315711809948e-5f;
    pc = pc * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    float cs = (1.0f - 0.5f * z) + (z * z) * pc;
    switch (q & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}

/* ---- f16 ---- */

uint16_t orc_f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        if (ax == 0x7f800000u) return (uint16_t)(sign | 0x7c00u);
        return (uint16_t)(sign | 0x7e00u | ((ax >> 13) & 0x1ffu));
    }
    if (ax >= 0x477ff000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x33000001u) /* <= 2^-25 rounds to zero (tie to even) */
        return (uint16_t)sign;
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    uint32_t shift, half_bits;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        half_bits = 0;
    } else {
        shift = 13;
        half_bits = (uint32_t)(e + 15) << 10;
        m &= 0x7fffffu;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    return (uint16_t)(sign | (half_bits + q)); /* carry into exponent is correct */
}

float orc_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 127 - 15) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* ---- intersections ---- */

static OrcHit miss_hit(void) /* HitInfo(-1.0, vec3(), vec3(), vec3(), 0, 0) */
{
    OrcHit h;
    memset(&h, 0, sizeof h);
    h.t = -1.0f;
    h.id = -1;
    return h;
}

static OrcHit make_hit(float t, v3 color, v3 normal, v3 emission, float smooth, float spec)
{
    OrcHit h;
    h.t = t;
    st3(h.color, color); st3(h.normal, normal); st3(h.emission, emission);
    h.smoothness = smooth; h.specularProbability = spec;
    h.id = -1;
    return h;
}

static OrcHit ray_sphere(v3 ro, v3 rd, const SphereV *sp) /* common.wgsl:137-168 */
{
    v3 oc = vsub(ro, sp->center);
    float a = vdot(rd, rd);
    float b = 2.0f * vdot(oc, rd);
    float c = vdot(oc, oc) - sp->radius * sp->radius;
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant < 0.0f) return miss_hit();

    float sqrt_discriminant = sqrtf(discriminant);
    float t1 = (-b - sqrt_discriminant) / (2.0f * a);
    float t2 = (-b + sqrt_discriminant) / (2.0f * a);

    v3 emission = vscale(sp->emissionColor, sp->emissionStrength);

    if (t1 > 0.01f) {
        v3 hit_point = vadd(ro, vscale(rd, t1));
        v3 normal = vnormalize(vsub(hit_point, sp->center));
        return make_hit(t1, sp->color, normal, emission, sp->smoothness, sp->specularProbability);
    }
    if (t2 > 0.01f) {
        v3 hit_point = vadd(ro, vscale(rd, t2));
        v3 normal = vnormalize(vsub(hit_point, sp->center));
        /* common.wgsl:164 returns t1 here, not t2 */
        return make_hit(t1, sp->color, normal, emission, sp->smoothness, sp->specularProbability);
    }
    return miss_hit();
}

static OrcHit ray_triangle(v3 ro, v3 rd, const TriV *tri) /* common.wgsl:170-211 */
{
    v3 edge01 = vsub(tri->v1, tri->v0);
    v3 edge02 = vsub(tri->v2, tri->v0);

    v3 h = vcross(rd, edge02);
    float a = vdot(edge01, h);
    if (fabsf(a) < 0.001f) return miss_hit();

    float f = 1.0f / a;
    v3 s = vsub(ro, tri->v0);
    float u = f * vdot(s, h);
    if (u < 0.0f || u > 1.0f) return miss_hit();

    v3 q = vcross(s, edge01);
    float v = f * vdot(rd, q);
    if (v < 0.0f || u + v > 1.0f) return miss_hit();

    float t = f * vdot(edge02, q);
    if (t < 0.01f) return miss_hit();

    v3 normal = vnormalize(vcross(edge01, edge02));
    if (vdot(normal, rd) > 0.0f) return miss_hit();

    v3 emission = vscale(tri->emissionColor, tri->emissionStrength);
    return make_hit(t, tri->color, normal, emission, tri->smoothness, tri->specularProbability);
}

typedef struct {
    const unsigned char *spheres; uint32_t n_spheres;
    const unsigned char *tris; uint32_t n_tris;
    int sphere_lights; /* ORC_SPHERE_LIGHTS: ray_trace also samples emissive spheres (an extension, see below) */
    int debug_view;    /* ORC_DEBUG_VIEW: ray_all also tests every triangle's normal marker (an extension, see below) */
} SceneV;

/* common.wgsl:235-240: a sphere of radius 0.1 placed 0.2 along the normal from the triangle's centre,
 * coloured by the normal.  The shader defines it and never calls it; ray_all below calls it only
 * with ORC_DEBUG_VIEW. */
static OrcHit ray_triangle_normal(v3 ro, v3 rd, const TriV *tri)
{
    v3 triangleCenter = vdivs(vadd(vadd(tri->v0, tri->v1), tri->v2), 3.0f);
    v3 triangleNormal = vnormalize(vcross(vsub(tri->v1, tri->v0), vsub(tri->v2, tri->v0)));
    SphereV sp;
    sp.center = vadd(triangleCenter, vscale(triangleNormal, 0.2f));
    sp.radius = 0.1f;
    sp.color = triangleNormal;
    sp.smoothness = 0.0f;
    sp.emissionColor = V(0.0f, 0.0f, 0.0f);
    sp.emissionStrength = 0.0f;
    sp.specularProbability = 0.0f;
    return ray_sphere(ro, rd, &sp);
}

static OrcHit ray_all(v3 ro, v3 rd, const SceneV *sc) /* raytracer_compute.wgsl:15-39 */
{
    OrcHit closest = miss_hit();
    for (uint32_t i = 0; i < sc->n_spheres; i++) {
        SphereV sp = load_sphere(sc->spheres + (size_t)i * ORC_SPHERE_STRIDE);
        OrcHit hit = ray_sphere(ro, rd, &sp);
        if (hit.t > 0.0f && (closest.t < 0.0f || hit.t < closest.t)) {
            closest = hit;
            closest.id = (int32_t)i;
        }
    }
    for (uint32_t i = 0; i < sc->n_tris; i++) {
        TriV tr = load_tri(sc->tris + (size_t)i * ORC_TRIANGLE_STRIDE);
        OrcHit hit = ray_triangle(ro, rd, &tr);
        if (hit.t > 0.0f && (closest.t < 0.0f || hit.t < closest.t)) {
            closest = hit;
            closest.id = (int32_t)(sc->n_spheres + i);
        }
        /* Extension, off unless ORC_DEBUG_VIEW: the debug view that uniforms.debugEnabled
         * (src/webgpu-renderer.ts:228) and ray_triangle_normal_intersect were written for and that the
         * shader never got: the marker of each triangle is one more candidate, right after its
         * triangle, under the same selection rule; ids of markers follow those of the triangles. */
        if (sc->debug_view) {
            OrcHit mark = ray_triangle_normal(ro, rd, &tr);
            if (mark.t > 0.0f && (closest.t < 0.0f || mark.t < closest.t)) {
                closest = mark;
                closest.id = (int32_t)(sc->n_spheres + sc->n_tris + i);
            }
        }
    }
    return closest;
}

/* ---- sampling ---- */

static float luminance(v3 c) /* raytracer_compute.wgsl:11-13 */
{
    return vdot(c, V(0.2126f, 0.7152f, 0.0722f));
}

static float triangle_area(const TriV *t) /* common.wgsl:228-232 */
{
    v3 edge1 = vsub(t->v1, t->v0);
    v3 edge2 = vsub(t->v2, t->v0);
    return 0.5f * vlength(vcross(edge1, edge2));
}

static v3 sample_triangle_point(const TriV *t, uint32_t *state) /* common.wgsl:214-225 */
{
    float r1 = orc_rand_f(state);
    float r2 = orc_rand_f(state);
    float sqrt_r1 = sqrtf(r1);
    float u = 1.0f - sqrt_r1;
    float v = r2 * sqrt_r1;
    float w = 1.0f - u - v;
    return vadd(vadd(vscale(t->v0, u), vscale(t->v1, v)), vscale(t->v2, w));
}

typedef struct { v3 direction; v3 emission; float distance; float pdf; } LightSample;

static LightSample zero_light_sample(void)
{
    LightSample s;
    s.direction = V(0, 0, 0); s.emission = V(0, 0, 0); s.distance = 0.0f; s.pdf = 0.0f;
    return s;
}

/* raytracer_compute.wgsl:42-107 */
static LightSample sample_light_triangles(v3 hit_point, uint32_t *state, const SceneV *sc)
{
    float total_power = 0.0f;
    uint32_t light_count = 0;
    for (uint32_t i = 0; i < sc->n_tris; i++) {
        TriV tr = load_tri(sc->tris + (size_t)i * ORC_TRIANGLE_STRIDE);
        if (tr.emissionStrength <= 0.0f) continue;
        float area = triangle_area(&tr);
        float power = tr.emissionStrength * area * luminance(tr.emissionColor);
        total_power += power;
        light_count++;
    }
    if (light_count == 0u || total_power <= 0.0f) return zero_light_sample();

    float random_power = orc_rand_f(state) * total_power;
    float accumulated_power = 0.0f;
    TriV selected;
    memset(&selected, 0, sizeof selected); /* var selected_triangle: Triangle; zero value */
    int found = 0;
    for (uint32_t i = 0; i < sc->n_tris; i++) {
        TriV tr = load_tri(sc->tris + (size_t)i * ORC_TRIANGLE_STRIDE);
        if (tr.emissionStrength > 0.0f) {
            float area = triangle_area(&tr);
            float power = tr.emissionStrength * area * luminance(tr.emissionColor);
            accumulated_power += power;
            if (accumulated_power >= random_power && !found) {
                selected = tr;
                found = 1;
            }
        }
    }

    v3 light_point = sample_triangle_point(&selected, state);
    v3 light_direction = vsub(light_point, hit_point);
    float distance = vlength(light_direction);
    v3 normalized_direction = vdivs(light_direction, distance);

    v3 edge1 = vsub(selected.v1, selected.v0);
    v3 edge2 = vsub(selected.v2, selected.v0);
    v3 light_normal = vnormalize(vcross(edge1, edge2));

    float cos_light = -vdot(light_normal, normalized_direction);
    if (cos_light <= 0.0f) return zero_light_sample();

    float area = triangle_area(&selected);
    float power = selected.emissionStrength * area * luminance(selected.emissionColor);
    float pdf = (distance * distance) / (area * cos_light) * (power / total_power);

    LightSample out;
    out.direction = normalized_direction;
    out.emission = vscale(selected.emissionColor, selected.emissionStrength);
    out.distance = distance;
    out.pdf = pdf;
    return out;
}

/*
 * raytracer_compute.wgsl:109-171.  The shader defines this function and never calls it
 * (its author's TODO at :226); ray_trace below calls it only with ORC_SPHERE_LIGHTS.
 * `var selected_sphere: Sphere` is zero-initialised in WGSL, which is what remains when
 * u32(rand * count) lands on count (rand_f can reach 1.0000005).
 */
static LightSample sample_light_spheres(v3 hit_point, uint32_t *state, const SceneV *sc)
{
    uint32_t light_count = 0u;
    for (uint32_t i = 0; i < sc->n_spheres; i++) {
        SphereV sp = load_sphere(sc->spheres + 64u * (size_t)i);
        if (sp.emissionStrength > 0.0f) light_count++;
    }
    if (light_count == 0u) return zero_light_sample();

    uint32_t random_sphere_index = (uint32_t)(orc_rand_f(state) * (float)light_count);
    uint32_t current_index = 0u;
    SphereV selected;
    memset(&selected, 0, sizeof selected);
    for (uint32_t i = 0; i < sc->n_spheres; i++) {
        SphereV sp = load_sphere(sc->spheres + 64u * (size_t)i);
        if (sp.emissionStrength > 0.0f) {
            if (current_index == random_sphere_index) { selected = sp; break; }
            current_index++;
        }
    }

    v3 to_center = vsub(selected.center, hit_point);
    float distance_to_center = vlength(to_center);
    if (distance_to_center <= selected.radius) return zero_light_sample();

    v3 center_direction = vdivs(to_center, distance_to_center);
    v3 up = fabsf(center_direction.y) > 0.99f ? V(1.0f, 0.0f, 0.0f) : V(0.0f, 1.0f, 0.0f);
    v3 right = vnormalize(vcross(up, center_direction));
    v3 forward = vcross(center_direction, right);

    float cos_theta_max = sqrtf(1.0f - (selected.radius * selected.radius) / (distance_to_center * distance_to_center));
    float cos_theta = 1.0f - orc_rand_f(state) * (1.0f - cos_theta_max);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * ORC_PI * orc_rand_f(state);
    float sphi, cphi;
    orc_sincosf(phi, &sphi, &cphi);

    v3 sample_direction = vnormalize(vadd(vadd(vscale(center_direction, cos_theta), vscale(right, sin_theta * cphi)),
                                          vscale(forward, sin_theta * sphi)));

    LightSample out;
    out.direction = sample_direction;
    out.emission = vscale(selected.emissionColor, selected.emissionStrength);
    out.distance = distance_to_center - selected.radius;
    out.pdf = 1.0f / (2.0f * ORC_PI * (1.0f - cos_theta_max) * (float)light_count);
    return out;
}

static v3 sample_cosine_hemisphere(v3 normal, uint32_t *state) /* common.wgsl:107-134 */
{
    float r1 = orc_rand_f(state);
    float r2 = orc_rand_f(state);

    float phi = 2.0f * ORC_PI * r1;
    float cos_theta = sqrtf(r2);
    float sin_theta = sqrtf(1.0f - r2);

    float sphi, cphi;
    orc_sincosf(phi, &sphi, &cphi);
    v3 local_dir = V(cphi * sin_theta, cos_theta, sphi * sin_theta);

    v3 up = (fabsf(normal.y) > 0.99f) ? V(1.0f, 0.0f, 0.0f) : V(0.0f, 1.0f, 0.0f);
    v3 tangent = vnormalize(vcross(up, normal));
    v3 bitangent = vcross(normal, tangent);

    return vnormalize(vadd(vadd(vscale(tangent, local_dir.x), vscale(normal, local_dir.y)),
                           vscale(bitangent, local_dir.z)));
}

static float power_heuristic(float pdf_a, float pdf_b) /* raytracer_compute.wgsl:174-178 */
{
    float a = pdf_a * pdf_a;
    float b = pdf_b * pdf_b;
    return a / (a + b);
}

static float brdf_pdf(v3 hit_normal, v3 direction) /* raytracer_compute.wgsl:181-187 */
{
    float cos_theta = vdot(hit_normal, direction);
    if (cos_theta <= 0.0f) return 0.0f;
    return cos_theta / ORC_PI;
}

static float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* raytracer_compute.wgsl:189-278 */
static v3 ray_trace(v3 ro, v3 rd, uint32_t maxBounceCount, uint32_t *state,
                    const SceneV *sc, uint64_t *n_rays, int32_t *first_id)
{
    v3 sky_color = vscale(V(1.0f, 1.0f, 1.0f), 0.4f);
    v3 color = V(1, 1, 1);
    v3 light = V(0, 0, 0);
    int is_specular_bounce = 0;

    v3 cur_o = ro, cur_d = rd;
    for (uint32_t i = 0; i < maxBounceCount; i++) {
        OrcHit hit_info = ray_all(cur_o, cur_d, sc);
        (*n_rays)++;
        if (i == 0u && first_id) *first_id = hit_info.id;
        if (hit_info.t < 0.0f) {
            light = vadd(light, vmul(sky_color, color));
            break;
        }
        v3 hit_normal = ld3(hit_info.normal);
        v3 hit_emission = ld3(hit_info.emission);
        v3 hit_point = vadd(cur_o, vscale(cur_d, hit_info.t));

        LightSample ts = sample_light_triangles(hit_point, state, sc);
        if (ts.pdf > 0.0f) {
            float cos_theta = vdot(hit_normal, ts.direction);
            if (cos_theta > 0.0f) {
                v3 so = vadd(hit_point, vscale(hit_normal, 0.01f));
                OrcHit shadow_hit = ray_all(so, ts.direction, sc);
                (*n_rays)++;
                if (shadow_hit.t < 0.0f || shadow_hit.t > ts.distance - 0.1f) {
                    float brdf_pdf_val = brdf_pdf(hit_normal, ts.direction);
                    float mis_weight = power_heuristic(ts.pdf, brdf_pdf_val);
                    float brdf = cos_theta / ORC_PI;
                    v3 c = vscale(ts.emission, brdf);
                    c = vmul(c, color);
                    c = vscale(c, mis_weight);
                    c = vdivs(c, ts.pdf);
                    light = vadd(light, c);
                }
            }
        }

        /* Extension, off unless ORC_SPHERE_LIGHTS: the sphere-light sampling the shader's TODO at
         * raytracer_compute.wgsl:226 asks for, wired exactly like the triangle block above. */
        if (sc->sphere_lights) {
            LightSample ss = sample_light_spheres(hit_point, state, sc);
            if (ss.pdf > 0.0f) {
                float cos_theta = vdot(hit_normal, ss.direction);
                if (cos_theta > 0.0f) {
                    v3 so = vadd(hit_point, vscale(hit_normal, 0.01f));
                    OrcHit shadow_hit = ray_all(so, ss.direction, sc);
                    (*n_rays)++;
                    if (shadow_hit.t < 0.0f || shadow_hit.t > ss.distance - 0.1f) {
                        float brdf_pdf_val = brdf_pdf(hit_normal, ss.direction);
                        float mis_weight = power_heuristic(ss.pdf, brdf_pdf_val);
                        float brdf = cos_theta / ORC_PI;
                        v3 c = vscale(ss.emission, brdf);
                        c = vmul(c, color);
                        c = vscale(c, mis_weight);
                        c = vdivs(c, ss.pdf);
                        light = vadd(light, c);
                    }
                }
            }
        }

        int is_primary_ray = (i == 0u);
        int is_emissive = (hit_emission.x + hit_emission.y + hit_emission.z) > 0.01f;
        if ((is_primary_ray && is_emissive) || is_specular_bounce) {
            v3 incoming_dir = vneg(vnormalize(cur_d));
            float brdf_pdf_val = brdf_pdf(hit_normal, incoming_dir);
            float light_pdf_estimate = 0.001f;
            float mis_weight = power_heuristic(brdf_pdf_val, light_pdf_estimate);
            light = vadd(light, vscale(vmul(hit_emission, color), mis_weight));
        }

        color = vmul(color, ld3(hit_info.color));
        if (color.x + color.y + color.z < 0.01f) break;

        if (i >= 3u) {
            float lum = luminance(color);
            float p = clampf(lum, 0.05f, 0.95f);
            if (orc_rand_f(state) > p) break;
            color = vdivs(color, p);
        }

        is_specular_bounce = hit_info.specularProbability >= orc_rand_f(state);

        v3 diffuseDir = sample_cosine_hemisphere(hit_normal, state);
        /* reflect(e1, e2) = e1 - 2 * dot(e2, e1) * e2 */
        v3 specularDir = vsub(cur_d, vscale(hit_normal, 2.0f * vdot(hit_normal, cur_d)));
        float mt = is_specular_bounce ? hit_info.smoothness : 0.0f;
        /* mix(a, b, t) = a * (1 - t) + b * t */
        v3 mixed = vadd(vscale(diffuseDir, 1.0f - mt), vscale(specularDir, mt));
        v3 new_direction = vnormalize(mixed);

        v3 new_origin = vadd(vadd(cur_o, vscale(cur_d, hit_info.t)), vscale(hit_normal, 0.01f));
        cur_o = new_origin;
        cur_d = new_direction;
    }
    return light;
}

/* common.wgsl:80-96 */
static v3 rotate_yaw_pitch(v3 v, float yaw, float pitch)
{
    float cy, sy, cx, sx;
    orc_sincosf(yaw, &sy, &cy);
    orc_sincosf(pitch, &sx, &cx);
    v3 r0 = V(cy, sy * sx, sy * cx);
    v3 r1 = V(0.0f, cx, -sx);
    v3 r2 = V(-sy, cy * sx, cy * cx);
    return vnormalize(V(vdot(r0, v), vdot(r1, v), vdot(r2, v)));
}

/* raytracer_compute.wgsl:290-329 */
static uint32_t primary_ray(const UniV *u, int px, int py, v3 *ro, v3 *rd)
{
    float aspect = u->res_x / u->res_y;
    float uvx = ((float)px + 0.5f) / u->res_x;
    float uvy = ((float)py + 0.5f) / u->res_y;

    uint32_t seed = (uint32_t)py * (uint32_t)u->res_x + (uint32_t)px + (uint32_t)u->frameIndex * 12345u;
    uint32_t state = orc_wang_hash(seed);

    v3 rotation_rad = vdivs(vscale(u->cam_rotation, 3.14159f), 180.0f);
    float yaw = rotation_rad.y;
    float pitch = rotation_rad.x;
    v3 forward = rotate_yaw_pitch(V(0, 0, -1), yaw, pitch);
    v3 world_up = V(0.0f, 1.0f, 0.0f);
    v3 right = vnormalize(vcross(forward, world_up));
    v3 up = vcross(right, forward);

    float fov = u->cam_fov * 3.14159f / 180.0f;
    float sh, ch;
    orc_sincosf(fov * 0.5f, &sh, &ch);
    float focal_length = 1.0f / (sh / ch);

    float jx = orc_rand_f(&state) - 0.5f;
    float jy = orc_rand_f(&state) - 0.5f;
    float offx = jx / u->res_x;
    float offy = jy / u->res_y;
    float cx = ((uvx + offx) * 2.0f - 1.0f) * aspect;
    float cy = 1.0f - (uvy + offy) * 2.0f;
    v3 dir = vnormalize(vadd(vadd(vscale(right, cx), vscale(up, cy)), vscale(forward, focal_length)));
    *ro = u->cam_position;
    *rd = dir;
    return state;
}

/* ---- frame driver: raytracer_compute.wgsl:280-341 + accumulator.wgsl:44-68 ---- */

typedef struct {
    UniV u;
    SceneV sc;
    uint32_t max_bounces;
    int x0, y0, x1, y1, W, H;
    float *accum;
    uint16_t *frame_f16;
    int32_t *ids;
    int next_row; /* atomic */
    uint64_t n_rays; /* atomic */
} FrameJob;

static void render_pixel(FrameJob *j, int px, int py, uint64_t *n_rays)
{
    v3 ro, rd;
    uint32_t state = primary_ray(&j->u, px, py, &ro, &rd);

    uint32_t spp = (uint32_t)j->u.samplesPerPixel;
    v3 total = V(0, 0, 0);
    int32_t pid = -1;
    for (uint32_t i = 0; i < spp; i++) {
        int32_t fid = -1;
        total = vadd(total, ray_trace(ro, rd, j->max_bounces, &state, &j->sc, n_rays, &fid));
        if (i == 0) pid = fid;
    }
    v3 color = vdivs(total, j->u.samplesPerPixel);

    /* textureStore to rgba16float (src/webgpu-renderer.ts:108-112) */
    uint16_t h[4];
    h[0] = orc_f32_to_f16(color.x); h[1] = orc_f32_to_f16(color.y);
    h[2] = orc_f32_to_f16(color.z); h[3] = orc_f32_to_f16(1.0f);
    size_t p = (size_t)py * (size_t)j->W + (size_t)px;
    if (j->frame_f16) memcpy(j->frame_f16 + 4 * p, h, sizeof h);
    if (j->ids) j->ids[p] = pid;

    /* accumulator.wgsl:44-68 */
    size_t plane = (size_t)j->W * (size_t)j->H;
    v3 cur = V(orc_f16_to_f32(h[0]), orc_f16_to_f32(h[1]), orc_f16_to_f32(h[2]));
    v3 stored = V(0, 0, 0);
    if (j->u.frameIndex > 0.0f)
        stored = V(j->accum[p], j->accum[plane + p], j->accum[2 * plane + p]);
    float weight = 1.0f / (j->u.frameIndex + 1.0f);
    v3 acc = vadd(vscale(stored, 1.0f - weight), vscale(cur, weight));
    acc = V(clampf(acc.x, 0.0f, 1.0f), clampf(acc.y, 0.0f, 1.0f), clampf(acc.z, 0.0f, 1.0f));
    j->accum[p] = acc.x;
    j->accum[plane + p] = acc.y;
    j->accum[2 * plane + p] = acc.z;
}

static void *frame_worker(void *arg)
{
    FrameJob *j = (FrameJob *)arg;
    uint64_t n_rays = 0;
    for (;;) {
        int y = __atomic_fetch_add(&j->next_row, 1, __ATOMIC_RELAXED);
        if (y >= j->y1) break;
        for (int x = j->x0; x < j->x1; x++) render_pixel(j, x, y, &n_rays);
    }
    __atomic_fetch_add(&j->n_rays, n_rays, __ATOMIC_RELAXED);
    return NULL;
}

int orc_render_frame(const void *uniforms80,
                     const void *spheres, uint32_t n_spheres,
                     const void *tris, uint32_t n_tris,
                     uint32_t max_bounces,
                     int x0, int y0, int x1, int y1,
                     float *accum, uint16_t *frame_f16, int32_t *ids,
                     uint64_t *counters, int n_threads)
{
    return orc_render_frame_ex(uniforms80, spheres, n_spheres, tris, n_tris, max_bounces, x0, y0, x1, y1, accum,
                               frame_f16, ids, counters, n_threads, 0u);
}

int orc_render_frame_ex(const void *uniforms80,
                        const void *spheres, uint32_t n_spheres,
                        const void *tris, uint32_t n_tris,
                        uint32_t max_bounces,
                        int x0, int y0, int x1, int y1,
                        float *accum, uint16_t *frame_f16, int32_t *ids,
                        uint64_t *counters, int n_threads, uint32_t flags)
{
    if (!uniforms80 || !accum) return -1;
    FrameJob j;
    memset(&j, 0, sizeof j);
    j.u = load_uniforms(uniforms80);
    j.W = (int)j.u.res_x; /* vec2<i32>(uniforms.resolution) */
    j.H = (int)j.u.res_y;
    if (j.W <= 0 || j.H <= 0) return -2;
    if (x0 < 0 || y0 < 0 || x1 > j.W || y1 > j.H || x0 > x1 || y0 > y1) return -3;
    j.sc.spheres = (const unsigned char *)spheres; j.sc.n_spheres = n_spheres;
    j.sc.tris = (const unsigned char *)tris; j.sc.n_tris = n_tris;
    j.sc.sphere_lights = (flags & ORC_SPHERE_LIGHTS) != 0u;
    /* the debug view is drawn only while the uniform says so, as a shader reading it would */
    j.sc.debug_view = (flags & ORC_DEBUG_VIEW) != 0u && j.u.debugEnabled != 0.0f;
    j.max_bounces = max_bounces;
    j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1;
    j.accum = accum; j.frame_f16 = frame_f16; j.ids = ids;
    j.next_row = y0;

    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    int started = 0;
    for (int i = 1; i < n_threads; i++) {
        if (pthread_create(&th[started], NULL, frame_worker, &j) == 0) started++;
    }
    frame_worker(&j);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    if (counters) counters[0] += j.n_rays;
    return 0;
}

/* ---- exported single-function entry points for the unit tests ---- */

void orc_ray_sphere_intersect(const float origin[3], const float dir[3],
                              const void *sphere, OrcHit *out)
{
    SphereV sp = load_sphere(sphere);
    *out = ray_sphere(ld3(origin), ld3(dir), &sp);
}

void orc_ray_triangle_intersect(const float origin[3], const float dir[3],
                                const void *tri, OrcHit *out)
{
    TriV tr = load_tri(tri);
    *out = ray_triangle(ld3(origin), ld3(dir), &tr);
}

void orc_ray_all(const float origin[3], const float dir[3],
                 const void *spheres, uint32_t n_spheres,
                 const void *tris, uint32_t n_tris, OrcHit *out)
{
    SceneV sc = {(const unsigned char *)spheres, n_spheres, (const unsigned char *)tris, n_tris, 0, 0};
    *out = ray_all(ld3(origin), ld3(dir), &sc);
}

float orc_sample_light_triangles(const float hit_point[3], uint32_t *state,
                                 const void *tris, uint32_t n_tris,
                                 float out_dir[3], float out_emission[3],
                                 float *out_distance)
{
    SceneV sc = {NULL, 0, (const unsigned char *)tris, n_tris, 0, 0};
    LightSample s = sample_light_triangles(ld3(hit_point), state, &sc);
    st3(out_dir, s.direction);
    st3(out_emission, s.emission);
    *out_distance = s.distance;
    return s.pdf;
}

void orc_sample_cosine_hemisphere(const float normal[3], uint32_t *state, float out[3])
{
    st3(out, sample_cosine_hemisphere(ld3(normal), state));
}

uint32_t orc_primary_ray(const void *uniforms80, int px, int py, float origin[3], float dir[3])
{
    UniV u = load_uniforms(uniforms80);
    v3 ro, rd;
    uint32_t state = primary_ray(&u, px, py, &ro, &rd);
    st3(origin, ro);
    st3(dir, rd);
    return state;
}

void orc_ray_trace(const float origin[3], const float dir[3], uint32_t max_bounces,
                   uint32_t *state, const void *spheres, uint32_t n_spheres,
                   const void *tris, uint32_t n_tris, float out_light[3],
                   uint64_t *n_rays, int32_t *first_id)
{
    SceneV sc = {(const unsigned char *)spheres, n_spheres, (const unsigned char *)tris, n_tris, 0, 0};
    uint64_t local = 0;
    int32_t fid = -1;
    st3(out_light, ray_trace(ld3(origin), ld3(dir), max_bounces, state, &sc, &local, &fid));
    if (n_rays) *n_rays += local;
    if (first_id) *first_id = fid;
}
