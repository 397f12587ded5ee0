/*
 * CPU oracle for the guymor4/raytracer hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's WGSL path tracer and accumulator:
 *   public/common.wgsl            (structs, rand_f, wang_hash, intersections, sampling)
 *   public/raytracer_compute.wgsl (ray_all, sample_light_triangles, ray_trace, main)
 *   public/accumulator.wgsl:44-68 (running-mean recurrence)
 * plus the host sequencing of src/webgpu-renderer.ts:214-268.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer
 * vectors, and no WGSL runtime exists in the build image, so this restatement
 * cannot be checked against outputs of the reference itself.  It is pinned only
 * by hand-computed known answers (tests/test_oracle_*.py) and by an independent
 * second restatement in numpy (oracle/wgsl_numpy.py).
 *
 * Nothing in the shipped renderer may include, link or call this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Floating-point contract of the restatement (WGSL leaves these open):
 *   - every f32 operation is a single IEEE-754 binary32 operation, round to
 *     nearest even, no FMA contraction (build with -ffp-contract=off);
 *   - division and sqrt are correctly rounded;
 *   - dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;
 *   - sin/cos/tan are the fixed polynomial routines orc_sincosf below;
 *   - f32 -> f16 (rgba16float store) rounds to nearest even;
 *   - min/max return the non-NaN operand.
 */
#ifndef WGSL_ORACLE_H
#define WGSL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Byte sizes of the three GPU buffers (src/gpu-schemas.ts:3-38, WGSL layout rules). */
#define ORC_SPHERE_STRIDE 64
#define ORC_TRIANGLE_STRIDE 96
#define ORC_UNIFORMS_SIZE 80

/* common.wgsl:66-73 */
uint32_t orc_wang_hash(uint32_t s);
/* common.wgsl:60-64; advances *state */
float orc_rand_f(uint32_t *state);
/* fixed sin/cos used wherever the WGSL calls sin(), cos(), tan() */
void orc_sincosf(float x, float *s, float *c);
/* f32 -> f16 bits (RNE) and back */
uint16_t orc_f32_to_f16(float f);
float orc_f16_to_f32(uint16_t h);

/* One intersection result: common.wgsl:50-57 plus the id AOV
 * (index in ray_all order: spheres 0..S-1, triangles S..S+T-1, miss -1). */
typedef struct {
    float t;
    float color[3];
    float normal[3];
    float emission[3];
    float smoothness;
    float specularProbability;
    int32_t id;
} OrcHit;

/* common.wgsl:137-168; sphere points at one 64-byte record */
void orc_ray_sphere_intersect(const float origin[3], const float dir[3],
                              const void *sphere, OrcHit *out);
/* common.wgsl:170-211; tri points at one 96-byte record */
void orc_ray_triangle_intersect(const float origin[3], const float dir[3],
                                const void *tri, OrcHit *out);
/* raytracer_compute.wgsl:15-39 */
void orc_ray_all(const float origin[3], const float dir[3],
                 const void *spheres, uint32_t n_spheres,
                 const void *tris, uint32_t n_tris, OrcHit *out);

/* raytracer_compute.wgsl:42-107.  out4 = {dir.xyz, distance}, out_em = emission,
 * returns pdf. */
float orc_sample_light_triangles(const float hit_point[3], uint32_t *state,
                                 const void *tris, uint32_t n_tris,
                                 float out_dir[3], float out_emission[3],
                                 float *out_distance);

/* common.wgsl:107-134 */
void orc_sample_cosine_hemisphere(const float normal[3], uint32_t *state, float out[3]);

/* Camera ray for one pixel: raytracer_compute.wgsl:290-329.  Returns the rng
 * state after the two jitter draws. */
uint32_t orc_primary_ray(const void *uniforms80, int px, int py,
                         float origin[3], float dir[3]);

/* raytracer_compute.wgsl:189-278; rays (ray_all calls) are added to *n_rays,
 * *first_id receives the id of the first ray_all hit. */
void orc_ray_trace(const float origin[3], const float dir[3], uint32_t max_bounces,
                   uint32_t *state, const void *spheres, uint32_t n_spheres,
                   const void *tris, uint32_t n_tris, float out_light[3],
                   uint64_t *n_rays, int32_t *first_id);

/*
 * One frame = compute pass (raytracer_compute.wgsl:280-341) followed by the
 * accumulator pass (accumulator.wgsl:44-68) over the pixel rectangle
 * [x0,x1) x [y0,y1) of a W x H image (W,H,frameIndex,spp come from the
 * uniforms blob, exactly as the shader reads them).
 *
 *   accum      in/out, planar: R plane, G plane, B plane, each W*H f32
 *   frame_f16  out, optional (may be NULL): W*H*4 half values (rgba16float)
 *   ids        out, optional: W*H int32 primary-hit ids
 *   counters   in/out, optional: [0] += number of ray_all calls
 *   n_threads  worker threads (>=1)
 * Pixels outside the rectangle are left untouched.  Returns 0 on success.
 */
int orc_render_frame(const void *uniforms80,
                     const void *spheres, uint32_t n_spheres,
                     const void *tris, uint32_t n_tris,
                     uint32_t max_bounces,
                     int x0, int y0, int x1, int y1,
                     float *accum, uint16_t *frame_f16, int32_t *ids,
                     uint64_t *counters, int n_threads);

/*
 * orc_render_frame with options.  ORC_SPHERE_LIGHTS: every bounce also samples the emissive
 * spheres with sample_light_spheres (raytracer_compute.wgsl:109-171, defined there and
 * never called) through a second shadow ray, wired like the triangle-light block at
 * :211-225.  The reference does not do this (its TODO at :226); it is an extension that
 * renderer and oracle share, off by default.
 *
 * ORC_DEBUG_VIEW: while uniforms.debugEnabled is non-zero, ray_all also tests, right after each
 * triangle, the marker sphere of ray_triangle_normal_intersect (common.wgsl:235-240, defined there
 * and never called): radius 0.1, 0.2 along the normal from the triangle's centre, coloured by the
 * normal.  Marker i reports id n_spheres + n_tris + i.  Also an extension shared by renderer and
 * oracle, off by default: the reference writes debugEnabled (src/webgpu-renderer.ts:228) and no
 * shader reads it.
 */
#define ORC_SPHERE_LIGHTS 1u
#define ORC_DEBUG_VIEW 2u
int orc_render_frame_ex(const void *uniforms80,
                        const void *spheres, uint32_t n_spheres,
                        const void *tris, uint32_t n_tris,
                        uint32_t max_bounces,
                        int x0, int y0, int x1, int y1,
                        float *accum, uint16_t *frame_f16, int32_t *ids,
                        uint64_t *counters, int n_threads, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif
