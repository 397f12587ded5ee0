"""Second, independent restatement of the reference shaders in Python with
numpy float32 scalars -- TEST INFRASTRUCTURE ONLY (see wgsl_oracle.h).

Purpose: the reference has no tests and no WGSL runtime exists here, so the C
oracle (wgsl_oracle.c) cannot be pinned against reference output.  This file
restates the same WGSL text a second time, in another language and with a
different code structure (vectors are numpy arrays, loops are Python), and
tests/test_oracle_cross.py requires both restatements to agree bit for bit on
small images.  A slip in transcribing one operation into one of them shows up
as a mismatch.  Every function names the WGSL lines it follows.

Same floating-point contract as the C oracle: every operation is one IEEE
binary32 operation in the order written; dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;
sin/cos are the fixed polynomial routine `sincos`.
"""
import numpy as np

f32 = np.float32
u32mask = 0xFFFFFFFF
PI = f32(3.14159265358979323846)  # common.wgsl:4


def vec3(x, y, z):
    return np.array([x, y, z], dtype=np.float32)


def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def cross(a, b):
    return vec3(f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]),
                f32(a[0] * b[1]) - f32(a[1] * b[0]))


def length(a):
    return f32(np.sqrt(dot(a, a)))


def normalize(a):
    with np.errstate(divide="ignore", invalid="ignore"):
        return (a / length(a)).astype(np.float32)


def wang_hash(s):  # common.wgsl:66-73
    ns = ((s ^ 61) ^ (s >> 16)) & u32mask
    ns = (ns * 9) & u32mask
    ns = ns ^ (ns >> 4)
    ns = (ns * 0x27D4EB2D) & u32mask
    ns = ns ^ (ns >> 15)
    return ns


class Rng:
    def __init__(self, state):
        self.state = state

    def rand_f(self):  # common.wgsl:60-64
        self.state = (self.state * 747796405 + 2891336453) & u32mask
        word = (((self.state >> ((self.state >> 28) + 4)) ^ self.state) * 277803737) & u32mask
        scale = np.array([0x2F800004], dtype=np.uint32).view(np.float32)[0]
        return f32(f32((word >> 22) ^ word) * scale)


def sincos(x):
    """Cody-Waite reduction by pi/2 in three parts + cephes polynomials (contract of the oracle)."""
    x = f32(x)
    k = f32(np.rint(f32(x * f32(0.636619772367581343))))
    q = int(k)
    r = f32(x - f32(k * f32(1.5703125)))
    r = f32(r - f32(k * f32(4.837512969970703125e-4)))
    r = f32(r - f32(k * f32(7.54978995489188216e-8)))
    z = f32(r * r)
    ps = f32(-1.9515295891e-4)
    ps = f32(f32(ps * z) + f32(8.3321608736e-3))
    ps = f32(f32(ps * z) - f32(1.6666654611e-1))
    sn = f32(r + f32(r * f32(z * ps)))
    pc = f32(2.443315711809948e-5)
    pc = f32(f32(pc * z) - f32(1.388731625493765e-3))
    pc = f32(f32(pc * z) + f32(4.166664568298827e-2))
    cs = f32(f32(f32(1.0) - f32(f32(0.5) * z)) + f32(f32(z * z) * pc))
    m = q & 3
    if m == 0:
        return sn, cs
    if m == 1:
        return cs, f32(-sn)
    if m == 2:
        return f32(-sn), f32(-cs)
    return f32(-cs), sn


class Hit:  # common.wgsl:50-57
    def __init__(self, t=f32(-1.0), color=None, normal=None, emission=None, smoothness=f32(0), spec=f32(0), ident=-1):
        z = vec3(0, 0, 0)
        self.t, self.color, self.normal = f32(t), z if color is None else color, z if normal is None else normal
        self.emission = z if emission is None else emission
        self.smoothness, self.specularProbability, self.id = f32(smoothness), f32(spec), ident


def ray_sphere_intersect(ro, rd, s):  # common.wgsl:137-168; s = 16 floats
    center, radius = s[0:3], s[3]
    oc = ro - center
    a = dot(rd, rd)
    b = f32(f32(2.0) * dot(oc, rd))
    c = f32(dot(oc, oc) - f32(radius * radius))
    disc = f32(f32(b * b) - f32(f32(f32(4.0) * a) * c))
    if disc < 0:
        return Hit()
    sq = f32(np.sqrt(disc))
    t1 = f32(f32(f32(-b) - sq) / f32(f32(2.0) * a))
    t2 = f32(f32(f32(-b) + sq) / f32(f32(2.0) * a))
    emission = (s[8:11] * s[11]).astype(np.float32)
    if t1 > f32(0.01):
        n = normalize((ro + rd * t1) - center)
        return Hit(t1, s[4:7], n, emission, s[7], s[12])
    if t2 > f32(0.01):
        n = normalize((ro + rd * t2) - center)
        return Hit(t1, s[4:7], n, emission, s[7], s[12])  # common.wgsl:164 returns t1
    return Hit()


def ray_triangle_intersect(ro, rd, tr):  # common.wgsl:170-211; tr = 24 floats
    v0, v1, v2 = tr[0:3], tr[4:7], tr[8:11]
    e01 = v1 - v0
    e02 = v2 - v0
    h = cross(rd, e02)
    a = dot(e01, h)
    if abs(a) < f32(0.001):
        return Hit()
    f = f32(f32(1.0) / a)
    s = ro - v0
    u = f32(f * dot(s, h))
    if u < 0 or u > 1:
        return Hit()
    q = cross(s, e01)
    v = f32(f * dot(rd, q))
    if v < 0 or f32(u + v) > 1:
        return Hit()
    t = f32(f * dot(e02, q))
    if t < f32(0.01):
        return Hit()
    n = normalize(cross(e01, e02))
    if dot(n, rd) > 0:
        return Hit()
    return Hit(t, tr[12:15], n, (tr[16:19] * tr[19]).astype(np.float32), tr[20], tr[21])


def ray_triangle_normal_intersect(ro, rd, tr):  # common.wgsl:235-240 (never called by the shader)
    center = (((tr[0:3] + tr[4:7]) + tr[8:11]) / f32(3.0)).astype(np.float32)
    n = normalize(cross(tr[4:7] - tr[0:3], tr[8:11] - tr[0:3]))
    s = np.zeros(16, dtype=np.float32)
    s[0:3] = center + n * f32(0.2)
    s[3] = f32(0.1)
    s[4:7] = n
    return ray_sphere_intersect(ro, rd, s)


class Scene:
    def __init__(self, spheres, tris, debug_view=False):
        self.spheres = np.asarray(spheres, dtype=np.float32).reshape(-1, 16)
        self.tris = np.asarray(tris, dtype=np.float32).reshape(-1, 24)
        self.rays = 0
        self.debug_view = debug_view   # extension: ray_all also tests every triangle's normal marker


def ray_all(sc, ro, rd):  # raytracer_compute.wgsl:15-39
    sc.rays += 1
    closest = Hit()
    for i, s in enumerate(sc.spheres):
        h = ray_sphere_intersect(ro, rd, s)
        if h.t > 0 and (closest.t < 0 or h.t < closest.t):
            closest = h
            closest.id = i
    for i, tr in enumerate(sc.tris):
        h = ray_triangle_intersect(ro, rd, tr)
        if h.t > 0 and (closest.t < 0 or h.t < closest.t):
            closest = h
            closest.id = len(sc.spheres) + i
        if sc.debug_view:   # extension: the marker follows its triangle, same rule; marker ids follow the triangles'
            h = ray_triangle_normal_intersect(ro, rd, tr)
            if h.t > 0 and (closest.t < 0 or h.t < closest.t):
                closest = h
                closest.id = len(sc.spheres) + len(sc.tris) + i
    return closest


def luminance(c):  # raytracer_compute.wgsl:11-13
    return dot(c, vec3(0.2126, 0.7152, 0.0722))


def triangle_area(tr):  # common.wgsl:228-232
    return f32(f32(0.5) * length(cross(tr[4:7] - tr[0:3], tr[8:11] - tr[0:3])))


def sample_light_triangles(sc, hit_point, rng):  # raytracer_compute.wgsl:42-107
    zero = (vec3(0, 0, 0), vec3(0, 0, 0), f32(0), f32(0))
    total_power = f32(0)
    light_count = 0
    for tr in sc.tris:
        if tr[19] <= 0:
            continue
        power = f32(f32(tr[19] * triangle_area(tr)) * luminance(tr[16:19]))
        total_power = f32(total_power + power)
        light_count += 1
    if light_count == 0 or total_power <= 0:
        return zero
    random_power = f32(rng.rand_f() * total_power)
    acc = f32(0)
    sel = np.zeros(24, dtype=np.float32)
    found = False
    for tr in sc.tris:
        if tr[19] > 0:
            power = f32(f32(tr[19] * triangle_area(tr)) * luminance(tr[16:19]))
            acc = f32(acc + power)
            if acc >= random_power and not found:
                sel = tr
                found = True
    # common.wgsl:214-225
    r1 = rng.rand_f()
    r2 = rng.rand_f()
    sqrt_r1 = f32(np.sqrt(r1))
    u = f32(f32(1.0) - sqrt_r1)
    v = f32(r2 * sqrt_r1)
    w = f32(f32(f32(1.0) - u) - v)
    light_point = ((sel[0:3] * u) + (sel[4:7] * v)) + (sel[8:11] * w)
    light_direction = light_point - hit_point
    distance = length(light_direction)
    with np.errstate(divide="ignore", invalid="ignore"):
        nd = (light_direction / distance).astype(np.float32)
    light_normal = normalize(cross(sel[4:7] - sel[0:3], sel[8:11] - sel[0:3]))
    cos_light = f32(-dot(light_normal, nd))
    if cos_light <= 0:
        return zero
    area = triangle_area(sel)
    power = f32(f32(sel[19] * area) * luminance(sel[16:19]))
    with np.errstate(divide="ignore", invalid="ignore"):
        pdf = f32(f32(f32(distance * distance) / f32(area * cos_light)) * f32(power / total_power))
    return nd, (sel[16:19] * sel[19]).astype(np.float32), distance, pdf


def sample_light_spheres(sc, hit_point, rng):  # raytracer_compute.wgsl:109-171 (defined in the shader, never called there)
    zero = (vec3(0, 0, 0), vec3(0, 0, 0), f32(0), f32(0))
    lights = [s for s in sc.spheres if s[11] > 0]
    if not lights:
        return zero
    index = int(f32(rng.rand_f() * f32(len(lights))))        # u32(rand * count); count itself when rand_f exceeds 1
    sel = lights[index] if index < len(lights) else np.zeros(16, dtype=np.float32)   # zero-initialised var
    to_center = sel[0:3] - hit_point
    dist = length(to_center)
    if dist <= sel[3]:
        return zero
    with np.errstate(divide="ignore", invalid="ignore"):
        cdir = (to_center / dist).astype(np.float32)
        up = vec3(1, 0, 0) if abs(cdir[1]) > f32(0.99) else vec3(0, 1, 0)
        right = normalize(cross(up, cdir))
        forward = cross(cdir, right)
        cos_max = f32(np.sqrt(f32(f32(1.0) - f32(f32(sel[3] * sel[3]) / f32(dist * dist)))))
        cos_t = f32(f32(1.0) - f32(rng.rand_f() * f32(f32(1.0) - cos_max)))
        sin_t = f32(np.sqrt(f32(f32(1.0) - f32(cos_t * cos_t))))
        phi = f32(f32(f32(2.0) * PI) * rng.rand_f())
        sphi, cphi = sincos(phi)
        direction = normalize(((cdir * cos_t) + (right * f32(sin_t * cphi))) + (forward * f32(sin_t * sphi)))
        pdf = f32(f32(1.0) / f32(f32(f32(f32(2.0) * PI) * f32(f32(1.0) - cos_max)) * f32(len(lights))))
    return direction, (sel[8:11] * sel[11]).astype(np.float32), f32(dist - sel[3]), pdf


def sample_cosine_hemisphere(normal, rng):  # common.wgsl:107-134
    r1 = rng.rand_f()
    r2 = rng.rand_f()
    phi = f32(f32(f32(2.0) * PI) * r1)
    cos_theta = f32(np.sqrt(r2))
    with np.errstate(invalid="ignore"):
        sin_theta = f32(np.sqrt(f32(f32(1.0) - r2)))
    sphi, cphi = sincos(phi)
    local = vec3(f32(cphi * sin_theta), cos_theta, f32(sphi * sin_theta))
    up = vec3(1, 0, 0) if abs(normal[1]) > f32(0.99) else vec3(0, 1, 0)
    tangent = normalize(cross(up, normal))
    bitangent = cross(normal, tangent)
    return normalize(((tangent * local[0]) + (normal * local[1])) + (bitangent * local[2]))


def power_heuristic(a, b):  # raytracer_compute.wgsl:174-178
    a2 = f32(a * a)
    b2 = f32(b * b)
    with np.errstate(divide="ignore", invalid="ignore"):
        return f32(a2 / f32(a2 + b2))


def brdf_pdf(n, d):  # raytracer_compute.wgsl:181-187
    c = dot(n, d)
    if c <= 0:
        return f32(0)
    return f32(c / PI)


def ray_trace(sc, ro, rd, max_bounces, rng, sphere_lights=False):  # raytracer_compute.wgsl:189-278
    sky = (vec3(1, 1, 1) * f32(0.4)).astype(np.float32)
    color = vec3(1, 1, 1)
    light = vec3(0, 0, 0)
    spec_bounce = False
    first_id = -1
    for i in range(max_bounces):
        hit = ray_all(sc, ro, rd)
        if i == 0:
            first_id = hit.id
        if hit.t < 0:
            light = light + sky * color
            break
        hit_point = ro + rd * hit.t
        ldir, lem, ldist, lpdf = sample_light_triangles(sc, hit_point, rng)
        if lpdf > 0:
            cos_theta = dot(hit.normal, ldir)
            if cos_theta > 0:
                sh = ray_all(sc, hit_point + hit.normal * f32(0.01), ldir)
                if sh.t < 0 or sh.t > f32(ldist - f32(0.1)):
                    mis = power_heuristic(lpdf, brdf_pdf(hit.normal, ldir))
                    brdf = f32(cos_theta / PI)
                    light = light + (((lem * brdf) * color) * mis) / lpdf
        if sphere_lights:   # extension: the sampling the shader's TODO at :226 asks for, wired like the block above
            sdir, sem, sdist, spdf = sample_light_spheres(sc, hit_point, rng)
            if spdf > 0:
                cos_theta = dot(hit.normal, sdir)
                if cos_theta > 0:
                    sh = ray_all(sc, hit_point + hit.normal * f32(0.01), sdir)
                    if sh.t < 0 or sh.t > f32(sdist - f32(0.1)):
                        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                            mis = power_heuristic(spdf, brdf_pdf(hit.normal, sdir))
                            brdf = f32(cos_theta / PI)
                            light = light + (((sem * brdf) * color) * mis) / spdf
        emissive = f32(f32(hit.emission[0] + hit.emission[1]) + hit.emission[2]) > f32(0.01)
        if (i == 0 and emissive) or spec_bounce:
            incoming = -normalize(rd)
            mis = power_heuristic(brdf_pdf(hit.normal, incoming), f32(0.001))
            light = light + (hit.emission * color) * mis
        color = color * hit.color
        if f32(f32(color[0] + color[1]) + color[2]) < f32(0.01):
            break
        if i >= 3:
            p = f32(min(max(luminance(color), f32(0.05)), f32(0.95)))
            if rng.rand_f() > p:
                break
            color = (color / p).astype(np.float32)
        spec_bounce = bool(hit.specularProbability >= rng.rand_f())
        diffuse = sample_cosine_hemisphere(hit.normal, rng)
        specular = rd - hit.normal * f32(f32(2.0) * dot(hit.normal, rd))  # reflect(e1,e2) = e1 - 2*dot(e2,e1)*e2
        mt = hit.smoothness if spec_bounce else f32(0)
        new_dir = normalize(diffuse * f32(f32(1.0) - mt) + specular * mt)  # mix
        ro = (ro + rd * hit.t) + hit.normal * f32(0.01)
        rd = new_dir
    return light.astype(np.float32), first_id


def rotate_yaw_pitch(v, yaw, pitch):  # common.wgsl:80-96
    sy, cy = sincos(yaw)
    sx, cx = sincos(pitch)
    r0 = vec3(cy, f32(sy * sx), f32(sy * cx))
    r1 = vec3(0, cx, f32(-sx))
    r2 = vec3(f32(-sy), f32(cy * sx), f32(cy * cx))
    return normalize(vec3(dot(r0, v), dot(r1, v), dot(r2, v)))


def f16_round_trip(x):
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=np.float32).astype(np.float16)


def render_frame(uniforms, spheres, tris, accum, max_bounces=6, sphere_lights=False, debug_view=False):
    """raytracer_compute.wgsl:280-341 + accumulator.wgsl:44-68 over the whole image.
    uniforms: 20 float32; accum: (3, H, W) float32 in/out.  Returns (frame_f16 (H,W,4) uint16,
    ids (H,W) int32, rays)."""
    u = np.asarray(uniforms, dtype=np.float32)
    cam_pos, cam_rot, fov_deg = u[0:3], u[4:7], u[7]
    frame_index, res_x, res_y, spp_f = u[12], u[14], u[15], u[16]
    W, H = int(res_x), int(res_y)
    sc = Scene(spheres, tris, debug_view=debug_view and u[17] != 0)
    frame = np.zeros((H, W, 4), dtype=np.uint16)
    ids = np.zeros((H, W), dtype=np.int32)
    aspect = f32(res_x / res_y)
    rot = ((cam_rot * f32(3.14159)) / f32(180.0)).astype(np.float32)
    forward = rotate_yaw_pitch(vec3(0, 0, -1), rot[1], rot[0])
    right = normalize(cross(forward, vec3(0, 1, 0)))
    up = cross(right, forward)
    fov = f32(f32(fov_deg * f32(3.14159)) / f32(180.0))
    sh, ch = sincos(f32(fov * f32(0.5)))
    focal = f32(f32(1.0) / f32(sh / ch))
    for py in range(H):
        for px in range(W):
            uvx = f32(f32(f32(px) + f32(0.5)) / res_x)
            uvy = f32(f32(f32(py) + f32(0.5)) / res_y)
            seed = (py * int(res_x) + px + int(frame_index) * 12345) & u32mask
            rng = Rng(wang_hash(seed))
            jx = f32(rng.rand_f() - f32(0.5))
            jy = f32(rng.rand_f() - f32(0.5))
            ox = f32(jx / res_x)
            oy = f32(jy / res_y)
            cx = f32(f32(f32(f32(uvx + ox) * f32(2.0)) - f32(1.0)) * aspect)
            cy = f32(f32(1.0) - f32(f32(uvy + oy) * f32(2.0)))
            rd = normalize(((right * cx) + (up * cy)) + (forward * focal))
            total = vec3(0, 0, 0)
            pid = -1
            for s in range(int(spp_f)):
                light, fid = ray_trace(sc, cam_pos.copy(), rd, max_bounces, rng, sphere_lights)
                total = total + light
                if s == 0:
                    pid = fid
            color = (total / spp_f).astype(np.float32)
            h = f16_round_trip(color)
            frame[py, px, 0:3] = h.view(np.uint16)
            frame[py, px, 3] = 0x3C00
            ids[py, px] = pid
            cur = h.astype(np.float32)
            stored = accum[:, py, px].copy() if frame_index > 0 else vec3(0, 0, 0)
            weight = f32(f32(1.0) / f32(frame_index + f32(1.0)))
            acc = stored * f32(f32(1.0) - weight) + cur * weight
            accum[:, py, px] = np.minimum(np.maximum(acc, f32(0)), f32(1)).astype(np.float32)
    return frame, ids, sc.rays
