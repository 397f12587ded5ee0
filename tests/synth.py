"""Deterministic synthetic scenes for tests: a height-field mesh inside the
reference's box room (no random numbers)."""
import math

import numpy as np

from raytracer_amd import scene as S


def heightfield_scene(n: int, amplitude: float = 0.6) -> S.Scene:
    """scene_suzanne's room, light and camera with its OBJ model replaced by an
    n x n grid of quads (2*n*n triangles) over the floor, raised by a smooth
    bump pattern.  Triangles stay large enough for the reference's
    abs(a) < 0.001 rejection (common.wgsl:179) not to hide them all."""
    base = S.load_scene(S.scene_path("suzanne"))
    tris = list(base.triangles[:14])
    x0, x1, z0, z1 = -2.4, 2.4, -4.9, -0.1

    def height(i, j):
        u, v = i / n, j / n
        return 0.3 + amplitude * (0.5 + 0.5 * math.sin(7.0 * u + 1.3) * math.cos(5.0 * v - 0.4))

    def vert(i, j):
        return [x0 + (x1 - x0) * i / n, height(i, j), z0 + (z1 - z0) * j / n]

    mat = dict(color=[0.8, 0.7, 0.3], emissionColor=[0, 0, 0], emissionStrength=0, smoothness=0.6,
               specularProbability=0.2)
    for j in range(n):
        for i in range(n):
            a, b, c, d = vert(i, j), vert(i + 1, j), vert(i + 1, j + 1), vert(i, j + 1)
            tris.append(dict(v0=a, v1=c, v2=b, **mat))   # wound to face up (+y)
            tris.append(dict(v0=a, v1=d, v2=c, **mat))
    return S.Scene(camera=base.camera, spheres=base.spheres, triangles=tris)


def subdivided_suzanne_packed(levels: int):
    """scene_suzanne with the triangles of its OBJ model split at the edge midpoints
    `levels` times (968 * 4**levels triangles; 991,232 at 5), as packed records:
    returns (camera, spheres (n,16) float32, triangles (n,24) float32).  Room, light
    and sphere are kept.  Midpoints are taken in float64 and rounded to float32 once.

    With every split the triangles' determinant in ray_triangle_intersect shrinks by
    4; from about the third level most of them fall under the reference's
    abs(a) < 0.001 rejection (common.wgsl:179) and stop being visible, while they
    are still tested: a traversal-cost workload more than a picture."""
    base = S.load_scene(S.scene_path("suzanne"))
    room = S.pack_triangles(base.triangles[:14])
    model = S.pack_triangles(base.triangles[14:])
    v = np.stack([model[:, 0:3], model[:, 4:7], model[:, 8:11]], axis=1).astype(np.float64)  # (n, 3 verts, 3)
    for _ in range(levels):
        a, b, c = v[:, 0], v[:, 1], v[:, 2]
        ab, bc, ca = 0.5 * (a + b), 0.5 * (b + c), 0.5 * (c + a)
        v = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1),
                            np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], axis=1).reshape(-1, 3, 3)
    out = np.repeat(model[:1], len(v), axis=0)
    out[:, 0:3], out[:, 4:7], out[:, 8:11] = v[:, 0], v[:, 1], v[:, 2]
    return base.camera, S.pack_spheres(base.spheres), np.concatenate([room, out], axis=0)
