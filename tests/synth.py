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
