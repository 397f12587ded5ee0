"""Deterministic synthetic scenes for tests: a height-field mesh inside the
reference's box room and other constructed scenes (no random numbers), and
seeded random scenes (random_scene)."""
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


from raytracer_amd.scene import subdivided_suzanne_packed  # noqa: E402,F401  (the generator of BASELINE's synthetic configuration)


def edge_scenes() -> dict:
    """Scenes that exercise what the reference's three scene files do not: empty inputs, no
    lights, emissive and specular spheres, a rotated camera, coincident and degenerate
    triangles, many spheres.  Built from scene_boxes / scene_spheres data, no random numbers."""
    import copy

    boxes = S.load_scene(S.scene_path("boxes"))
    spheres = S.load_scene(S.scene_path("spheres"))
    out = {}
    out["empty"] = S.Scene(camera=boxes.camera)
    out["spheres_only"] = S.Scene(camera=spheres.camera, spheres=spheres.spheres)
    out["no_lights"] = S.Scene(camera=boxes.camera, spheres=boxes.spheres,
                               triangles=[t for t in boxes.triangles if not t["emissionStrength"] > 0])
    cam = copy.deepcopy(spheres.camera)
    cam["rotation"] = [12.0, -25.0, 0.0]
    cam["position"] = [1.5, 2.2, 4.0]
    out["rotated_camera"] = S.Scene(camera=cam, spheres=spheres.spheres, triangles=spheres.triangles)
    cam = copy.deepcopy(spheres.camera)
    cam["rotation"] = [-89.5, 181.0, 7.0]      # looks almost straight down; roll component is ignored by the shader
    cam["position"] = [0.0, 4.5, -2.5]
    cam["fov"] = 75
    out["camera_down"] = S.Scene(camera=cam, spheres=spheres.spheres, triangles=spheres.triangles)
    sp = copy.deepcopy(spheres.spheres)
    sp[0].update(emissionColor=[1.0, 0.6, 0.2], emissionStrength=3.0)
    sp[1].update(smoothness=1.0, specularProbability=1.0)
    out["emissive_and_mirror_spheres"] = S.Scene(camera=spheres.camera, spheres=sp, triangles=spheres.triangles)
    # every triangle twice (the first copy must win every tie), plus zero-area and needle triangles
    tris = []
    for t in boxes.triangles:
        tris += [t, copy.deepcopy(t)]
    tris.append(dict(boxes.triangles[0], v0=[0, 1, -2], v1=[0, 1, -2], v2=[0, 1, -2]))
    tris.append(dict(boxes.triangles[0], v0=[-1, 1, -2], v1=[0, 1, -2], v2=[1, 1, -2]))
    tris.append(dict(boxes.triangles[0], v0=[-1, 0.5, -2], v1=[1, 0.5, -2], v2=[0, 0.5004, -2]))
    out["duplicates_and_degenerates"] = S.Scene(camera=boxes.camera, spheres=boxes.spheres, triangles=tris)
    many = []
    for k in range(24):
        many.append(dict(center=[-2.0 + 0.35 * (k % 12), 0.3 + 0.9 * (k // 12), -2.0 - 0.15 * k], radius=0.16 + 0.01 * (k % 5),
                         color=[0.3 + 0.05 * (k % 7), 0.8 - 0.03 * k, 0.4], smoothness=0.1 * (k % 10),
                         emissionColor=[1, 1, 1], emissionStrength=2.0 if k == 5 else 0.0,
                         specularProbability=0.05 * (k % 20)))
    # two coincident spheres: the lower index must be reported
    many.append(copy.deepcopy(many[3]))
    out["many_spheres"] = S.Scene(camera=boxes.camera, spheres=many, triangles=boxes.triangles)
    # every triangle of the boxes scene three times, in a seeded shuffle: every hit is an exact tie in t
    # between copies whose original order and order in the acceleration structure differ; the lowest
    # original index must be reported, as by the reference's scan
    import random
    rng = random.Random(20240611)
    tripled = [copy.deepcopy(t) for t in boxes.triangles for _ in range(3)]
    rng.shuffle(tripled)
    out["shuffled_duplicates"] = S.Scene(camera=boxes.camera, spheres=boxes.spheres, triangles=tripled)
    return out


def card_deck_scene(n_cards: int) -> S.Scene:
    """scene_suzanne's room and light with its model replaced by n_cards parallel quads stacked along the
    view axis, all facing the camera: every box of the tree overlaps every central ray, so the first
    descent of a camera ray leaves siblings on the traversal stack at every level (the deepest stacks a
    tree of that size can produce), and the nearest card must still win."""
    base = S.load_scene(S.scene_path("suzanne"))
    tris = list(base.triangles[:14])
    z_front, z_back = -0.6, -4.4
    for k in range(n_cards):
        z = z_front + (z_back - z_front) * k / max(n_cards - 1, 1)
        shrink = 0.2 * k / max(n_cards - 1, 1)              # cards further back are slightly smaller
        x0, x1, y0, y1 = -1.2 + shrink, 1.2 - shrink, 0.3 + shrink, 2.3 - shrink
        mat = dict(color=[0.3 + 0.6 * ((k * 7) % 10) / 10.0, 0.5, 0.9 - 0.5 * ((k * 3) % 10) / 10.0], emissionColor=[0, 0, 0],
                   emissionStrength=0, smoothness=0.0, specularProbability=0.0)
        a, b, c, d = [x0, y0, z], [x1, y0, z], [x1, y1, z], [x0, y1, z]
        tris.append(dict(v0=a, v1=b, v2=c, **mat))           # counter-clockwise seen from +z: faces the camera
        tris.append(dict(v0=a, v1=c, v2=d, **mat))
    return S.Scene(camera=base.camera, spheres=base.spheres, triangles=tris)


def sphere_field_scene(n: int, with_triangles: bool = True) -> S.Scene:
    """n spheres on a deterministic lattice inside scene_boxes' room (more than a ray tests one by one, so they
    become leaves of the acceleration structure), of varied size and material, some emissive, some mirrors, the
    last two coincident with earlier ones (ties in t: the lower index must be reported), one enclosing the
    camera's near field partly (the t2 branch of the sphere test).  No random numbers."""
    import copy

    boxes = S.load_scene(S.scene_path("boxes"))
    spheres = []
    for k in range(n):
        i, j, l = k % 7, (k // 7) % 5, k // 35
        spheres.append(dict(center=[-2.1 + 0.7 * i + 0.05 * l, 0.25 + 0.75 * j, -0.8 - 0.9 * l - 0.1 * i],
                            radius=0.08 + 0.02 * ((k * 5) % 9),
                            color=[0.25 + 0.1 * (k % 7), 0.9 - 0.08 * (k % 9), 0.35 + 0.05 * (k % 11)],
                            smoothness=0.125 * (k % 9), emissionColor=[1.0, 0.8, 0.6],
                            emissionStrength=2.5 if k % 23 == 4 else 0.0, specularProbability=0.1 * (k % 11)))
    if n >= 12:
        spheres[-1] = copy.deepcopy(spheres[3])
        spheres[-2] = copy.deepcopy(spheres[10])
        spheres[5]["center"] = [0.0, 2.5, 4.0]   # one unit in front of the camera (scene_boxes: position 0, 2.5, 5)
        spheres[5]["radius"] = 0.995
    return S.Scene(camera=boxes.camera, spheres=spheres, triangles=boxes.triangles if with_triangles else [])



def random_scene(seed: int, n_tris: int = -1) -> S.Scene:
    """A seeded random scene: 0..14 spheres and 0..400 triangles of random size, place and material inside a
    6 x 4 x 6 volume in front of a randomly placed and rotated camera; about one primitive in eight emits, one in
    five is a mirror or partly one; clusters of small triangles next to wall-sized ones, a few exact duplicates,
    and now and then a primitive around or behind the camera.  n_tris overrides the random triangle count (the kernels
    for scenes of more than 4096 triangles).  Same seed and count, same scene."""
    import random

    rng = random.Random(seed)

    def material():
        emits = rng.random() < 0.125
        shiny = rng.random() < 0.2
        return dict(color=[rng.random(), rng.random(), rng.random()],
                    emissionColor=[rng.random(), rng.random(), rng.random()] if emits else [0.0, 0.0, 0.0],
                    emissionStrength=rng.uniform(0.5, 6.0) if emits else 0.0,
                    smoothness=rng.random() if shiny else 0.0,
                    specularProbability=rng.choice([1.0, rng.random()]) if shiny else 0.0)

    def point(spread=1.0):
        return [rng.uniform(-3.0, 3.0) * spread, rng.uniform(0.0, 4.0) * spread, rng.uniform(-6.0, 0.0) * spread]

    spheres = []
    for _ in range(rng.choice([0, 1, 2, 3, 8, 9, 14])):
        c = point()
        spheres.append(dict(center=c, radius=rng.choice([0.05, 0.3, 0.8, 2.5]) * rng.uniform(0.5, 1.5), **material()))
    triangles = []
    drawn = rng.choice([0, 1, 7, 40, 150, 400])
    n_tris = drawn if n_tris < 0 else n_tris
    while len(triangles) < n_tris:
        size = rng.choice([0.02, 0.2, 1.0, 5.0])
        a = point()
        b = [a[k] + rng.uniform(-size, size) for k in range(3)]
        c = [a[k] + rng.uniform(-size, size) for k in range(3)]
        m = material()
        triangles.append(dict(v0=a, v1=b, v2=c, **m))
        if rng.random() < 0.5:                      # the same triangle facing the other way
            triangles.append(dict(v0=a, v1=c, v2=b, **m))
        if rng.random() < 0.03:                     # an exact duplicate: the first must win the tie
            triangles.append(dict(triangles[-1]))
    if spheres and rng.random() < 0.3:              # a sphere around the camera
        spheres[0]["center"], spheres[0]["radius"] = [0.2, 2.0, 3.1], 1.5
    camera = dict(position=[rng.uniform(-1.0, 1.0), rng.uniform(1.0, 3.0), rng.uniform(2.0, 4.0)],
                  rotation=[rng.uniform(-20.0, 10.0), rng.uniform(-25.0, 25.0), 0.0],
                  fov=rng.uniform(30.0, 90.0), nearPlane=0.1, farPlane=100.0)
    return S.Scene(camera=camera, spheres=spheres, triangles=triangles)
