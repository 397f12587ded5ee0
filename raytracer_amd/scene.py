"""Scene loading and GPU-buffer packing (Python host mirror).

Restates, for the Python host (bench.py, tests), what the reference does in
TypeScript:

* ``parse_obj`` / ``load_scene``   -- src/common.ts:45-148
* ``pack_spheres`` / ``pack_triangles`` / ``pack_uniforms``
                                  -- src/gpu-schemas.ts:3-38 as written by
                                     src/webgpu-renderer.ts:146-187,214-230

The byte layouts are the WGSL ones: Sphere stride 64 B, Triangle stride 96 B,
Uniforms 80 B.  Numbers go double -> float32 with round-to-nearest-even, which
is what a ``Float32Array`` store does in the reference.
"""
from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

SPHERE_STRIDE = 64
TRIANGLE_STRIDE = 96
UNIFORMS_SIZE = 80

_FLOAT_RE = re.compile(r"^[+-]?(?:Infinity|(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?))")
_INT_RE = re.compile(r"^[+-]?\d+")


def js_parse_float(text: Optional[str]) -> float:
    """``parseFloat``: longest numeric prefix after leading blanks, else NaN."""
    if text is None:
        return float("nan")
    m = _FLOAT_RE.match(text.lstrip())
    if not m:
        return float("nan")
    tok = m.group(0)
    if tok.endswith("Infinity"):
        return float("-inf") if tok.startswith("-") else float("inf")
    return float(tok)


def js_parse_int(text: str) -> Optional[int]:
    """``parseInt(x)`` for base 10; ``None`` stands for NaN."""
    m = _INT_RE.match(text.lstrip())
    return int(m.group(0)) if m else None


@dataclass
class Material:
    color: Sequence[float] = (1.0, 1.0, 1.0)
    emissionColor: Sequence[float] = (0.0, 0.0, 0.0)
    emissionStrength: float = 0.0
    smoothness: float = 0.0
    specularProbability: float = 0.0


@dataclass
class Scene:
    """src/types.ts:48-52 -- camera, spheres, triangles as plain dicts."""

    camera: dict
    spheres: List[dict] = field(default_factory=list)
    triangles: List[dict] = field(default_factory=list)


def parse_obj(obj_text: str) -> List[dict]:
    """src/common.ts:45-109.

    Quirks kept: lines are split on ``\\n`` and trimmed; a face is split on a
    single space; each face vertex needs a position *and* a normal index
    (``v//vn`` or ``v/vt/vn``) or it is skipped; out-of-range indices are
    skipped; polygons are fan-triangulated.  The reference would push a
    triangle with missing vertices and fail later while packing; here that is
    reported immediately as ``ValueError``.
    """
    positions: List[List[float]] = []
    n_normals = 0
    triangles: List[dict] = []
    for raw in obj_text.split("\n"):
        line = raw.strip()
        if line.startswith("v "):
            parts = re.split(r"\s+", line)
            xyz = [js_parse_float(parts[i]) if i < len(parts) else float("nan") for i in (1, 2, 3)]
            positions.append(xyz)
        elif line.startswith("vn "):
            n_normals += 1
        elif line.startswith("f "):
            parts = line[2:].strip().split(" ")
            if len(parts) < 3:
                continue
            for i in range(1, len(parts) - 1):
                verts = []
                for fv in (parts[0], parts[i], parts[i + 1]):
                    fields = [js_parse_int(x) if x else None for x in fv.split("/")]
                    v_idx = fields[0] if len(fields) > 0 else None
                    n_idx = fields[2] if len(fields) > 2 else None
                    if v_idx is None or n_idx is None:
                        continue
                    if v_idx < 1 or v_idx > len(positions) or n_idx < 1 or n_idx > n_normals:
                        continue
                    verts.append(positions[v_idx - 1])
                if len(verts) != 3:
                    raise ValueError("OBJ face vertex without usable position/normal index: %r" % (line,))
                triangles.append(
                    dict(
                        v0=verts[0], v1=verts[1], v2=verts[2],
                        color=[1, 1, 1], emissionColor=[0, 0, 0],
                        emissionStrength=0, smoothness=0, specularProbability=0,
                    )
                )
    return triangles


def transform_point(p: Sequence[float], model: dict) -> List[float]:
    """A model vertex under the model's ``scale``, ``rotation`` (degrees about x, then y, then z) and
    ``position``, in that order.  The reference declares these fields (src/types.ts:34-39) and never
    applies them, so this is an extension with a convention of its own; see ``load_scene``."""
    import math

    sx, sy, sz = model.get("scale", [1, 1, 1])
    rx, ry, rz = (math.radians(a) for a in model.get("rotation", [0, 0, 0]))
    tx, ty, tz = model.get("position", [0, 0, 0])
    x, y, z = p[0] * sx, p[1] * sy, p[2] * sz
    y, z = y * math.cos(rx) - z * math.sin(rx), y * math.sin(rx) + z * math.cos(rx)
    x, z = x * math.cos(ry) + z * math.sin(ry), -x * math.sin(ry) + z * math.cos(ry)
    x, y = x * math.cos(rz) - y * math.sin(rz), x * math.sin(rz) + y * math.cos(rz)
    return [x + tx, y + ty, z + tz]


def load_scene(scene_path: str, apply_model_transforms: bool = False) -> Scene:
    """src/common.ts:111-148.

    Model paths resolve against the scene file's directory (the reference
    fetches them relative to the page, which serves ``public/``).  Model
    ``position``/``rotation``/``scale`` are read and ignored, as in the
    reference; only the five material fields are copied onto OBJ triangles.
    A model that fails to load is skipped (the reference logs and continues).

    ``apply_model_transforms=True`` is an opt-in extension: the model's vertices
    go through ``transform_point``.  Off by default because it changes the image
    (scene_suzanne.json asks for half size at (1, 0, -3); the reference shows the
    model full size at the origin).
    """
    with open(scene_path, "r") as fh:
        raw = json.load(fh)
    base = os.path.dirname(os.path.abspath(scene_path))
    triangles = list(raw.get("triangles", []))
    for model in raw.get("models") or []:
        path = model.get("path")
        if not path:
            continue
        try:
            with open(os.path.join(base, path), "r") as fh:
                obj_tris = parse_obj(fh.read())
        except (OSError, ValueError):
            continue
        for tri in obj_tris:
            t = dict(tri)
            if apply_model_transforms:
                for key in ("v0", "v1", "v2"):
                    t[key] = transform_point(t[key], model)
            for key in ("color", "emissionColor", "emissionStrength", "smoothness", "specularProbability"):
                t[key] = model[key]
            triangles.append(t)
    return Scene(camera=raw["camera"], spheres=list(raw.get("spheres", [])), triangles=triangles)


def pack_spheres(spheres: Sequence[dict]) -> np.ndarray:
    """Sphere records, shape (n, 16) float32 = 64 B each (src/gpu-schemas.ts:19-27)."""
    out = np.zeros((len(spheres), SPHERE_STRIDE // 4), dtype=np.float32)
    for i, s in enumerate(spheres):
        out[i, 0:3] = s["center"]
        out[i, 3] = s["radius"]
        out[i, 4:7] = s["color"]
        out[i, 7] = s["smoothness"]
        out[i, 8:11] = s["emissionColor"]
        out[i, 11] = s["emissionStrength"]
        out[i, 12] = s["specularProbability"]
    return out


def pack_triangles(triangles: Sequence[dict]) -> np.ndarray:
    """Triangle records, shape (n, 24) float32 = 96 B each (src/gpu-schemas.ts:29-38)."""
    out = np.zeros((len(triangles), TRIANGLE_STRIDE // 4), dtype=np.float32)
    for i, t in enumerate(triangles):
        out[i, 0:3] = t["v0"]
        out[i, 4:7] = t["v1"]
        out[i, 8:11] = t["v2"]
        out[i, 12:15] = t["color"]
        out[i, 16:19] = t["emissionColor"]
        out[i, 19] = t["emissionStrength"]
        out[i, 20] = t["smoothness"]
        out[i, 21] = t["specularProbability"]
    return out


def pack_uniforms(camera: dict, frame_index: float, width: int, height: int,
                  samples_per_pixel: float, debug_enabled: bool = False) -> np.ndarray:
    """The 80-byte Uniforms blob (src/gpu-schemas.ts:3-17, written by
    src/webgpu-renderer.ts:214-230).  frameIndex, samplesPerPixel and
    debugEnabled are stored as f32, as the reference does."""
    u = np.zeros(UNIFORMS_SIZE // 4, dtype=np.float32)
    u[0:3] = camera["position"]
    u[4:7] = camera["rotation"]
    u[7] = camera["fov"]
    u[8] = camera["nearPlane"]
    u[9] = camera["farPlane"]
    u[12] = frame_index
    u[14] = width
    u[15] = height
    u[16] = samples_per_pixel
    u[17] = 1.0 if debug_enabled else 0.0
    return u


def scene_dir() -> str:
    """Directory holding the reference's scene data files kept as fixtures."""
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(here), "tests", "golden", "scenes")


def scene_path(name: str) -> str:
    """``scene_path('suzanne')`` -> .../scene_suzanne.json"""
    fname = name if name.endswith(".json") else "scene_%s.json" % name
    return os.path.join(scene_dir(), fname)


def subdivided_suzanne_packed(levels: int, scale: float = 1.0):
    """scene_suzanne with the triangles of its OBJ model split at the edge midpoints
    `levels` times (968 * 4**levels triangles; 991,232 at 5), as packed records:
    returns (camera, spheres (n,16) float32, triangles (n,24) float32).  Room, light
    and sphere are kept.  Midpoints are taken in float64 and rounded to float32 once.

    With every split the triangles' determinant in ray_triangle_intersect shrinks by
    4; from about the third level most of them fall under the reference's
    abs(a) < 0.001 rejection (common.wgsl:179) and stop being visible, while they
    are still tested: a traversal-cost workload more than a picture.  `scale`
    multiplies every coordinate of the scene (geometry, sphere radius, camera
    position): at 32 the five-times-split triangles are as large against that
    threshold as the unsplit ones at 1, so the mesh is visible again and the
    picture is the original one up to the shader's absolute offsets."""
    import copy

    base = load_scene(scene_path("suzanne"))
    room = pack_triangles(base.triangles[:14])
    model = pack_triangles(base.triangles[14:])
    v = np.stack([model[:, 0:3], model[:, 4:7], model[:, 8:11]], axis=1).astype(np.float64)  # (n, 3 verts, 3)
    for _ in range(levels):
        a, b, c = v[:, 0], v[:, 1], v[:, 2]
        ab, bc, ca = 0.5 * (a + b), 0.5 * (b + c), 0.5 * (c + a)
        v = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1),
                            np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], axis=1).reshape(-1, 3, 3)
    out = np.repeat(model[:1], len(v), axis=0)
    out[:, 0:3], out[:, 4:7], out[:, 8:11] = v[:, 0], v[:, 1], v[:, 2]
    tris = np.concatenate([room, out], axis=0)
    spheres = pack_spheres(base.spheres)
    camera = base.camera
    if scale != 1.0:
        k = np.float32(scale)
        for c0 in (0, 4, 8):
            tris[:, c0:c0 + 3] *= k
        spheres[:, 0:4] *= k          # centre and radius
        camera = copy.deepcopy(base.camera)
        camera["position"] = [float(np.float32(x) * k) for x in camera["position"]]
    return camera, spheres, tris
