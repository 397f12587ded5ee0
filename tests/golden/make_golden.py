#!/usr/bin/env python3
"""Generates the image fixtures in tests/golden/ with the CPU oracle.

The reference ships no golden images (it has no tests), so these fixtures pin
the oracle against itself: a later change to oracle/ or to the scene loader
that alters any bit of these outputs fails tests/test_golden.py.  Run from the
repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracer_amd import scene as S  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

# name, scene, width, height, frames, spp
CASES = [
    ("spheres_64x36_1f_16spp", "spheres", 64, 36, 1, 16),
    ("spheres_64x36_4f_1spp", "spheres", 64, 36, 4, 1),
    ("boxes_160x90_4f_1spp", "boxes", 160, 90, 4, 1),
    ("suzanne_160x90_4f_1spp", "suzanne", 160, 90, 4, 1),
]


def render(scene_name, width, height, frames, spp):
    orc = O.OracleRenderer(S.load_scene(S.scene_path(scene_name)), width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(frames)
    return dict(accum=orc.accum_rgb(), ids=orc.ids.astype(np.int16), frame_f16=orc.frame_f16,
                rays=np.array([orc.counters[0]], dtype=np.uint64))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, scene_name, w, h, frames, spp in CASES:
        out = render(scene_name, w, h, frames, spp)
        np.savez_compressed(os.path.join(here, name + ".npz"), **out)
        print(name, "rays", int(out["rays"][0]))
