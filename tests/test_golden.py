"""Committed image fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py
from the CPU oracle).  The CPU test keeps the oracle and the scene loader from
drifting; the GPU test checks the HIP path against the same files, so it needs
nothing but the repository."""
import os

import numpy as np
import pytest

from raytracer_amd import scene as S
from tests.golden.make_golden import CASES, render

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_golden(case):
    name, scene_name, w, h, frames, spp = case
    gold = np.load(os.path.join(HERE, name + ".npz"))
    out = render(scene_name, w, h, frames, spp)
    for key in ("accum", "ids", "frame_f16", "rays"):
        assert np.array_equal(out[key], gold[key]), key


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_reproduces_golden(case):
    from raytracer_amd.hip_renderer import HipRenderer

    name, scene_name, w, h, frames, spp = case
    gold = np.load(os.path.join(HERE, name + ".npz"))
    r = HipRenderer.create(w, h, S.scene_path(scene_name))
    assert r is not None
    r.set_samples_per_pixel(spp)
    for _ in range(frames):
        r.render()
    assert np.array_equal(r.read_ids(), gold["ids"].astype(np.int32))
    assert np.array_equal(r.read_frame_f16(), gold["frame_f16"])
    assert np.array_equal(r.read_accum(), gold["accum"])
    assert int(r.ctx.read_counters()[0]) == int(gold["rays"][0])
    r.close()
