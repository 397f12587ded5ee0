"""Node visits and triangle tests per traversed ray of the 4-wide tree, from the lane emulation on the host
(no GPU): the figure of merit of the builder.  usage: python tests/harness/tree_stats.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytracer_amd import scene as S
from tests.harness_lib import HarnessRenderer
from tests.synth import heightfield_scene

for name in sys.argv[1:] or ["suzanne", "boxes", "spheres", "hf64"]:
    sc = heightfield_scene(int(name[2:])) if name.startswith("hf") else S.load_scene(S.scene_path(name))
    har = HarnessRenderer(sc, 240, 135, max_bounces=8 if name.startswith("hf") else 6)
    har.render_frames_lanes(4, 1, 1, packed_links=True)
    c = [int(x) for x in har.counters]
    rays = c[13]
    print("%-8s triangles %6d wide nodes %6d | per traversed ray: node visits %.3f triangle tests %.3f | stack need %d" % (
        name, len(sc.triangles), c[15], c[1] / 2 / rays, c[2] / rays, c[14]))
