"""The renderer's shared core (traversal, shading, scene preparation) built
for the host and compared bit for bit with the oracle.  This checks the logic
the HIP kernels run, in a container without a GPU; it is not a CPU rendering
path of the product."""
import numpy as np
import pytest

from tests import oracle_lib as O
from tests.harness_lib import HarnessRenderer


def compare(scene, width, height, frames, spp, brute, max_bounces=6, max_leaf=0):
    orc = O.OracleRenderer(scene, width, height, max_bounces=max_bounces)
    orc.set_samples_per_pixel(spp)
    har = HarnessRenderer(scene, width, height, brute=brute, max_bounces=max_bounces, max_leaf=max_leaf)
    for _ in range(frames):
        orc.render()
        har.render(spp)
        assert np.array_equal(har.ids, orc.ids)
        assert np.array_equal(har.frame_f16, orc.frame_f16)
    assert np.array_equal(har.accum, orc.accum)
    assert int(har.counters[0]) == int(orc.counters[0])
    return har


@pytest.mark.parametrize("brute", [True, False])
@pytest.mark.parametrize("name,width,height,frames,spp", [
    ("spheres", 96, 54, 2, 4),
    ("boxes", 96, 54, 2, 4),
    ("suzanne", 64, 36, 2, 2),
])
def test_core_matches_oracle(scenes, brute, name, width, height, frames, spp):
    compare(scenes[name], width, height, frames, spp, brute)


@pytest.mark.parametrize("max_leaf", [1, 2, 8])
def test_leaf_size_does_not_change_results(scenes, max_leaf):
    har = compare(scenes["suzanne"], 64, 36, 1, 4, False, max_leaf=max_leaf)
    assert int(har.counters[6]) < 64  # tree depth within the traversal stack bound


def test_axis_parallel_ray_regression(scenes):
    # pixel (437,220) of spheres 640x360 frame 1 produces a bounce ray d = (-1, 0, 0); the slab
    # test must stay finite for zero direction components
    compare(scenes["spheres"], 640, 360, 1, 16, False)


def test_max_bounces_parameter(scenes):
    compare(scenes["boxes"], 48, 27, 1, 4, False, max_bounces=8)
    compare(scenes["boxes"], 48, 27, 1, 4, False, max_bounces=1)


@pytest.mark.parametrize("name,width,height,frames,spp,shade_num,leaf_lanes,max_bounces", [
    ("spheres", 100, 61, 3, 4, 3, 8, 6),
    ("boxes", 100, 61, 2, 4, 0, 1, 6),
    ("suzanne", 100, 61, 2, 4, 8, 64, 6),
    ("suzanne", 64, 36, 2, 4, 2, 16, 8),
    ("spheres", 64, 36, 2, 16, 3, 8, 1),
])
def test_lane_state_machine_matches_oracle(scenes, name, width, height, frames, spp, shade_num, leaf_lanes, max_bounces):
    """rt_lane.h driven by an emulated wave (the persistent kernel's schedule), frames batched."""
    scene = scenes[name]
    orc = O.OracleRenderer(scene, width, height, max_bounces=max_bounces)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(frames)
    har = HarnessRenderer(scene, width, height, max_bounces=max_bounces)
    har.render_frames_lanes(spp, 1, frames, shade_num, leaf_lanes)
    assert np.array_equal(har.ids, orc.ids)
    assert np.array_equal(har.frame_f16, orc.frame_f16)
    assert np.array_equal(har.accum, orc.accum)
    assert int(har.counters[0]) == int(orc.counters[0])


EDGE_CASES = ["empty", "spheres_only", "no_lights", "rotated_camera", "camera_down", "emissive_and_mirror_spheres",
              "duplicates_and_degenerates", "many_spheres", "shuffled_duplicates"]


@pytest.mark.parametrize("name", EDGE_CASES)
def test_edge_scenes_match_oracle(name):
    from tests.synth import edge_scenes

    scene = edge_scenes()[name]
    compare(scene, 72, 41, 2, 3, False)
    orc = O.OracleRenderer(scene, 40, 23)
    orc.set_samples_per_pixel(2)
    orc.render_frames(2)
    har = HarnessRenderer(scene, 40, 23)
    har.render_frames_lanes(2, 1, 2)
    assert np.array_equal(har.ids, orc.ids)
    assert np.array_equal(har.frame_f16, orc.frame_f16)
    assert np.array_equal(har.accum, orc.accum)


@pytest.mark.parametrize("width,height", [(1, 1), (1, 9), (9, 1), (8, 8), (65, 3)])
def test_tiny_and_thin_images(scenes, width, height):
    compare(scenes["boxes"], width, height, 2, 2, False)


def test_late_frame_indices(scenes):
    """frameIndex 1000..1001: the seed offset frameIndex * 12345 and the weight 1/(frameIndex+1)."""
    scene = scenes["spheres"]
    orc = O.OracleRenderer(scene, 48, 27)
    orc.set_samples_per_pixel(2)
    orc.frame_index = 999
    orc.render_frames(2)
    har = HarnessRenderer(scene, 48, 27)
    har.frame_index = 999
    har.render(2)
    har.render(2)
    assert np.array_equal(har.frame_f16, orc.frame_f16)
    assert np.array_equal(har.accum, orc.accum)


def test_card_deck_needs_more_than_sixteen_stack_entries():
    """The scene the GPU suite uses to reach the traversal stack's overflow area (entries beyond the
    16 kept in LDS for large scenes): the emulated lanes must actually get that deep, and still match."""
    from tests.synth import card_deck_scene

    scene = card_deck_scene(3000)
    orc = O.OracleRenderer(scene, 40, 23)
    orc.set_samples_per_pixel(2)
    orc.render_frames(1)
    har = HarnessRenderer(scene, 40, 23)
    har.render_frames_lanes(2, 1, 1)
    assert np.array_equal(har.ids, orc.ids)
    assert np.array_equal(har.accum, orc.accum)
    assert int(har.counters[7]) > 16 and len(scene.triangles) > 4096


@pytest.mark.parametrize("which,expect", [("suzanne", 16), ("boxes", 16), ("card_deck", 0), ("heightfield", 0)])
def test_links_packed_into_the_node_planes(scenes, which, expect):
    """The builder writes every 4-wide node's child links into the low bytes of its x and y planes
    when they fit 16 bits, and the kernels for such scenes read them from there instead of
    fetching the link vector.  The emulated lanes read the links that way here and must match the
    oracle (scenes of more than 4096 triangles keep the link vector); the emulation fails if a decoded
    link leads outside the node or triangle tables."""
    from tests.synth import card_deck_scene, heightfield_scene

    scene = {"card_deck": lambda: card_deck_scene(3000), "heightfield": lambda: heightfield_scene(48)}.get(
        which, lambda: scenes[which])()
    orc = O.OracleRenderer(scene, 40, 23)
    orc.set_samples_per_pixel(2)
    orc.render_frames(2)
    har = HarnessRenderer(scene, 40, 23)
    assert har.render_frames_lanes(2, 1, 2, packed_links=True) == expect
    assert np.array_equal(har.ids, orc.ids)
    assert np.array_equal(har.frame_f16, orc.frame_f16)
    assert np.array_equal(har.accum, orc.accum)
    assert int(har.counters[0]) == int(orc.counters[0])


@pytest.mark.parametrize("which", ["suzanne", "boxes", "card_deck", "heightfield", "rotated_camera", "shuffled_duplicates"])
def test_lanes_over_the_64_byte_nodes_match_oracle(scenes, which):
    """The large-scene kernels visit 64-byte nodes: child boxes as 8-bit multiples of a power-of-two step from
    the node's own box, rounded outwards.  Boxes only decide what is visited: the emulated lanes over those
    nodes must give the oracle's image bit for bit on any scene."""
    from tests.synth import card_deck_scene, edge_scenes, heightfield_scene

    scene = {"card_deck": lambda: card_deck_scene(3000), "heightfield": lambda: heightfield_scene(48),
             "rotated_camera": lambda: edge_scenes()["rotated_camera"],
             "shuffled_duplicates": lambda: edge_scenes()["shuffled_duplicates"]}.get(which, lambda: scenes[which])()
    orc = O.OracleRenderer(scene, 40, 23)
    orc.set_samples_per_pixel(2)
    orc.render_frames(2)
    har = HarnessRenderer(scene, 40, 23)
    har.render_frames_lanes(2, 1, 2, quant_nodes=True)
    assert np.array_equal(har.ids, orc.ids)
    assert np.array_equal(har.frame_f16, orc.frame_f16)
    assert np.array_equal(har.accum, orc.accum)
    assert int(har.counters[0]) == int(orc.counters[0])


def test_sphere_light_extension_in_the_lane_state_machine():
    """The same extension through the per-lane state machine of the persistent kernel: a bounce can have two
    shadow rays, the second one set aside while the first is traced; the additions to `light` stay in the
    shader's order."""
    from tests.synth import edge_scenes

    for name in ("emissive_and_mirror_spheres", "many_spheres", "no_lights"):
        scene = edge_scenes()[name]
        orc = O.OracleRenderer(scene, 64, 36, sphere_lights=True)
        orc.set_samples_per_pixel(3)
        orc.render_frames(2)
        har = HarnessRenderer(scene, 64, 36, sphere_lights=True)
        har.render_frames_lanes(3, 1, 2)
        assert np.array_equal(har.ids, orc.ids)
        assert np.array_equal(har.frame_f16, orc.frame_f16)
        assert np.array_equal(har.accum, orc.accum)
        assert int(har.counters[0]) == int(orc.counters[0])


@pytest.mark.parametrize("n,with_triangles", [(9, True), (60, True), (150, False)])
def test_spheres_as_leaves_of_the_tree(n, with_triangles):
    """More than eight spheres go into the acceleration structure as leaves of their own kind.  Both traversals
    of the shared core (the one-thread-per-pixel loop and the lane state machine) against the oracle, which
    tests every sphere per ray as the reference does: ids (original sphere indices, spheres before
    triangles on ties), frame, accumulator, ray count; with and without triangles in the scene."""
    from tests.synth import sphere_field_scene

    scene = sphere_field_scene(n, with_triangles)
    orc = O.OracleRenderer(scene, 64, 36)
    orc.set_samples_per_pixel(2)
    orc.render_frames(2)
    assert (orc.ids >= 0).any() and (orc.ids < n).any()
    har = HarnessRenderer(scene, 64, 36)
    for _ in range(2):
        har.render(2)
    lanes = HarnessRenderer(scene, 64, 36)
    lanes.render_frames_lanes(2, 1, 2, packed_links=True)
    for got in (har, lanes):
        assert np.array_equal(got.ids, orc.ids)
        assert np.array_equal(got.frame_f16, orc.frame_f16)
        assert np.array_equal(got.accum, orc.accum)
        assert int(got.counters[0]) == int(orc.counters[0])


def test_64_byte_nodes_enclose_the_128_byte_ones(scenes):
    from raytracer_amd import scene as S
    from tests.harness_lib import lib
    from tests.synth import heightfield_scene
    import ctypes as C

    L = lib()
    L.har_check_quant_nodes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.har_check_quant_nodes.restype = C.c_int
    for scene in (scenes["suzanne"], scenes["boxes"], heightfield_scene(64)):
        sp, tr = S.pack_spheres(scene.spheres), S.pack_triangles(scene.triangles)
        assert L.har_check_quant_nodes(sp.ctypes.data_as(C.c_void_p), len(sp), tr.ctypes.data_as(C.c_void_p), len(tr)) == 0


def test_link_packing_only_moves_planes_outwards(scenes):
    """Property check of the builder's link packing against an unpacked build of the same scene: planes
    move outwards by at most 512 ulp, the bytes spell the links, the z planes and the link vector are
    untouched; scenes with coordinates of both signs, zeros and tiny values included."""
    import ctypes as C

    from raytracer_amd import scene as S
    from tests.harness_lib import lib
    from tests.synth import edge_scenes

    L = lib()
    cases = dict(scenes)
    cases.update(edge_scenes())
    checked = 0
    for name, sc in cases.items():
        sp, tr = S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles)
        rc = L.har_check_link_packing(sp.ctypes.data_as(C.c_void_p), len(sp), tr.ctypes.data_as(C.c_void_p), len(tr))
        assert rc <= 0, "%s: %d violations" % (name, rc)
        checked += rc == 0
    assert checked >= 5


@pytest.mark.parametrize("brute", [True, False])
def test_sphere_light_extension_matches_oracle(brute):
    """The opt-in sampling of emissive spheres (the shader's own sample_light_spheres, which it never
    calls): shared core against the oracle, on scenes with one and with no emissive sphere."""
    from tests.synth import edge_scenes

    for name in ("emissive_and_mirror_spheres", "many_spheres", "no_lights"):
        scene = edge_scenes()[name]
        orc = O.OracleRenderer(scene, 64, 36, sphere_lights=True)
        orc.set_samples_per_pixel(3)
        har = HarnessRenderer(scene, 64, 36, brute=brute, sphere_lights=True)
        for _ in range(2):
            orc.render()
            har.render(3)
        assert np.array_equal(har.ids, orc.ids)
        assert np.array_equal(har.frame_f16, orc.frame_f16)
        assert np.array_equal(har.accum, orc.accum)
        assert int(har.counters[0]) == int(orc.counters[0])
    # and it does change the image where there is an emissive sphere
    scene = edge_scenes()["emissive_and_mirror_spheres"]
    off = O.OracleRenderer(scene, 64, 36)
    off.set_samples_per_pixel(3)
    off.render()
    on = O.OracleRenderer(scene, 64, 36, sphere_lights=True)
    on.set_samples_per_pixel(3)
    on.render()
    assert not np.array_equal(on.accum, off.accum) and int(on.counters[0]) > int(off.counters[0])


@pytest.mark.parametrize("seed", range(10))
def test_random_scenes_match_oracle(seed):
    """Seeded random scenes (tests.synth.random_scene): the traversal of the shared core and the lane state
    machine against the oracle's brute-force scan."""
    from tests.synth import random_scene

    scene = random_scene(seed)
    compare(scene, 40, 23, 1, 2, False)
    orc = O.OracleRenderer(scene, 40, 23)
    orc.set_samples_per_pixel(2)
    orc.render_frames(2)
    har = HarnessRenderer(scene, 40, 23)
    har.render_frames_lanes(2, 1, 2)
    assert np.array_equal(har.ids, orc.ids)
    assert np.array_equal(har.frame_f16, orc.frame_f16)
    assert np.array_equal(har.accum, orc.accum)
    assert int(har.counters[0]) == int(orc.counters[0])

