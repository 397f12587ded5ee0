"""The JavaScript host (raytracer_amd/js): loaders and packers against the Python
mirror without a GPU; HipRenderer through the N-API addon against the oracle on
the GPU box.  Node 12 is the runtime the image provides."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from raytracer_amd import scene as S
from tests import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "raytracer_amd", "js")
NODE = shutil.which("node")

pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def node(script, *args):
    out = subprocess.run([NODE, "-e", script] + list(args), capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("name", ["spheres", "boxes", "suzanne"])
def test_js_loader_and_packers_match_python(name, tmp_path):
    script = """
const C = require('%s/common'), G = require('%s/gpu-schemas'), fs = require('fs');
C.loadScene(process.argv[1]).then((s) => {
  fs.writeFileSync(process.argv[2] + '.sp', Buffer.from(G.packSpheres(s.spheres).buffer));
  fs.writeFileSync(process.argv[2] + '.tr', Buffer.from(G.packTriangles(s.triangles).buffer));
  fs.writeFileSync(process.argv[2] + '.un', Buffer.from(G.packUniforms(s.camera, 7, 640, 360, 16, true).buffer));
});
""" % (JS, JS)
    node(script, S.scene_path(name), str(tmp_path / "o"))
    sc = S.load_scene(S.scene_path(name))
    assert np.array_equal(np.fromfile(tmp_path / "o.sp", np.float32).reshape(-1, 16), S.pack_spheres(sc.spheres))
    assert np.array_equal(np.fromfile(tmp_path / "o.tr", np.float32).reshape(-1, 24), S.pack_triangles(sc.triangles))
    assert np.array_equal(np.fromfile(tmp_path / "o.un", np.float32), S.pack_uniforms(sc.camera, 7, 640, 360, 16, True))


def test_js_model_transform_extension_matches_python(tmp_path):
    script = """
const C = require('%s/common'), G = require('%s/gpu-schemas'), fs = require('fs');
C.loadScene(process.argv[1], { applyModelTransforms: true }).then((s) => {
  fs.writeFileSync(process.argv[2], Buffer.from(G.packTriangles(s.triangles).buffer));
  console.log(JSON.stringify(C.transformPoint([1, 2, 3], { scale: [2, 1, 1], rotation: [10, 20, 30], position: [0.5, 0, -1] })));
});
""" % (JS, JS)
    out = node(script, S.scene_path("suzanne"), str(tmp_path / "t.bin"))
    sc = S.load_scene(S.scene_path("suzanne"), apply_model_transforms=True)
    got = np.fromfile(tmp_path / "t.bin", np.float32).reshape(-1, 24)
    assert np.allclose(got, S.pack_triangles(sc.triangles), atol=1e-6)
    py = S.transform_point([1, 2, 3], dict(scale=[2, 1, 1], rotation=[10, 20, 30], position=[0.5, 0, -1]))
    assert np.allclose(json.loads(out.strip().splitlines()[-1]), py, atol=1e-12)


def test_png_and_ppm_encoders(tmp_path):
    """image-writer.js: a 5x3 RGBA image through both encoders; the PNG is decoded here by hand
    (signature, IHDR, one zlib stream of filter-0 scanlines, chunk CRCs)."""
    import struct
    import zlib

    script = """
const W = require('%s/image-writer'), fs = require('fs');
const px = new Uint8Array(5 * 3 * 4);
for (let i = 0; i < px.length; i++) px[i] = (i * 37 + 11) & 255;
W.writeImage(process.argv[1] + '.png', px, 5, 3);
W.writeImage(process.argv[1] + '.ppm', px, 5, 3);
""" % JS
    node(script, str(tmp_path / "img"))
    rgba = ((np.arange(60) * 37 + 11) & 255).astype(np.uint8).reshape(3, 5, 4)
    ppm = (tmp_path / "img.ppm").read_bytes()
    assert ppm == b"P6\n5 3\n255\n" + rgba[:, :, :3].tobytes()
    png = (tmp_path / "img.png").read_bytes()
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(png):
        (length,) = struct.unpack(">I", png[pos:pos + 4])
        tag, data = png[pos + 4:pos + 8], png[pos + 8:pos + 8 + length]
        (crc,) = struct.unpack(">I", png[pos + 8 + length:pos + 12 + length])
        assert crc == zlib.crc32(tag + data) & 0xFFFFFFFF
        chunks.append((tag, data))
        pos += 12 + length
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (5, 3, 8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(3, 1 + 5 * 3)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(3, 5, 3), rgba[:, :, :3])


def test_js_obj_parser_quirks():
    script = """
const C = require('%s/common');
const t = C.parseOBJ(['v 0 0 0','v 1 0 0','v 1 1 0','v 0 1 0','vn 0 0 1','f 1//1 2//1 3//1 4//1','f 1/9/1 2/9/1 3/9/1','f 1 2'].join('\\n')).triangles;
const bad = C.parseOBJ('v 0 0 0\\nv 1 0 0\\nv 1 1 0\\nvn 0 0 1\\nf 1 2 3').triangles;
console.log(JSON.stringify({n: t.length, second: t[1], bad: bad.length, badv0: bad[0].v0 === undefined}));
""" % JS
    r = json.loads(node(script))
    assert r["n"] == 3 and r["second"]["v1"] == [1, 1, 0] and r["second"]["v2"] == [0, 1, 0]
    # without a normal index every vertex is skipped and the triangle is left without vertices, as in the reference
    assert r["bad"] == 1 and r["badv0"] is True


def test_addon_exports_the_abi():
    script = "const a = require('%s/rt_hip_napi.node'); console.log(JSON.stringify(Object.keys(a).sort()))" % JS
    names = json.loads(node(script))
    for n in ["setSphereLights", "create", "destroy", "setScene", "setUniforms", "renderFrame", "renderFrames", "readAccum",
              "readPrimaryIds", "readFrameF16", "takeTriangleTests", "readCounters", "deviceCount", "sync",
              "clearAccum", "renderFrameSums", "finishFrameSums", "writeAccum", "lastRenderFrames", "present"]:
        assert n in names


def test_create_resolves_null_on_failure():
    # a missing scene file: Create reports the error and resolves to null, it never rejects
    # (src/webgpu-renderer.ts:97-100); without a GPU it fails even earlier, same outcome
    script = """
const { HipRenderer } = require('%s/hip-renderer');
HipRenderer.Create({ width: 16, height: 9 }, '/nonexistent/scene.json').then((r) => console.log(r === null ? 'NULL' : 'OBJ'));
""" % JS
    assert node(script).strip().endswith("NULL")


@pytest.mark.gpu
@pytest.mark.parametrize("per_frame", [False, True])
def test_js_renderer_matches_oracle(scenes, tmp_path, per_frame):
    width, height, frames, spp = 96, 54, 3, 4
    args = [NODE, os.path.join(JS, "cli.js"), "--scene", S.scene_path("suzanne"), "--width", str(width), "--height",
            str(height), "--spp", str(spp), "--frames", str(frames), "--dump", str(tmp_path / "d"), "--ppm",
            str(tmp_path / "img.ppm"), "--preview-every", "2"] + (["--per-frame"] if per_frame else [])
    out = subprocess.run(args, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    info = json.loads(out.stdout.strip().splitlines()[-1])
    orc = O.OracleRenderer(scenes["suzanne"], width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(frames)
    assert np.array_equal(np.fromfile(tmp_path / "d.ids.i32", np.int32).reshape(height, width), orc.ids)
    assert np.array_equal(np.fromfile(tmp_path / "d.frame.f16", np.uint16).reshape(height, width, 4), orc.frame_f16)
    acc = np.fromfile(tmp_path / "d.accum.f32", np.float32).reshape(height, width, 3)
    assert np.array_equal(acc, orc.accum_rgb())
    assert info["rays"] == int(orc.counters[0]) and info["frame_index"] == frames
    assert info["triangle_tests_u32"] == (int(orc.counters[0]) * 982) % 2**32
    # presented image = accumulator as 8-bit unorm, no gamma (accumulator.wgsl:67)
    ppm = (tmp_path / "img.ppm").read_bytes()
    header = ("P6\n%d %d\n255\n" % (width, height)).encode()
    assert ppm.startswith(header)
    img = np.frombuffer(ppm[len(header):], np.uint8).reshape(height, width, 3)
    assert np.array_equal(img, np.floor(orc.accum_rgb().astype(np.float64) * 255 + 0.5).astype(np.uint8))


@pytest.mark.gpu
def test_js_render_loop_runs_unfenced_and_in_order(scenes, tmp_path):
    """startRenderLoop submits one frame per tick with no fence, as the reference's loop does; the native
    call keeps two frames in flight.  Stopped after at least 7 frames, the accumulator must equal the
    oracle's after exactly that many."""
    width, height, spp = 96, 54, 2
    script = """
const fs = require('fs');
const { HipRenderer } = require('%(js)s/hip-renderer');
const canvas = { width: %(w)d, height: %(h)d, pixels: new Uint8Array(%(w)d * %(h)d * 4) };
HipRenderer.Create(canvas, '%(scene)s', { presentEvery: 1 }).then((r) => {
  r.setSamplesPerPixel(%(spp)d);
  r.startRenderLoop();
  const poll = () => {
    if (r.frameIndex < 7) return setImmediate(poll);
    r.stopRenderLoop();
    r.sync();
    fs.writeFileSync('%(out)s', Buffer.from(r.readAccum().buffer));
    const shown = r.presentNow();
    fs.writeFileSync('%(out)s.rgba', Buffer.from(canvas.pixels.buffer));
    console.log(JSON.stringify({ frames: r.frameIndex, fps: r.getFPS(), shown: shown }));
    r.destroy();
  };
  setImmediate(poll);
});
""" % dict(js=JS, w=width, h=height, spp=spp, scene=S.scene_path("boxes"), out=str(tmp_path / "loop.f32"))
    info = json.loads(node(script).strip().splitlines()[-1])
    assert info["frames"] >= 7
    orc = O.OracleRenderer(scenes["boxes"], width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(info["frames"])
    acc = np.fromfile(tmp_path / "loop.f32", np.float32).reshape(height, width, 3)
    assert np.array_equal(acc, orc.accum_rgb())
    # the canvas image (presented every frame without waiting during the loop, then once waited for)
    shown = np.fromfile(str(tmp_path / "loop.f32") + ".rgba", np.uint8).reshape(height, width, 4)
    assert info["shown"] == info["frames"]
    assert np.array_equal(shown[:, :, :3], np.floor(orc.accum_rgb().astype(np.float64) * 255 + 0.5).astype(np.uint8))
    assert (shown[:, :, 3] == 255).all()


@pytest.mark.gpu
def test_js_row_split_assembles_the_unstriped_image(scenes, tmp_path):
    """raytracer_amd/js/multi-gpu.js with three striped renderers on the one GPU of the box: rows read
    back with readOwnedRows, assembled with importRows on the first renderer; bit-equal to the oracle."""
    width, height, spp, frames = 100, 61, 2, 3
    script = """
const fs = require('fs');
const { renderSplit } = require('%(js)s/multi-gpu');
renderSplit('%(scene)s', { width: %(w)d, height: %(h)d, spp: %(spp)d, frames: %(f)d, devices: [0, 0, 0] }).then((r) => {
  fs.writeFileSync('%(out)s', Buffer.from(r.accum.buffer));
  console.log(JSON.stringify({ rays: r.rays }));
}).catch((e) => { console.error(e); process.exit(1); });
""" % dict(js=JS, w=width, h=height, spp=spp, f=frames, scene=S.scene_path("suzanne"), out=str(tmp_path / "split.f32"))
    info = json.loads(node(script).strip().splitlines()[-1])
    orc = O.OracleRenderer(scenes["suzanne"], width, height)
    orc.set_samples_per_pixel(spp)
    orc.render_frames(frames)
    acc = np.fromfile(tmp_path / "split.f32", np.float32).reshape(height, width, 3)
    assert np.array_equal(acc, orc.accum_rgb())
    assert info["rays"] == int(orc.counters[0])


@pytest.mark.gpu
def test_js_frame_split_matches_the_recurrence_up_to_rounding(scenes, tmp_path):
    """raytracer_amd/js/multi-gpu.js renderFrameSplit with two renderers on the one GPU of the box: frames
    dealt round-robin, sums added on the host, closed with finishFrameSums.  Against the oracle's
    sequential accumulator: rounding-level agreement on pixels the recurrence never clamps, the exact
    closed form everywhere, and the same ray count."""
    width, height, spp, frames = 96, 54, 2, 5
    script = """
const fs = require('fs');
const { renderFrameSplit } = require('%(js)s/multi-gpu');
renderFrameSplit('%(scene)s', { width: %(w)d, height: %(h)d, spp: %(spp)d, frames: %(f)d, devices: [0, 0] }).then((r) => {
  fs.writeFileSync('%(out)s', Buffer.from(r.accum.buffer));
  console.log(JSON.stringify({ rays: r.rays }));
}).catch((e) => { console.error(e); process.exit(1); });
""" % dict(js=JS, w=width, h=height, spp=spp, f=frames, scene=S.scene_path("boxes"), out=str(tmp_path / "fs.f32"))
    info = json.loads(node(script).strip().splitlines()[-1])
    orc = O.OracleRenderer(scenes["boxes"], width, height)
    orc.set_samples_per_pixel(spp)
    sums = [np.zeros((height, width, 3), np.float32) for _ in range(2)]
    clamped = np.zeros((height, width), bool)
    for f in range(1, frames + 1):
        before = orc.accum_rgb()
        orc.render()
        cur = orc.frame_f16.view(np.float16).astype(np.float32)[:, :, :3]
        sums[(f - 1) % 2] += cur       # each renderer adds its frames in increasing order, in f32
        w = np.float32(1) / np.float32(f + 1)
        mean = before * (np.float32(1) - w) + cur * w
        clamped |= ~((mean >= 0) & (mean <= 1)).all(axis=2)
    total = sums[0] + sums[1]
    with np.errstate(invalid="ignore"):
        closed = np.minimum(np.maximum(total / np.float32(frames + 1), np.float32(0)), np.float32(1))
    got = np.fromfile(tmp_path / "fs.f32", np.float32).reshape(height, width, 3)
    assert np.array_equal(got, closed, equal_nan=True)
    assert np.abs(got - orc.accum_rgb())[~clamped].max() <= 2e-6
    assert info["rays"] == int(orc.counters[0])


@pytest.mark.gpu
def test_js_debug_view(scenes, tmp_path):
    """Create(..., { debugView: true }) + setDebug(true): the triangles' normal markers are drawn (the
    extension of rt_set_debug_view), bit-equal to the oracle with the same option; setDebug(true) without
    the option changes nothing, as in the reference."""
    width, height, spp = 64, 36, 2
    script = """
const fs = require('fs');
const { HipRenderer } = require('%(js)s/hip-renderer');
(async () => {
  for (const debugView of [true, false]) {
    const r = await HipRenderer.Create({ width: %(w)d, height: %(h)d }, '%(scene)s', { debugView });
    r.setSamplesPerPixel(%(spp)d);
    r.setDebug(true);
    r.render(); r.render();
    fs.writeFileSync('%(out)s.' + debugView, Buffer.from(r.readAccum().buffer));
    fs.writeFileSync('%(out)s.ids.' + debugView, Buffer.from(r.readIds().buffer));
    r.destroy();
  }
  console.log('ok');
})();
""" % dict(js=JS, w=width, h=height, spp=spp, scene=S.scene_path("boxes"), out=str(tmp_path / "dbg"))
    assert node(script).strip().endswith("ok")
    for flag in (True, False):
        orc = O.OracleRenderer(scenes["boxes"], width, height, debug_view=flag)
        orc.debug_enabled = True
        orc.set_samples_per_pixel(spp)
        orc.render_frames(2)
        acc = np.fromfile(str(tmp_path / "dbg") + "." + ("true" if flag else "false"), np.float32).reshape(height, width, 3)
        ids = np.fromfile(str(tmp_path / "dbg") + ".ids." + ("true" if flag else "false"), np.int32).reshape(height, width)
        assert np.array_equal(ids, orc.ids)
        assert np.array_equal(acc, orc.accum_rgb())
        assert bool((ids >= 1 + 34).any()) == flag
