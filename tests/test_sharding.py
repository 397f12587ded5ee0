"""Multi-GPU row sharding: ownership rule and the gather/assemble step, with
two gloo ranks on the CPU.  Each rank renders its stripes with the CPU oracle
(the HIP path cannot run here); the assembled image must be bit-identical to a
single-process render, which is the property the N-GPU bench relies on."""
import os
import socket
import subprocess
import sys

import numpy as np

from raytracer_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import scene as S, sharding
from tests import oracle_lib as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, frames, spp, R = 48, 27, 2, 2, 8
sc = S.load_scene(S.scene_path("boxes"))
rows = sharding.owned_rows(H, R, rank, world)
accum = np.zeros((3, H, W), np.float32)
# render the owned stripes only (pixels are independent, so stripe by stripe is the same image)
stripes = sorted(set(int(y) // R for y in rows))
for f in range(1, frames + 1):
    u = S.pack_uniforms(sc.camera, f, W, H, spp)
    sp, tr = S.pack_spheres(sc.spheres), S.pack_triangles(sc.triangles)
    for s in stripes:
        y0, y1 = s * R, min(H, (s + 1) * R)
        rc = O.lib().orc_render_frame(u.ctypes.data, sp.ctypes.data, len(sp), tr.ctypes.data, len(tr), 6,
                                      0, y0, W, y1, accum.ctypes.data, None, None, None, 1)
        assert rc == 0
local = torch.from_numpy(np.ascontiguousarray(accum[:, rows, :]))
blocks = sharding.all_gather_rows(local, world)
if rank == 0:
    full = sharding.assemble([b.numpy() for b in blocks], H, W, R)
    ref = O.OracleRenderer(sc, W, H); ref.set_samples_per_pixel(spp); ref.render_frames(frames)
    assert np.array_equal(full, ref.accum), "assembled image differs from the single-process render"
    print("SHARD_OK")
dist.destroy_process_group()
'''


def test_ownership_partitions_rows():
    for height, stripe, world in [(27, 8, 2), (1080, 8, 8), (2160, 8, 4), (5, 8, 3), (64, 8, 8)]:
        seen = np.concatenate([sharding.owned_rows(height, stripe, r, world) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(height))
    assert sharding.owned_rows(27, 8, 1, 2).tolist() == list(range(8, 16)) + list(range(24, 27))


def test_assemble_inverts_ownership():
    h, w, r, world = 27, 5, 8, 3
    img = np.arange(3 * h * w, dtype=np.float32).reshape(3, h, w)
    blocks = [img[:, sharding.owned_rows(h, r, k, world), :] for k in range(world)]
    assert np.array_equal(sharding.assemble(blocks, h, w, r), img)


def test_two_rank_gloo_gather_is_bit_identical(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "SHARD_OK" in out.stdout


FRAME_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import scene as S, sharding
from tests import oracle_lib as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, frames, spp = 48, 27, 5, 2
sc = S.load_scene(S.scene_path("boxes"))
first, count, stride = sharding.frames_of_rank(1, frames, rank, world)
sums = np.zeros((3, H, W), np.float32)
orc = O.OracleRenderer(sc, W, H); orc.set_samples_per_pixel(spp)
for k in range(count):
    orc.frame_index = first + k * stride - 1      # render() pre-increments
    orc.render()
    f = orc.frame_f16.view(np.float16).astype(np.float32)
    sums += np.transpose(f[:, :, :3], (2, 0, 1))
t = torch.from_numpy(sums)
sharding.all_reduce_sums(t)
if rank == 0:
    got = sharding.finish_sums(t.numpy(), frames)
    ref = O.OracleRenderer(sc, W, H); ref.set_samples_per_pixel(spp)
    clamped = np.zeros((H, W), bool)    # pixels whose running mean left [0, 1] at some frame (or is NaN)
    for f in range(1, frames + 1):
        before = ref.accum.copy()
        ref.render()
        cur = np.transpose(ref.frame_f16.view(np.float16).astype(np.float32)[:, :, :3], (2, 0, 1))
        w = np.float32(1) / np.float32(f + 1)
        raw = before * (np.float32(1) - w) + cur * w
        clamped |= ~((raw >= 0) & (raw <= 1)).all(axis=0)
    diff = np.abs(got - ref.accum)
    assert diff[:, ~clamped].max() <= 2e-6, diff[:, ~clamped].max()
    print("FRAME_SPLIT_OK", float(diff[:, ~clamped].max()), int(clamped.sum()))
dist.destroy_process_group()
'''


def test_frames_of_rank_deals_every_frame_once():
    for first, n, world in [(1, 5, 2), (1, 64, 8), (7, 3, 4), (1, 1, 8), (1, 0, 2)]:
        seen = []
        for r in range(world):
            f0, count, stride = sharding.frames_of_rank(first, n, r, world)
            seen += [f0 + k * stride for k in range(count)]
        assert sorted(seen) == list(range(first, first + n))


def test_two_rank_gloo_frame_split_matches_recurrence(tmp_path):
    script = tmp_path / "frame_worker.py"
    script.write_text(FRAME_WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "FRAME_SPLIT_OK" in out.stdout
