"""Sharding across GPUs: by image rows (exact) or by frames (approximate), and how
the accumulator is combined afterwards.

The reference renders on one GPU.  A pixel's value depends only on
(x, y, width, frameIndex) and the replicated scene
(public/raytracer_compute.wgsl:299), so rows can be dealt to ranks with no
exchange while rendering; the only communication is one all-gather of the
finished accumulator rows.  Rows are dealt in 8-row stripes (the height of the
8x8 pixel tiles the kernel works in), round-robin, which balances cheap sky
rows and expensive rows: rank r owns row y iff (y // stripe_rows) % world == r.
This is the rule rt_create implements natively (include/rt_hip.h).

The frame split deals frames instead: rank r renders frames first + r,
first + r + world, ... of the whole image and keeps the plain f32 sum of their
rgba16float colours; one all-reduce adds the sums and
saturate(sum / (n_frames + 1)) stands in for the accumulator recurrence
(public/accumulator.wgsl:53-60).  It suits scenes so heavy that a frame should
stay on one GPU; it is not bit-identical to a single-GPU run (addition order,
no per-frame clamp of pixels whose running mean exceeds 1).
"""
from __future__ import annotations

from typing import List

import numpy as np


def owned_rows(height: int, stripe_rows: int, index: int, count: int) -> np.ndarray:
    """Global row numbers owned by stripe `index` of `count`, increasing."""
    y = np.arange(height)
    return y[(y // stripe_rows) % count == index]


def all_gather_rows_padded(local_block, world: int, counts=None):
    """All-gather of per-rank row blocks of different heights, in one collective.

    local_block: torch tensor (3, rows_r, width) on the device the process group
    communicates on (CUDA for RCCL, CPU for gloo).  Returns (gathered, counts):
    gathered is one tensor (world, 3, tallest, width), block r padded with zero rows
    to the tallest block (the layout rt_import_gathered_rows takes), counts the rows
    per rank.  counts: rows per rank when the caller knows them (owned_rows gives
    them without communication); otherwise one extra collective exchanges them."""
    import torch
    import torch.distributed as dist

    if counts is None:
        rows = torch.tensor([local_block.shape[1]], dtype=torch.int64, device=local_block.device)
        all_rows = [torch.zeros_like(rows) for _ in range(world)]
        dist.all_gather(all_rows, rows)
        counts = [int(r.item()) for r in all_rows]
    tallest = max(counts)
    if local_block.shape[1] == tallest:
        padded = local_block.contiguous()
    else:
        padded = torch.zeros((3, tallest, local_block.shape[2]), dtype=local_block.dtype, device=local_block.device)
        padded[:, : local_block.shape[1]] = local_block
    out = torch.empty((world,) + tuple(padded.shape), dtype=padded.dtype, device=padded.device)
    dist.all_gather(list(out.unbind(0)), padded)   # views of one buffer: the collective writes in place
    return out, counts


def all_gather_rows(local_block, world: int, counts=None):
    """all_gather_rows_padded, returned as the list of the ranks' dense blocks."""
    out, counts = all_gather_rows_padded(local_block, world, counts)
    return [out[r, :, :n].contiguous() for r, n in enumerate(counts)]


def assemble(blocks: List[np.ndarray], height: int, width: int, stripe_rows: int) -> np.ndarray:
    """Full (3, height, width) image from the per-rank blocks (rank order)."""
    count = len(blocks)
    full = np.zeros((3, height, width), dtype=np.float32)
    for index, block in enumerate(blocks):
        rows = owned_rows(height, stripe_rows, index, count)
        assert block.shape == (3, len(rows), width), (block.shape, len(rows))
        full[:, rows, :] = block
    return full


def frames_of_rank(first: int, n_frames: int, rank: int, world: int):
    """(first frame, frame count, stride) of `rank` when frames first .. first + n_frames - 1
    are dealt round-robin: frame k goes to rank (k - first) % world."""
    count = (n_frames - rank + world - 1) // world if n_frames > rank else 0
    return first + rank, count, world


def all_reduce_sums(planes):
    """In-place sum over ranks of the (3, height, width) float32 planes (torch tensor on the
    process group's device)."""
    import torch.distributed as dist

    dist.all_reduce(planes, op=dist.ReduceOp.SUM)
    return planes


def finish_sums(sums: np.ndarray, n_frames: int) -> np.ndarray:
    """Accumulator value of frames 1..n_frames from the sum of their colours (host-side
    counterpart of rt_finish_frame_sums)."""
    mean = sums.astype(np.float32) / np.float32(n_frames + 1)
    return np.fmin(np.fmax(mean, np.float32(0)), np.float32(1)).astype(np.float32)  # NaN -> 0, as the kernel's saturate
