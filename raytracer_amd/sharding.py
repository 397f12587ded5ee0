"""Image-row sharding across GPUs: which rows a rank renders and how the
accumulator rows are combined afterwards.

The reference renders on one GPU.  A pixel's value depends only on
(x, y, width, frameIndex) and the replicated scene
(public/raytracer_compute.wgsl:299), so rows can be dealt to ranks with no
exchange while rendering; the only communication is one all-gather of the
finished accumulator rows.  Rows are dealt in 8-row stripes (the height of the
8x8 pixel tiles the kernel works in), round-robin, which balances cheap sky
rows and expensive rows: rank r owns row y iff (y // stripe_rows) % world == r.
This is the rule rt_create implements natively (include/rt_hip.h).
"""
from __future__ import annotations

from typing import List

import numpy as np


def owned_rows(height: int, stripe_rows: int, index: int, count: int) -> np.ndarray:
    """Global row numbers owned by stripe `index` of `count`, increasing."""
    y = np.arange(height)
    return y[(y // stripe_rows) % count == index]


def all_gather_rows(local_block, world: int):
    """All-gather of per-rank row blocks of different heights.

    local_block: torch tensor (3, rows_r, width) on the device the process group
    communicates on (CUDA for RCCL, CPU for gloo).  Returns the list of blocks of
    all ranks.  One collective for the sizes, one for the payload (padded to the
    tallest block)."""
    import torch
    import torch.distributed as dist

    rows = torch.tensor([local_block.shape[1]], dtype=torch.int64, device=local_block.device)
    all_rows = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(all_rows, rows)
    counts = [int(r.item()) for r in all_rows]
    tallest = max(counts)
    padded = torch.zeros((3, tallest, local_block.shape[2]), dtype=local_block.dtype, device=local_block.device)
    padded[:, : local_block.shape[1]] = local_block
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return [o[:, :n].contiguous() for o, n in zip(out, counts)]


def assemble(blocks: List[np.ndarray], height: int, width: int, stripe_rows: int) -> np.ndarray:
    """Full (3, height, width) image from the per-rank blocks (rank order)."""
    count = len(blocks)
    full = np.zeros((3, height, width), dtype=np.float32)
    for index, block in enumerate(blocks):
        rows = owned_rows(height, stripe_rows, index, count)
        assert block.shape == (3, len(rows), width), (block.shape, len(rows))
        full[:, rows, :] = block
    return full
