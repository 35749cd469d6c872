"""Multi-GPU frame tiling: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).

The reference parallelises a frame over threads pulling 32x32 chunks from an atomic counter
(src/render.cpp:468-493).  Across GPUs the unit is a band of RENDER_CHUNK_SIZE = 32 rows: band b is
rendered by rank b % world (round-robin, so the object in the middle of the image is shared out
evenly).  Pixels are independent, so there is no collective on the data path; the ONLY exchange is
the optional gather of the finished bands to one rank (`gather_framebuffer`).
"""
import numpy as np

BAND_ROWS = 32      # RENDER_CHUNK_SIZE, render.cpp:43


def owned_rows(height, rank, world, band_rows=BAND_ROWS):
    """Image rows rendered by `rank`, in the order they are stored in its compact buffer."""
    rows = []
    nbands = (height + band_rows - 1) // band_rows
    for band in range(rank, nbands, max(world, 1)):
        rows.extend(range(band * band_rows, min((band + 1) * band_rows, height)))
    return np.asarray(rows, dtype=np.int64)


def compact_len(fmt, rank, world, band_rows=BAND_ROWS):
    return len(owned_rows(fmt.height, rank, world, band_rows)) * fmt.pitch


def render_bands(renderer, scene, fmt, dest, rank, world, collect_stats=False):
    """Render this rank's bands of one frame into `dest` (a compact buffer: owned rows only).
    `dest` may be a host buffer or a torch device tensor (then the launch is only enqueued)."""
    return renderer.render(dest, fmt, scene, band_rank=rank, band_world=world, compact=True, collect_stats=collect_stats)


def gather_framebuffer(compact, fmt, rank, world, dst=0, band_rows=BAND_ROWS, group=None):
    """Gather every rank's compact band buffer to `dst` and de-interleave into a full
    [height, pitch] uint8 image (returned on `dst`, None elsewhere).

    One collective per frame: ranks own different numbers of rows when the band count is not a
    multiple of `world`, so buffers are padded to the largest and moved with a single
    all_gather_into_tensor-free `gather` (RCCL: each peer -> root over its own xGMI link)."""
    import torch
    import torch.distributed as dist

    rows = [owned_rows(fmt.height, r, world, band_rows) for r in range(world)]
    max_rows = max(len(r) for r in rows)
    flat = compact.reshape(-1)
    send = flat
    if flat.numel() != max_rows * fmt.pitch:
        send = torch.zeros(max_rows * fmt.pitch, dtype=torch.uint8, device=flat.device)
        send[:flat.numel()] = flat
    if world == 1:
        parts = [send]
    else:
        parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, parts, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.empty((fmt.height, fmt.pitch), dtype=torch.uint8, device=flat.device)
    for r in range(world):
        idx = torch.as_tensor(rows[r], device=flat.device)
        if len(idx):
            full[idx] = parts[r][:len(idx) * fmt.pitch].reshape(len(idx), fmt.pitch)
    return full
