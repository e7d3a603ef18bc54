"""Image-tile data parallelism: one process per GPU, one gather of the framebuffer.

The reference is single-GPU (one `render_pixel` launch, PathTrace.cu:1731).
Pixels share nothing but the read-only scene and every pixel owns its RNG
stream, so the image splits into the reference's 16x16 tiles
(TILE_SIZE, PathTrace.cu:32): rank r renders the tiles (tx, ty) with
(tx + ty) % world == r (diagonal interleave, because cost concentrates on the
statue and a row-major deal degenerates into vertical stripes).  There is no data-path collective while rendering; the only
exchange is ONE gather of each rank's compact tile buffer to rank 0
(`torch.distributed.gather`: RCCL over xGMI with backend "nccl", gloo on CPU).

Load order: PyTorch bundles its own HIP runtime.  A process that uses both must import torch and
initialise CUDA (torch.cuda.init() / set_device) BEFORE the first call into libjade_hip.so; in the
other order torch finds no GPU.  bench.py does it in that order.
"""
import numpy as np

from . import _abi

TILE = _abi.TILE_SIZE
TILE_FLOATS = TILE * TILE * 3


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def owned_tile_ids(width, height, rank, nranks):
    """Row-major ids of the tiles (tx, ty) with (tx + ty) % nranks == rank, increasing (jade_rt.h)."""
    tx, ty = tile_grid(width, height)
    ids = np.arange(tx * ty, dtype=np.int64)
    return ids[(ids % tx + ids // tx) % nranks == rank]


def max_owned(width, height, nranks):
    return max(len(owned_tile_ids(width, height, r, nranks)) for r in range(nranks))


def pack_tiles(image, rank, nranks):
    """[H, W, 3] full frame -> this rank's compact [n_owned, 16, 16, 3] buffer (zeros outside the image)."""
    h, w = image.shape[:2]
    ids = owned_tile_ids(w, h, rank, nranks)
    tx, _ = tile_grid(w, h)
    out = np.zeros((len(ids), TILE, TILE, 3), image.dtype)
    for k, tid in enumerate(ids):
        y0, x0 = (tid // tx) * TILE, (tid % tx) * TILE
        hh, ww = min(TILE, h - y0), min(TILE, w - x0)
        out[k, :hh, :ww] = image[y0:y0 + hh, x0:x0 + ww]
    return out


def gather_framebuffer(tiles, width, height, dst=0, group=None):
    """Gather every rank's tile buffer on `dst` and assemble the full frame there.

    tiles: torch tensor [n_owned, 16, 16, 3] (CPU for gloo, CUDA for nccl/RCCL).
    Returns the [height, width, 3] tensor on `dst`, None elsewhere.  Ranks own
    different tile counts when the tile total is not a multiple of the world
    size, so buffers are padded to the maximum before the single gather.
    """
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0  # single process, single GPU: nothing to exchange
    n_max = max_owned(width, height, world)
    buf = torch.zeros((n_max, TILE, TILE, 3), dtype=tiles.dtype, device=tiles.device)
    buf[: tiles.shape[0]] = tiles
    if world == 1:
        parts = [buf]
    else:
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, parts, dst=dst, group=group)
    if rank != dst:
        return None
    tx, ty = tile_grid(width, height)
    # put every rank's k-th tile at its row-major id, then un-tile
    allt = torch.stack(parts, dim=0).reshape(world * n_max, TILE, TILE, 3)
    src = np.zeros(tx * ty, np.int64)
    for r in range(world):
        ids = owned_tile_ids(width, height, r, world)
        src[ids] = r * n_max + np.arange(len(ids))
    stacked = allt[torch.from_numpy(src).to(allt.device)]
    frame = stacked.reshape(ty, tx, TILE, TILE, 3).permute(0, 2, 1, 3, 4).reshape(ty * TILE, tx * TILE, 3)
    return frame[:height, :width].contiguous()
