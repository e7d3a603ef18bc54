"""The N > 1 path on CPU: tile partition + ONE gather (gloo, world_size 2 and 3)."""
import socket

import numpy as np
import pytest

from jaderaytracerendering_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_pack_and_owned_ids_cover_every_pixel_once():
    rng = np.random.default_rng(0)
    for (w, h, world) in [(70, 50, 3), (1920, 1080, 8), (16, 16, 2), (33, 1, 5)]:
        img = rng.random((h, w, 3)).astype(np.float32)
        ids = np.concatenate([D.owned_tile_ids(w, h, r, world) for r in range(world)])
        tx, ty = D.tile_grid(w, h)
        assert np.array_equal(np.sort(ids), np.arange(tx * ty))
        from jaderaytracerendering_amd.backend import assemble_tiles
        out = np.zeros_like(img)
        for r in range(world):
            assemble_tiles(D.pack_tiles(img, r, world), w, h, r, world, out)
        assert np.array_equal(out, img)
    assert D.max_owned(1920, 1080, 8) == 1020


def test_single_process_gather_is_identity():
    import torch
    img = np.random.default_rng(1).random((27, 45, 3)).astype(np.float32)
    frame = D.gather_framebuffer(torch.from_numpy(D.pack_tiles(img, 0, 1)), 45, 27)
    assert np.array_equal(frame.numpy(), img)


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_partition_and_gather_is_bit_exact(tmp_path, world):
    import torch.multiprocessing as mp
    from _dist_worker import run
    out = tmp_path / "result.txt"
    mp.spawn(run, args=(world, _free_port(), 70, 50, 2, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"
