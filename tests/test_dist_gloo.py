"""The N > 1 path on CPU: world_size-2 gloo.  Each rank fills its round-robin 8x8 tiles (the oracle stands in for the
GPU renderer here -- this is a test), raytrace_clj_amd.dist gathers them to rank 0, and the un-tiled frame must equal
the single-process frame: the partition, the padding to tiles_per_rank and the gather order are what is under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import raytrace_clj_amd as r
from raytrace_clj_amd import dist as rdist

NX, NY, NS, SEED = 44, 27, 2, 0x5EED0002  # partial tiles on both edges; 6 x 4 = 24 tiles


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _untile(gathered, nx, ny, world):
    """numpy restatement of rtmi_assemble_device's indexing (include/rtmi.h)"""
    tiles_x = (nx + 7) // 8
    out = np.zeros((ny, nx, 3))
    for y in range(ny):
        for x in range(nx):
            g = (y // 8) * tiles_x + x // 8
            out[y, x] = gathered[g % world, g // world, (y % 8) * 8 + x % 8]
    return out


def _worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import Oracle
        orc = Oracle("f64")
        flat = r.flatten.flatten(r.scene.make_random_scene(NX, NY, 3, True))
        per = rdist.tiles_per_rank(NX, NY, world)
        local = torch.zeros((per, 64, 3), dtype=torch.float64)
        tiles_x = (NX + 7) // 8
        ids = rdist.local_tile_ids(NX, NY, rank, world)
        assert len(ids) <= per and all(g % world == rank for g in ids)
        rays = 0
        for k, g in enumerate(ids):
            x0, y0 = (g % tiles_x) * 8, (g // tiles_x) * 8
            x1, y1 = min(x0 + 8, NX), min(y0 + 8, NY)
            lin, _, cnt = orc.render(flat, NX, NY, NS, 50, SEED, region=(x0, y0, x1, y1))
            tile = np.zeros((8, 8, 3))
            tile[: y1 - y0, : x1 - x0] = lin
            local[k] = torch.from_numpy(tile.reshape(64, 3))
            rays += int(cnt[0])
        gathered = rdist.gather_tiles(local, world, rank)
        total = torch.tensor([rays], dtype=torch.int64)
        dist.all_reduce(total)
        if rank == 0:
            assert gathered.shape == (world, per, 64, 3)
            np.save(os.path.join(tmpdir, "frame.npy"), _untile(gathered.numpy(), NX, NY, world))
            np.save(os.path.join(tmpdir, "rays.npy"), total.numpy())
        else:
            assert gathered is None
    finally:
        dist.destroy_process_group()


def test_tile_partition_gather_world2(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    frame = np.load(tmp_path / "frame.npy")
    flat = r.flatten.flatten(r.scene.make_random_scene(NX, NY, 3, True))
    full, _, cnt = oracle.render(flat, NX, NY, NS, 50, SEED)
    assert np.array_equal(frame, full), "tile-partitioned + gathered frame must equal the single-process frame"
    assert int(np.load(tmp_path / "rays.npy")[0]) == int(cnt[0])


def test_partition_arithmetic():
    for nx, ny in [(800, 400), (3840, 2160), (200, 100), (44, 27), (8, 8)]:
        n = rdist.n_tiles(nx, ny)
        assert n == r._ffi.lib().rtmi_local_tiles(nx, ny, 0, 1)
        for world in (1, 2, 3, 8):
            per = rdist.tiles_per_rank(nx, ny, world)
            allids = []
            for rank in range(world):
                ids = rdist.local_tile_ids(nx, ny, rank, world)
                assert len(ids) == r._ffi.lib().rtmi_local_tiles(nx, ny, rank, world) <= per
                allids += ids
            assert sorted(allids) == list(range(n))
    assert rdist.gather_tiles(torch.zeros(3, 64, 3), 1, 0).shape == (1, 3, 64, 3)
