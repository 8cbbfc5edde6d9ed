"""One process per GPU: the framebuffer is cut into 8x8 tiles dealt round-robin to ranks (the reference's
tiled-coords chunks, core.clj:59-71, become device tiles), every rank renders its tiles from a replicated scene,
and ONE gather (RCCL over xGMI when the backend is nccl) brings the tile-major buffers to rank 0, which un-tiles and
quantises them on its GPU.  Pixels are independent and the stream key is the global pixel index, so the image does not
depend on the partition."""
import torch
import torch.distributed as dist

from . import _ffi
from .core import check

TILE = _ffi.TILE
HOST_STAGED_GATHER = False  # rehearsal only (gloo on a one-GPU box): stage the gather through host memory


def n_tiles(nx, ny):
    return ((nx + TILE - 1) // TILE) * ((ny + TILE - 1) // TILE)


def tiles_per_rank(nx, ny, world):
    """every rank's buffer is padded to this many tiles so the gather is uniform"""
    return (n_tiles(nx, ny) + world - 1) // world


def local_tile_ids(nx, ny, rank, world):
    """global tile indices rank `rank` renders: rank, rank + world, ..."""
    return list(range(rank, n_tiles(nx, ny), world))


def gather_tiles(local_tiles, world, rank, dst=0, group=None):
    """local_tiles: [tiles_per_rank, 64, 3] float64 on this rank's device -> on dst: [world, tiles_per_rank, 64, 3]."""
    if world == 1:
        return local_tiles.unsqueeze(0)
    if HOST_STAGED_GATHER:
        host = local_tiles.cpu()
        if rank == dst:
            out = torch.empty((world,) + tuple(host.shape), dtype=host.dtype)
            dist.gather(host, list(out.unbind(0)), dst=dst, group=group)
            return out.to(local_tiles.device)
        dist.gather(host, None, dst=dst, group=group)
        return None
    if rank == dst:
        out = torch.empty((world,) + tuple(local_tiles.shape), dtype=local_tiles.dtype, device=local_tiles.device)
        dist.gather(local_tiles, list(out.unbind(0)), dst=dst, group=group)
        return out
    dist.gather(local_tiles, None, dst=dst, group=group)
    return None


class TileRenderer:
    """Per-rank driver of the tile-partitioned render."""

    def __init__(self, device_scene, nx, ny, rank, world):
        self.ds, self.nx, self.ny, self.rank, self.world = device_scene, nx, ny, rank, world
        self.per = tiles_per_rank(nx, ny, world)
        dev = torch.device("cuda", device_scene.ctx.device)
        self.local = torch.zeros((self.per, 64, 3), dtype=torch.float64, device=dev)
        self.counters = torch.zeros(2, dtype=torch.int64, device=dev)
        if rank == 0:
            self.linear = torch.zeros((ny, nx, 3), dtype=torch.float64, device=dev)
            self.rgb8 = torch.zeros((ny, nx, 3), dtype=torch.uint8, device=dev)

    def step(self, ns, depth=50, seed=0x5EED0002, precision="f64"):
        """render local tiles -> gather -> (rank 0) assemble.  Asynchronous on the current stream."""
        stream = torch.cuda.current_stream().cuda_stream
        self.ds.render_tiles_device(self.nx, self.ny, ns, self.rank, self.world, self.local, self.counters, depth, seed, precision, stream)
        gathered = gather_tiles(self.local, self.world, self.rank)
        if self.rank == 0:
            check(_ffi.lib().rtmi_assemble_device(self.ds.ctx.handle, self.nx, self.ny, self.world, self.per, _ffi.ptr(gathered),
                                                  _ffi.ptr(self.linear), _ffi.ptr(self.rgb8), _ffi.ptr(stream)))
