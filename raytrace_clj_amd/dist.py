"""One process per GPU: the framebuffer is cut into 8x8 tiles dealt round-robin to ranks (the reference's
tiled-coords chunks, core.clj:59-71, become device tiles), every rank renders its tiles from a replicated scene,
and ONE gather (RCCL over xGMI when the backend is nccl) brings the tile-major buffers to rank 0, which un-tiles and
quantises them on its GPU.  Pixels are independent and the stream key is the global pixel index, so the image does not
depend on the partition."""
import torch
import torch.distributed as dist

from . import _ffi
from .core import Context, DeviceScene, check

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


class FramePipeline:
    """`depth` frames in flight on one GPU: every slot owns a context (its own sample workspace and work queue), the scene
    and a stream, and consecutive frames go to consecutive slots.  A path tracer's launch ends with a die-off -- the last
    deep paths (up to `depth 50` bounces) finish in nearly empty waves; with two slots the next frame's workgroups take
    the CU slots the previous frame's workgroups vacate, and its gather / assemble overlap the next render as well.
    Frames are independent, so the images are those of the one-slot renderer (asserted in the GPU tests)."""

    def __init__(self, flat_scene, nx, ny, rank, world, device, depth=2, timing=False, options=None):
        self.slots = []
        for _ in range(max(1, depth)):
            ctx = Context(device, timing=timing)
            for k, v in (options or {}).items():
                ctx.set_option(k, v)
            ds = DeviceScene(flat_scene, ctx=ctx)
            self.slots.append((ctx, ds, TileRenderer(ds, nx, ny, rank, world), torch.cuda.Stream(device=device)))
        self.next = 0
        torch.cuda.synchronize(device)  # the slots' buffers were filled on the current stream

    def set_option(self, name, value):
        for ctx, _, _, _ in self.slots:
            ctx.set_option(name, value)

    def step(self, ns, **kw):
        """enqueue one frame on the next slot; returns that slot's TileRenderer (its buffers are valid after sync())"""
        ctx, ds, tr, stream = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        with torch.cuda.stream(stream):
            tr.step(ns, **kw)
        return tr

    def sync(self):
        for _, _, _, stream in self.slots:
            stream.synchronize()

    def last_trace_ms(self):
        """(sum of trace-kernel durations in ms, number of launches) over all slots since the last call"""
        ms, n = 0.0, 0
        for ctx, _, _, _ in self.slots:
            a, b = ctx.last_trace_ms()
            ms, n = ms + a, n + b
        return ms, n

    def close(self):
        for ctx, ds, _, _ in self.slots:
            ds.close()
            ctx.close()
        self.slots = []
