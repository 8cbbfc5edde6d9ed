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
        self.dev, self._side, self._ev = dev, None, None
        self.local = torch.zeros((self.per, 64, 3), dtype=torch.float64, device=dev)
        self.counters = torch.zeros(2, dtype=torch.int64, device=dev)
        if rank == 0:
            self.linear = torch.zeros((ny, nx, 3), dtype=torch.float64, device=dev)
            self.rgb8 = torch.zeros((ny, nx, 3), dtype=torch.uint8, device=dev)

    def step(self, ns, depth=50, seed=0x5EED0002, precision="f64"):
        """render local tiles -> gather -> (rank 0) assemble.  Asynchronous, ordered with torch's current stream.

        The C-ABI reads stream handle 0 (NULL) as "the context's own stream", and torch's default stream has handle 0: on
        the default stream the render would run un-ordered with torch's work (the zero-fill of the buffers above, the
        gather).  So from the default stream the whole step runs on a side stream that first waits for the current stream
        and that the current stream then waits for."""
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != 0:
            return self._step(cur.cuda_stream, ns, depth, seed, precision)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self._step(self._side.cuda_stream, ns, depth, seed, precision)
        cur.wait_stream(self._side)

    def _step(self, stream, ns, depth, seed, precision):
        self.ds.render_tiles_device(self.nx, self.ny, ns, self.rank, self.world, self.local, self.counters, depth, seed, precision, stream)
        if self.world > 1:
            if self._ev is None:
                self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        gathered = gather_tiles(self.local, self.world, self.rank)
        if self.world > 1:
            self._ev[1].record()
        if self.rank == 0:
            check(_ffi.lib().rtmi_assemble_device(self.ds.ctx.handle, self.nx, self.ny, self.world, self.per, _ffi.ptr(gathered),
                                                  _ffi.ptr(self.linear), _ffi.ptr(self.rgb8), _ffi.ptr(stream)))


    def last_gather_ms(self):
        """milliseconds the last step's gather took on this rank's stream (the transfer plus the wait for the slowest rank)"""
        if self._ev is None:
            return 0.0
        self._ev[1].synchronize()
        return self._ev[0].elapsed_time(self._ev[1])


class MultiDevice:
    """ONE host process driving several GPUs through the C-ABI (rtmi_render_multi*): what a one-JVM host of the reference
    (core.clj:100-108) calls.  devices = HIP device ordinals, one replica (context + cloned scene) each; a device may be
    listed more than once to rehearse the control flow on a one-GPU host (those replicas are gathered by device copies,
    distinct devices by ONE ncclGather inside the library)."""

    def __init__(self, flat_scene, devices, timing=False, options=None):
        self.ctxs, self.scenes = [], []
        for d in devices:
            ctx = Context(int(d), timing=timing)
            for k, v in (options or {}).items():
                ctx.set_option(k, v)
            self.ctxs.append(ctx)
            self.scenes.append(DeviceScene(flat_scene, ctx=ctx) if not self.scenes else self.scenes[0].clone(ctx))
        import ctypes as C
        self._arr = (C.c_void_p * len(self.scenes))(*[s.handle for s in self.scenes])
        self.n = len(self.scenes)

    def set_option(self, name, value):
        for ctx in self.ctxs:
            ctx.set_option(name, value)

    def render(self, nx, ny, ns, depth=50, seed=0x5EED0002, precision="f64"):
        """host buffers: (linear [ny,nx,3] float64, rgb8, counters)"""
        import numpy as np
        lin, q, cnt = np.zeros((ny, nx, 3)), np.zeros((ny, nx, 3), np.uint8), np.zeros(2, np.uint64)
        check(_ffi.lib().rtmi_render_multi(self.n, self._arr, nx, ny, ns, depth, seed, {"f64": 0, "f32": 1}[precision],
                                           _ffi.ptr(lin), _ffi.ptr(q), _ffi.ptr(cnt)))
        return lin, q, cnt

    def render_device(self, nx, ny, ns, out_linear, out_rgb8, out_counters, depth=50, seed=0x5EED0002, precision="f64"):
        """outputs: device pointers / torch tensors on devices[0]; asynchronous on replica 0's context stream"""
        check(_ffi.lib().rtmi_render_multi_device(self.n, self._arr, nx, ny, ns, depth, seed, {"f64": 0, "f32": 1}[precision],
                                                  _ffi.ptr(out_linear), _ffi.ptr(out_rgb8), _ffi.ptr(out_counters)))

    def sync(self):
        """wait for every replica (a multi render is complete when replica 0's stream is; the others finished before the gather)"""
        for ctx in self.ctxs:
            torch.cuda.synchronize(ctx.device)

    def last_gather_ms(self):
        import ctypes as C
        ms = C.c_double()
        check(_ffi.lib().rtmi_last_gather_ms(self.ctxs[0].handle, C.byref(ms)))
        return ms.value

    def last_gather_path(self):
        """"none" | "same-device" | "peer-copy" | "rccl": how the last render moved the replicas' records to replica 0 (rtmi_last_gather_path)"""
        import ctypes as C
        p = C.c_int32(-1)
        check(_ffi.lib().rtmi_last_gather_path(self.ctxs[0].handle, C.byref(p)))
        return _ffi.GATHER_PATHS[p.value]

    def last_trace_ms(self):
        """per replica: (sum of trace-kernel ms, launches) since the last call"""
        return [ctx.last_trace_ms() for ctx in self.ctxs]

    def close(self):
        for s in self.scenes:
            s.close()
        for c in self.ctxs:
            c.close()
        self.scenes, self.ctxs = [], []


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
        if tr.world == 1 and len(self.slots) > 1:
            # one GPU, several frames in flight: every slot runs on its CONTEXT'S OWN stream (stream = NULL in the C-ABI).  The HIP
            # runtime deals streams to a few hardware queues in creation order; the contexts' streams are created back to back and
            # land on different queues, whereas extra torch streams were seen to share one queue (kernels of two frames then
            # run one after the other, no overlap).  No torch work is queued between frames, so nothing needs ordering with torch.
            tr._step(None, ns, kw.get("depth", 50), kw.get("seed", 0x5EED0002), kw.get("precision", "f64"))
            return tr
        with torch.cuda.stream(stream):
            tr.step(ns, **kw)
        return tr

    def sync(self):
        for ctx, _, _, stream in self.slots:
            stream.synchronize()
        torch.cuda.synchronize(self.slots[0][0].device)  # the contexts' own streams

    def last_trace_ms(self):
        """(sum of trace-kernel durations in ms, number of launches) over all slots since the last call"""
        ms, n = 0.0, 0
        for ctx, _, _, _ in self.slots:
            a, b = ctx.last_trace_ms()
            ms, n = ms + a, n + b
        return ms, n

    def last_reduce_ms(self):
        """(sum of reduce_kernel durations in ms, launches) of the window the last last_trace_ms() closed"""
        ms, n = 0.0, 0
        for ctx, _, _, _ in self.slots:
            a, b = ctx.last_reduce_ms()
            ms, n = ms + a, n + b
        return ms, n

    def close(self):
        for ctx, ds, _, _ in self.slots:
            ds.close()
            ctx.close()
        self.slots = []
