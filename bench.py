#!/usr/bin/env python3
"""bench.py -- Msamples/s of the per-pixel Monte-Carlo sampling path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one frame of synthetic input (the seeded Shirley cover scene): every
rank renders its round-robin 8x8 tiles of the frame from the HBM-resident scene, one gather brings the tiles to rank 0,
rank 0 assembles + quantises.  At N=1 the workload is BASELINE configs[1] (800x400x64spp, cover scene n=11, depth 50);
at N>1 the per-GPU work is kept fixed (weak scaling): the same frame at 64*N spp.  Inputs and outputs stay in HBM.
Consecutive steps (frames) alternate between two render slots on two HIP streams (--frames-in-flight, default 2), so a
frame's first workgroups fill the CUs that the previous frame's last, deep paths leave idle; the JSON also carries the
one-frame-in-flight figure ("serial").
One JSON line is printed by rank 0; see DESIGN.md for the roofline arithmetic."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (nx, ny, ns, n, moving, mix)
    "C1": (200, 100, 4, 11, False, (0.8, 0.95)),
    "C2": (800, 400, 64, 11, False, (0.8, 0.95)),
    "C2m": (800, 400, 64, 11, True, (0.8, 0.95)),
    "C3": (1920, 1080, 256, 50, False, (0.8, 0.95)),
    "C4": (3840, 2160, 512, 11, False, (0.8, 0.95)),
    "C5": (1920, 1080, 4096, 11, False, (0.1, 0.2)),
    # section 8(f3): the classic Cornell box (scene.clj:230-316), 18 rectangles through Translate/RotateY/FlipNormals
    "CB": (600, 600, 256, 0, False, None),
    # the scene `lein run` renders as shipped (core.clj:90, scene.clj:415-489): 400 boxes, 1000 instanced spheres, two
    # ConstantMedium volumes, marble / image textures, a moving sphere, a rectangle light
    "FINAL": (500, 500, 128, 0, False, None),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6     # vector FP64 (spec)
FP32_PEAK_TFLOPS = 157.3


def cpu_baseline(flat, nx, ny, ns, budget_s=15.0):
    """The CPU oracle (oracle/rt_oracle.c, a restatement of the reference path -- NOT the JVM) timed on this box's host
    cores (32-pixel chunks over a thread pool, like core.clj:100-108) on a bounded sample of the same workload: every
    k-th row of the frame at full spp, k chosen so the sample takes about budget_s seconds."""
    from oracle.oracle import Oracle
    orc = Oracle("f64")
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(cores, 256)
    t0 = time.time()
    orc.render(flat, nx, ny, ns, 50, 0x5EED0002, region=(0, ny // 2, nx, ny // 2 + 1), nthreads=threads)
    per_row = max(time.time() - t0, 1e-3)
    n_rows = int(max(1, min(ny, budget_s / per_row)))
    rows = [int(round(i * (ny - 1) / max(1, n_rows - 1))) for i in range(n_rows)] if n_rows > 1 else [ny // 2]
    rows = sorted(set(rows))
    samples, t0 = 0, time.time()
    for y in rows:
        orc.render(flat, nx, ny, ns, 50, 0x5EED0002, region=(0, y, nx, y + 1), nthreads=threads)
        samples += nx * ns
    dt = time.time() - t0
    return {"value": round(samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%d evenly spaced rows of the %dx%d frame at %d spp (%d samples, %.1f s); C restatement of the reference path, not the JVM"
                      % (len(rows), nx, ny, ns, samples, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--accel", default="bvh", choices=["flat", "bvh"], help="bvh = bvh-node descent (hitable.clj:97-123, what every reference scene builds); flat = Hitlist scan (hitable.clj:15-26)")
    ap.add_argument("--single", action="store_true", help="do not also time the other acceleration structure")
    ap.add_argument("--scan-variant", type=int, default=-1, help="0 LDS literal, 1 LDS pipelined, 2 SGPR (default: library default)")
    ap.add_argument("--frames-in-flight", type=int, default=2, help="render slots (context + stream) that consecutive steps alternate between")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytrace_clj_amd as r
    from raytrace_clj_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    # RTMI_BENCH_REHEARSAL=1: several ranks share GPU 0 and the gather goes through gloo on the host -- only to rehearse the
    # multi-rank control flow on a one-GPU box; never a measurement
    rehearsal = os.environ.get("RTMI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            rdist.HOST_STAGED_GATHER = True
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    nx, ny, ns1, n, moving, mix = CONFIGS[args.config]
    ns = ns1 * world  # weak scaling: per-GPU work fixed
    if args.config == "CB":
        scene = r.scene.make_cornell_box(nx, ny)
    elif args.config == "FINAL":
        scene = r.scene.make_final(nx, ny)
    else:
        scene = r.scene.make_random_scene(nx, ny, n, moving, mix=mix)
    flat = r.flatten.flatten(scene)
    if args.config in ("CB", "FINAL"):  # the CPU baseline evaluates the nested records like the reference does
        from oracle.tree import attach_tree
        attach_tree(flat, scene["world"])
    options = {}
    if args.blocks_per_cu:
        options["blocks_per_cu"] = args.blocks_per_cu
    if args.scan_variant >= 0:
        options["scan_variant"] = args.scan_variant
    in_flight = max(1, args.frames_in_flight)
    pipe = rdist.FramePipeline(flat, nx, ny, rank, world, local_rank, depth=in_flight, timing=True, options=options)
    pipe1 = pipe if in_flight == 1 else rdist.FramePipeline(flat, nx, ny, rank, world, local_rank, depth=1, timing=True, options=options)
    n_prims = flat.n_prims
    rec = 32 if args.precision == "f64" else 16
    peak_t = FP64_PEAK_TFLOPS if args.precision == "f64" else FP32_PEAK_TFLOPS

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(accel, steps, warmup, pl):
        """time `steps` steps of the path with the given acceleration structure on pipeline `pl`; returns the result fields"""
        pl.set_option("accel", 1 if accel == "bvh" else 0)
        tr = pl.slots[0][2]
        for _ in range(max(warmup, len(pl.slots))):  # every slot allocates its sample workspace on its first frame: never in the timed region
            pl.step(ns, precision=args.precision)
        barrier()
        pl.last_trace_ms()  # reset the event window
        t0 = time.perf_counter()
        for _ in range(steps):
            pl.step(ns, precision=args.precision)
        barrier()
        dt = time.perf_counter() - t0
        trace_ms, launches = pl.last_trace_ms()
        if world > 1:
            coll_dev = "cpu" if rehearsal else "cuda"
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            seg = tr.counters[:1].clone().to(coll_dev)
            dist.all_reduce(seg, op=dist.ReduceOp.SUM)
            segments = int(seg.item())
        else:
            segments = int(tr.counters[0].item())
        samples = nx * ny * ns
        # roofline of the dominant kernel (trace_kernel) on this rank: SURVEY.md 8(d) algorithmic figures per launch
        seg_local, pix_local = int(tr.counters[0].item()), int(tr.counters[1].item())
        launches_per_step = max(1, launches // max(1, steps))
        launch_s = trace_ms / 1e3 / max(1, launches)
        bytes_alg = (seg_local * n_prims * rec / 256.0 + pix_local * 12.0) / launches_per_step
        flops_alg = seg_local * (n_prims * 20.0 + 120.0) / launches_per_step
        achieved = bytes_alg / launch_s / 1e9
        traffic = None
        try:
            t_all = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            e = t_all.get("%s/%s/%s" % (args.config, accel, args.precision))
            if e and world == 1:
                traffic = int((e["fetch_size_kb"] + e["write_size_kb"]) * 1024)
        except (OSError, ValueError, KeyError):
            pass
        return {
            "value": round(samples / (dt / steps) / 1e6, 3), "ms_per_step": round(dt / steps * 1e3, 4), "segments": segments,
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "kernel": "trace_kernel<%s,%s>" % (args.precision, "BVH" if accel == "bvh" else "flat scan + FP32 cull"),
                "launch_ms": round(launch_s * 1e3, 4), "launches_per_step": launches_per_step, "algorithmic_bytes_per_launch": round(bytes_alg),
                "note": "algorithmic bytes = S*N*REC/256 + 12*pixels (SURVEY.md 8d: what the reference's Hitlist scan must touch with a "
                        "256-ray LDS tile); the scene is on-chip (scalar cache / L2), so HBM is not what binds this path; traffic = "
                        "rocprofv3 FETCH_SIZE+WRITE_SIZE of this kernel from profiles/ (the write is the 24 B/sample buffer)",
                "valu": {"achieved": round(flops_alg / launch_s / 1e12, 3), "peak": peak_t, "unit": "TFLOP/s",
                         "frac": round(flops_alg / launch_s / 1e12 / peak_t, 5),
                         "note": "reference-equivalent flops S*(20N+120) per SURVEY.md 8d over the FP64 vector peak; the FP32 cull / "
                                 "BVH skip most of that work, so this can exceed 1 -- it measures work avoided, not ALU utilisation"}}}

    main_accel = args.accel
    res = measure(main_accel, args.steps, args.warmup, pipe)
    serial = measure(main_accel, max(2, min(args.steps, 5)), 1, pipe1) if in_flight > 1 else None
    other = None
    if world == 1 and not args.single:
        other_accel = "flat" if main_accel == "bvh" else "bvh"
        other = measure(other_accel, max(2, min(args.steps, 5)), 1, pipe)

    if rank == 0:
        samples = nx * ny * ns
        out = {
            "metric": "Msamples/sec (nx*ny*ns)", "value": res["value"], "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("%s: %dx%dx%dspp classic Cornell box (%d rectangles via Translate/RotateY/FlipNormals), depth 50, "
                                    "seeded counter RNG; accel=%s; tiles dealt round-robin to %d GPU(s)" % (args.config, nx, ny, ns, n_prims, main_accel, world))
                       if args.config == "CB" else
                       ("%s: %dx%dx%dspp make-final (scene.clj:415-489; %d primitives), depth 50; accel=%s; %d GPU(s)" % (args.config, nx, ny, ns, n_prims, main_accel, world))
                       if args.config == "FINAL" else
                       "%s: %dx%dx%dspp Shirley cover scene n=%d (%d spheres%s), depth 50, thin-lens camera, "
                       "seeded counter RNG; accel=%s; tiles dealt round-robin to %d GPU(s)" % (
                           args.config, nx, ny, ns, n, n_prims, ", moving" if moving else "", main_accel, world),
                       "nx": nx, "ny": ny, "ns": ns, "spheres": n_prims, "depth": 50, "accel": main_accel,
                       "frames_in_flight": in_flight, "segments_per_sample": round(res["segments"] / samples, 4)},
            "roofline": res["roofline"],
        }
        if serial is not None:
            # The roofline prices ONE launch of the dominant kernel.  With two frames in flight a launch is queued behind the
            # previous frame's launch on the other stream, so its HIP-event interval covers the wait for CU slots as well as
            # the run; the kernel's own duration is what the one-frame-in-flight steps (timed in this same run, same events)
            # measure -- the roofline uses those, and the queued interval is reported beside it.
            out["serial"] = {"value": serial["value"], "ms_per_step": serial["ms_per_step"], "steps": max(2, min(args.steps, 5)),
                             "note": "the same steps with one frame in flight (no overlap between consecutive frames)"}
            out["roofline"] = dict(serial["roofline"])
            out["roofline"]["pipelined_launch_ms"] = res["roofline"]["launch_ms"]
            out["roofline"]["note"] += ("; launch_ms / achieved are measured on the one-frame-in-flight steps of this run (`serial`): with "
                                        "frames_in_flight = %d a launch waits for the previous frame's workgroups to vacate the CUs, and its "
                                        "event interval (pipelined_launch_ms) includes that wait" % in_flight)
        if other is not None:
            out["other_accel"] = {"accel": other_accel, "value": other["value"], "ms_per_step": other["ms_per_step"], "roofline": other["roofline"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(flat, nx, ny, ns)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
