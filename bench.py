#!/usr/bin/env python3
"""bench.py -- Msamples/s of the per-pixel Monte-Carlo sampling path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one frame of synthetic input (the seeded Shirley cover scene): every GPU renders
its round-robin 8x8 tiles of the frame from the HBM-resident scene, one gather brings the tiles to GPU 0, GPU 0 assembles +
quantises.  Inputs and outputs stay in HBM.

  N = 1  BASELINE configs[2], the configuration the north star's target is quoted on: C3 = 1920x1080x256spp, cover scene
         n=50 (10 003 spheres), depth 50, BVH.  One frame in flight; `value` = samples / wall time of the K steps.  Extra keys
         (outside the timed region): "pipelined" (two frames in flight), "c2" (configs[1]), "c4" (configs[3] on one GPU, the
         N = 1 point of the strong-scaling curve), "cpu_baseline".
  N > 1  BASELINE configs[3]: C4 = 3840x2160x512spp cover scene n=11, STRONG scaling (the frame is fixed, its tiles are dealt
         round-robin over the N GPUs), the gather reported separately ("gather_ms").  Two launch forms, same arithmetic:
           * `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (the driver's form): one process per GPU,
             tiles gathered by torch.distributed (RCCL);
           * `python bench.py --gpus N` (one host process, as the reference's one JVM): rtmi_render_multi_device, the gather is
             ONE ncclGather inside librtmi.so.  With fewer than N visible devices the replicas share devices and the JSON says
             "rehearsal": true (control flow only, never a measurement).

One JSON line is printed by rank 0; DESIGN.md section 6 explains the roofline object."""
import argparse
import hashlib
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
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
L2_PEAK_GBS = 34500.0        # MI355X_MICROARCH.md: L2 aggregate ~34.5 TB/s
SIMDS, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; VALU issue peak = 1024 x 2.4e9 SIMD-cycles / s
FP64_PEAK_TFLOPS = 78.6
FP32_PEAK_TFLOPS = 157.3


def kernel_sha():
    """identifies the DEVICE code a PMC profile was taken with: rtmi_device.h and the kernel section of rtmi.hip (everything above
    the host-side C-ABI), so that host-only edits do not mark a profile stale"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytrace_clj_amd", "csrc")
    h.update(open(os.path.join(d, "rtmi_device.h"), "rb").read())
    h.update(open(os.path.join(d, "rtmi.hip"), "rb").read().split(b"// host side: C-ABI")[0])
    return h.hexdigest()[:12]


def build_scene(r, name):
    nx, ny, ns, n, moving, mix = CONFIGS[name]
    if name == "CB":
        scene = r.scene.make_cornell_box(nx, ny)
    elif name == "FINAL":
        scene = r.scene.make_final(nx, ny)
    else:
        scene = r.scene.make_random_scene(nx, ny, n, moving, mix=mix)
    return scene, r.flatten.flatten(scene)


def workload_text(name, nx, ny, ns, n_prims, accel, world):
    n, moving = CONFIGS[name][3], CONFIGS[name][4]
    if name == "CB":
        return "%s: %dx%dx%dspp classic Cornell box (%d rectangles via Translate/RotateY/FlipNormals), depth 50; accel=%s; %d GPU(s)" % (name, nx, ny, ns, n_prims, accel, world)
    if name == "FINAL":
        return "%s: %dx%dx%dspp make-final (scene.clj:415-489; %d primitives), depth 50; accel=%s; %d GPU(s)" % (name, nx, ny, ns, n_prims, accel, world)
    return ("%s: %dx%dx%dspp Shirley cover scene n=%d (%d spheres%s), depth 50, thin-lens camera, seeded counter RNG; accel=%s; "
            "8x8 tiles dealt round-robin to %d GPU(s)" % (name, nx, ny, ns, n, n_prims, ", moving" if moving else "", accel, world))


def cpu_baseline(scene, flat, nx, ny, ns, budget_s=15.0):
    """The CPU oracle (oracle/rt_oracle.c, a restatement of the reference path -- NOT the JVM) timed on this box's host
    cores (32-pixel chunks over a thread pool, like core.clj:100-108) on a bounded sample of the same workload: evenly
    spaced rows of the frame at full spp, as many as fit in about budget_s seconds.  It walks the world as the reference
    builds it (nested bvh-node records, both children visited: hitable.clj:97-123)."""
    import copy
    from oracle.oracle import Oracle
    from oracle.tree import attach_tree
    fl = copy.copy(flat)
    attach_tree(fl, scene["world"])
    orc = Oracle("f64")
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(cores, 256)
    t0 = time.time()
    orc.render(fl, nx, ny, ns, 50, 0x5EED0002, region=(0, ny // 2, nx, ny // 2 + 1), nthreads=threads)
    per_row = max(time.time() - t0, 1e-3)
    n_rows = int(max(1, min(ny, budget_s / per_row)))
    rows = [int(round(i * (ny - 1) / max(1, n_rows - 1))) for i in range(n_rows)] if n_rows > 1 else [ny // 2]
    rows = sorted(set(rows))
    samples, t0 = 0, time.time()
    for y in rows:
        orc.render(fl, nx, ny, ns, 50, 0x5EED0002, region=(0, y, nx, y + 1), nthreads=threads)
        samples += nx * ns
    dt = time.time() - t0
    return {"value": round(samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%d evenly spaced rows of the %dx%d frame at %d spp (%d samples, %.1f s); C restatement of the reference path "
                      "walking the reference's own bvh-node tree, not the JVM" % (len(rows), nx, ny, ns, samples, dt)}


def fp_flops(classes, lanes_active, launch_s, share=1.0):
    """executed floating-point work of one launch from the PMC instruction classes: wave-instructions x 64 lanes x the share of lanes that are live
    (an FMA counts two operations); FP64 against the 78.6 TFLOP/s vector peak, FP32 against 157.3"""
    if not classes or not lanes_active or launch_s <= 0:
        return None
    f64 = (2.0 * classes.get("FMA_F64", 0.0) + classes.get("ADD_F64", 0.0) + classes.get("MUL_F64", 0.0) + classes.get("TRANS_F64", 0.0)) * share
    f32 = (2.0 * classes.get("FMA_F32", 0.0) + classes.get("ADD_F32", 0.0) + classes.get("MUL_F32", 0.0) + classes.get("TRANS_F32", 0.0)) * share
    t64, t32 = f64 * 64.0 * lanes_active / launch_s / 1e12, f32 * 64.0 * lanes_active / launch_s / 1e12
    return {"fp64_tflops": round(t64, 2), "fp32_tflops": round(t32, 2), "fp64_peak": FP64_PEAK_TFLOPS, "fp32_peak": FP32_PEAK_TFLOPS,
            "frac": round(t64 / FP64_PEAK_TFLOPS + t32 / FP32_PEAK_TFLOPS, 4)}


def roofline(cfg, accel, precision, n_prims, launch_s, launches_per_step, counts, share=1.0):
    """The dominant kernel (trace_kernel) priced per launch.  accel: the acceleration structure that RAN ("bvh" / "flat": a small mixed-kind scene is
    scanned even when the tree was asked for) -- it names the kernel and keys the PMC profile.  counts: per FRAME on this GPU {segments, samples, pixels,
    aabb_tests, prim_tests} from the device counters (live, counting instantiation run outside the timed region).
    share: fraction of the profiled whole-frame instruction counts this GPU executes (1/N under the tile partition)."""
    per = 1.0 / max(1, launches_per_step)
    S, smp, pix = counts["segments"] * per, counts["samples"] * per, counts["pixels"]
    rec = 32 if precision == "f64" else 16
    out = {"kernel": "trace_kernel<%s,%s>" % (precision, "BVH" if accel == "bvh" else "flat scan"),
           "launch_ms": round(launch_s * 1e3, 4), "launches_per_step": launches_per_step}
    scene_bytes = n_prims * (32.0 + 96.0 + 8.0) + 4096.0  # node record + exact record + (kind, material) per primitive, + materials / camera
    if accel == "bvh":
        visits, leaves = counts["aabb_tests"] * per / 2.0, counts["prim_tests"] * per
        fetch = visits * 32.0 + leaves * 32.0 + S * 96.0
        fetch_model = "inner-node visits x 32 (one half-plane node record) + exact primitive tests x 32 (cx cy cz r^2) + segments x 96 (material record)"
    else:
        fetch = S * n_prims * rec / 256.0
        fetch_model = "S*N*REC/256 (SURVEY.md 8d: the Hitlist scan with a 256-ray LDS tile)"
        out["flat_scan_flops"] = {"achieved": round(S * (n_prims * 20.0 + 120.0) / launch_s / 1e12, 3), "unit": "TFLOP/s",
                                  "peak": FP64_PEAK_TFLOPS if precision == "f64" else FP32_PEAK_TFLOPS,
                                  "note": "reference-equivalent flops S*(20N+120) of SURVEY.md 8d; the FP32 cull skips most FP64 work, so this "
                                          "is work avoided, not ALU utilisation"}
    # HBM: what the launch must move to / from memory at least once -- the scene in, one colour per sample out
    bytes_alg = scene_bytes + smp * 24.0
    hbm_gbs = bytes_alg / launch_s / 1e9
    out["hbm"] = {"achieved": round(hbm_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 5),
                  "algorithmic_bytes_per_launch": round(bytes_alg), "model": "scene once (%d B) + samples x 24 B (the per-sample colour the in-order reduction reads back)" % scene_bytes,
                  "target_40pct_met": False,
                  "note": "the north star's >= 40 % of the HBM-read roofline is NOT met and cannot be by this design: the scene (<= 1.4 MB) stays on "
                          "chip, so HBM sees the per-sample colours, not scene reads (SURVEY.md 8d says the same of any on-chip-reuse design)"}
    # the per-ray fetches the traversal and shading issue (served by L1 / L2, not HBM)
    fetch_gbs = fetch / launch_s / 1e9
    out["on_chip_fetch"] = {"achieved": round(fetch_gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(fetch_gbs / L2_PEAK_GBS, 4),
                            "bytes_per_launch": round(fetch), "model": fetch_model, "as_fraction_of_hbm_peak": round(fetch_gbs / HBM_PEAK_GBS, 4),
                            "note": "per-lane node / primitive / material fetches over the launch time, against the aggregate L2 bandwidth; the same "
                                    "bytes against the HBM peak (as_fraction_of_hbm_peak) would be near or above 1: the caches' reuse of the scene is "
                                    "what keeps this path off the HBM roofline"}
    prof = None
    try:
        allp = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
        prof = allp.get("%s/%s/%s" % (cfg, accel, precision))
    except (OSError, ValueError):
        pass
    if prof:
        sha = kernel_sha()
        # the profile's counters are per launch of the profiled (one-GPU) run: x its launches per frame = per frame; this GPU executes
        # `share` of the frame in `launches_per_step` launches
        share = share * prof.get("launches_per_frame", 1) / max(1, launches_per_step)
        fp64 = prof["valu_fp64"] * share
        other = (prof["valu_total"] - prof["valu_fp64"]) * share
        peak = SIMDS * CLOCK_GHZ
        traffic = int((prof["fetch_kb"] + prof["write_kb"]) * 1024 * share)
        classes = prof.get("valu_classes")
        prices = None
        try:
            prices = json.load(open(os.path.join(ROOT, "profiles", "valu_prices.json")))["classes"]
        except (OSError, ValueError, KeyError):
            pass
        # the hardware's own busy time: SQ_ACTIVE_INST_VALU counts 4-cycle quanta per wave instruction in flight -- the HEADLINE fraction
        counter = prof["valu_active_quad_cycles"] * 4.0 * share / (SIMDS * launch_s * CLOCK_GHZ * 1e9) if prof.get("valu_active_quad_cycles") else None
        if classes and prices:
            # the priced model (a cross-check, never clamped): every PMC instruction class at the issue cost measured for it on this chip
            # (scripts/ubench/valu_cost.hip at the kernel's 4 waves per SIMD), the instructions no class names at the price of their mix
            named = sum(classes.values())
            cnt = dict(classes, OTHER=max(0.0, prof["valu_total"] - named))
            cyc = {k: sum(cnt[c] * prices[c][k] for c in cnt) * share for k in ("price", "low", "high")}
            m_frac = cyc["price"] / launch_s / 1e9 / peak
            model = {"frac": round(m_frac, 4), "frac_low": round(cyc["low"] / launch_s / 1e9 / peak, 4), "frac_high": round(cyc["high"] / launch_s / 1e9 / peak, 4),
                     "per_class_price_cycles": {c: prices[c]["price"] for c in cnt}, "instructions_per_launch": {c: round(cnt[c] * share) for c in cnt},
                     "prices_source": "profiles/valu_prices.json <- profiles/round3_ubench_valu_cost.txt",
                     "note": "sum over the PMC instruction classes of count x measured issue cost, over the SIMD-cycles of the launch; un-clamped: a value above 1 or "
                             "more than a few per cent above the counter means the class prices (chiefly the un-named 'OTHER' mix) are too high, not that the chip "
                             "ran faster than its clock"}
            if counter:
                model["over_counter"] = round(m_frac / counter, 4)
                model["agrees_with_counter"] = bool(m_frac <= 1.0 and abs(m_frac / counter - 1.0) <= 0.08)
        else:  # profiles taken before round 3 hold no per-class counts: the FMA-calibrated two-price model (FP64 x 4, everything else x 2)
            m_frac = (fp64 * 4.0 + other * 2.0) / launch_s / 1e9 / peak
            model = {"frac": round(m_frac, 4), "note": "no per-class counts in this profile: FP64 x 4 + other x 2 cycles"}
        frac = counter if counter is not None else m_frac
        lanes = prof.get("lanes_active")
        out.update({"bound": "valu", "achieved": round(frac * peak, 1), "peak": peak, "unit": "G SIMD issue-cycles/s", "frac": round(frac, 4),
                    "frac_source": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x launch time): the hardware counter" if counter is not None else "priced instruction classes (no SQ_ACTIVE_INST_VALU in this profile)",
                    "traffic": traffic, "issue_model": model, "lanes_active": lanes,
                    "useful_lane_frac": round(frac * lanes, 4) if lanes else None,
                    "fp": fp_flops(classes, lanes, launch_s, share),
                    "counts": {"valu_instructions_per_launch": round(prof["valu_total"] * share), "fp64_instructions_per_launch": round(fp64),
                               "fetch_kb": prof["fetch_kb"], "write_kb": prof["write_kb"], "source": prof.get("source"),
                               "kernel_sha": prof.get("kernel_sha"), "stale": prof.get("kernel_sha") != sha,
                               "note": "rocprofv3 PMC counters of this workload's trace_kernel launch (deterministic per launch), imported from "
                                       "profiles/pmc_counters.json; launch_ms is measured live (HIP events on the launch stream)"},
                    "note": "binding resource = VALU issue under per-lane traversal divergence.  frac = the share of the SIMDs' issue cycles in which a vector instruction was "
                            "in flight (occupancy of the issue slots, NOT efficiency); lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = the share of those "
                            "slots' lanes that are live; useful_lane_frac = frac x lanes_active = the share of the VALU lane-slot peak that executes anything; fp.frac = "
                            "executed floating-point operations against the FP64 / FP32 vector peaks (the rest of the issued instructions are integer, selects, compares, moves)"})
        out["fp_frac"] = out["fp"]["frac"] if out.get("fp") else None
        out["hbm"]["measured_traffic_gbs"] = round(traffic / launch_s / 1e9, 2)
        out["hbm"]["measured_traffic_frac"] = round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 5)
    else:
        out.update({"bound": "valu", "achieved": None, "peak": SIMDS * CLOCK_GHZ, "unit": "G SIMD issue-cycles/s", "frac": None, "traffic": None,
                    "note": "no PMC profile of this workload (%s/%s/%s) under profiles/pmc_counters.json: only the algorithmic HBM figure is priced" % (cfg, accel, precision)})
    return out


def one_frame_host(r, flat, nx, ny, ns, precision, accel):
    """What a host that renders ONE frame per process pays (core.clj:73-113: -main builds the scene, renders it once, saves it): scene creation (records, the
    device's tree and entry grid, upload) + one frame through the host entry rtmi_render (frame copied back), on a fresh context -- outside the timed region."""
    from raytrace_clj_amd import core
    import ctypes as C
    ctx = core.Context(0)
    ctx.set_option("accel", 1 if accel == "bvh" else 0)
    ds0 = core.DeviceScene(flat, ctx=ctx)  # (first use of the context: lazy runtime initialisation stays out of the figures)
    ds0.render(16, 16, 1)
    ds0.close()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        ds = core.DeviceScene(flat, ctx=ctx)
        t1 = time.perf_counter()
        ds.render(nx, ny, ns, precision=precision)
        t2 = time.perf_counter()
        nbytes = C.c_int64()
        core.check(r._ffi.lib().rtmi_scene_device_bytes(ds.handle, C.byref(nbytes)))
        ds.close()
        cur = {"scene_create_ms": round((t1 - t0) * 1e3, 3), "first_frame_ms": round((t2 - t1) * 1e3, 3), "end_to_end_ms": round((t2 - t0) * 1e3, 3), "upload_bytes": int(nbytes.value)}
        if best is None or cur["end_to_end_ms"] < best["end_to_end_ms"]:
            best = cur
    ctx.close()
    best["note"] = ("one-frame host (core.clj:73-113): rtmi_scene_create_ex (flat records -> device tree + entry grid -> upload) + ONE frame through the host entry "
                    "rtmi_render incl. the copy of the frame back over PCIe (the sample workspace already allocated); best of 3; not part of `value`")
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="default: C3 at --gpus 1, C4 at --gpus N > 1")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the one-frame-host figure (scene_create_ms / upload_bytes / end_to_end_ms); profiling runs use it so that every trace_kernel dispatch is a timed-workload launch")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the pipelined / C2 / C4 / other-accel extra measurements")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--accel", default="bvh", choices=["flat", "bvh"], help="bvh = bvh-node descent (hitable.clj:97-123, what every reference scene builds); flat = Hitlist scan (hitable.clj:15-26)")
    ap.add_argument("--single", action="store_true", help="same as --no-extras")
    ap.add_argument("--scan-variant", type=int, default=-1, help="0 LDS literal, 1 LDS pipelined, 2 SGPR (default: library default)")
    ap.add_argument("--frames-in-flight", type=int, default=1, help="render slots (context + stream) that consecutive steps alternate between (N = 1)")
    args = ap.parse_args()
    extras = not (args.no_extras or args.single)

    import torch
    import torch.distributed as dist

    import raytrace_clj_amd as r
    from raytrace_clj_amd import dist as rdist

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if env_world > 1 and args.gpus != env_world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, env_world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    multi_proc = env_world > 1                      # one process per GPU (torch.distributed.run)
    in_library = args.gpus > 1 and not multi_proc   # one host process, rtmi_render_multi_device
    world = args.gpus
    visible = torch.cuda.device_count()
    # RTMI_BENCH_REHEARSAL=1 (multi-process form): the ranks share GPU 0 and the gather goes through gloo on the host -- only to
    # rehearse the multi-rank control flow on a one-GPU box; never a measurement
    rehearsal = (multi_proc and os.environ.get("RTMI_BENCH_REHEARSAL") == "1") or (in_library and visible < world)
    if in_library and not rehearsal:
        os.environ["RTMI_MULTI_GATHER"] = "rccl"  # distinct devices: the gather is the in-library ncclGather or the run fails -- never silent peer copies
    # RTMI_BENCH_SHARE_DEVICE=1 (test hook): a mis-launched job -- every rank binds device 0 -- that does NOT call itself a rehearsal: the line must refuse
    share_dev = multi_proc and os.environ.get("RTMI_BENCH_SHARE_DEVICE") == "1"
    if multi_proc and (rehearsal or share_dev):
        local_rank = 0
    torch.cuda.set_device(local_rank if multi_proc else 0)
    if multi_proc:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal or share_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            rdist.HOST_STAGED_GATHER = True
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    cfg = args.config or ("C3" if world == 1 else "C4")
    nx, ny, ns = CONFIGS[cfg][:3]
    scene, flat = build_scene(r, cfg)
    options = {}
    if args.blocks_per_cu:
        options["blocks_per_cu"] = args.blocks_per_cu
    if args.scan_variant >= 0:
        options["scan_variant"] = args.scan_variant
    n_prims = flat.n_prims

    def barrier():
        for d in range(visible if in_library else 0):
            torch.cuda.synchronize(d)
        torch.cuda.synchronize()
        if multi_proc:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the step, in the three launch forms ---------------------------------------------------------------------------
    class OneGpu:  # N = 1 and the per-rank driver of the multi-process form
        def __init__(self, flat, nx, ny, in_flight):
            self.pl = rdist.FramePipeline(flat, nx, ny, rank if multi_proc else 0, world if multi_proc else 1, local_rank if multi_proc else 0,
                                          depth=in_flight, timing=True, options=options)
            self.tr = self.pl.slots[0][2]
            self.slots = len(self.pl.slots)

        def set_option(self, k, v):
            self.pl.set_option(k, v)

        def step(self, ns):
            self.pl.step(ns, precision=args.precision)

        def trace_ms(self):
            return self.pl.last_trace_ms()

        def reduce_ms(self):
            return self.pl.last_reduce_ms()

        def counters(self):
            return int(self.tr.counters[0].item()), int(self.tr.counters[1].item())

        def traversal(self):
            return self.pl.slots[0][0].last_traversal_counters()

        def accel_ran(self):
            return self.pl.slots[0][0].last_accel()

        def gather_ms(self):
            return self.tr.last_gather_ms() if multi_proc else None

        def gather_path(self):
            if not multi_proc:
                return "none"
            return "torch.distributed.gather, backend %s%s" % (dist.get_backend(), " (host-staged: rehearsal)" if rehearsal else (" = RCCL" if dist.get_backend() == "nccl" else ""))

    class InLibrary:  # one host process, N devices behind the C-ABI
        def __init__(self, flat, nx, ny):
            devs = [i % visible for i in range(world)]
            self.md = rdist.MultiDevice(flat, devs, timing=True, options=options)
            d0 = torch.device("cuda", devs[0])
            self.lin = torch.zeros((ny, nx, 3), dtype=torch.float64, device=d0)
            self.q = torch.zeros((ny, nx, 3), dtype=torch.uint8, device=d0)
            self.cnt = torch.zeros(2, dtype=torch.int64, device=d0)
            self.nx, self.ny, self.slots = nx, ny, 1
            torch.cuda.synchronize(d0)

        def set_option(self, k, v):
            self.md.set_option(k, v)

        def step(self, ns):
            self.md.render_device(self.nx, self.ny, ns, self.lin, self.q, self.cnt, precision=args.precision)

        def trace_ms(self):
            per = self.md.last_trace_ms()
            self.per_replica = per  # every replica's (ms, launches) of the window: the line reports min / max, the roofline prices the slowest
            return max(per, key=lambda x: x[0])

        def reduce_ms(self):
            return self.md.ctxs[0].last_reduce_ms()

        def counters(self):
            self.md.sync()
            return int(self.cnt[0].item()), int(self.cnt[1].item())

        def traversal(self):
            a = [c.last_traversal_counters() for c in self.md.ctxs]
            return sum(x[0] for x in a), sum(x[1] for x in a)

        def accel_ran(self):
            return self.md.ctxs[0].last_accel()

        def gather_ms(self):
            return self.md.last_gather_ms()

        def gather_path(self):
            return self.md.last_gather_path()

    def measure(drv, accel, steps, warmup, cfg_name, nx, ny, ns, n_prims, count=True):
        drv.set_option("accel", 1 if accel == "bvh" else 0)
        drv.set_option("count_traversal", 0)
        for _ in range(max(warmup, drv.slots)):  # every slot allocates its sample workspace on its first frame: never in the timed region
            drv.step(ns)
        barrier()
        drv.trace_ms()  # reset the event window
        t0 = time.perf_counter()
        for _ in range(steps):
            drv.step(ns)
        barrier()
        dt = time.perf_counter() - t0
        dt_local = dt
        trace_ms, launches = drv.trace_ms()
        per_replica = list(getattr(drv, "per_replica", None) or [])  # in-library form: every replica's (ms, launches) of the timed window
        reduce_ms, reduce_launches = drv.reduce_ms()
        gather_ms = drv.gather_ms() if world > 1 else None
        seg_local, pix_local = drv.counters()
        accel_ran = drv.accel_ran()  # small mixed-kind scenes answer a request for the tree with the scan (option flat_below)
        segments = seg_local
        if multi_proc:
            coll_dev = "cpu" if (rehearsal or share_dev) else "cuda"
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            seg = torch.tensor([seg_local], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(seg, op=dist.ReduceOp.SUM)
            segments = int(seg.item())
        samples = nx * ny * ns
        counts = {"segments": seg_local if multi_proc else segments / (world if in_library else 1), "samples": samples / world, "pixels": nx * ny / world,
                  "aabb_tests": 0, "prim_tests": 0}
        if count and accel == "bvh":  # one more frame with the counting instantiation (outside the timed region): node visits / leaf tests
            drv.set_option("count_traversal", 1)
            drv.step(ns)
            barrier()
            a, b = drv.traversal()
            drv.set_option("count_traversal", 0)
            drv.trace_ms()
            counts["aabb_tests"], counts["prim_tests"] = (a / world, b / world) if in_library else (a, b)
        launches_per_step = max(1, launches // max(1, steps))
        launch_s = trace_ms / 1e3 / max(1, launches)
        res = {"value": round(samples / (dt / steps) / 1e6, 3), "ms_per_step": round(dt / steps * 1e3, 4), "segments": segments,
               "segments_per_sample": round(segments / samples, 4), "accel_ran": accel_ran,
               "roofline": roofline(cfg_name, accel_ran, args.precision, n_prims, launch_s, launches_per_step, counts, share=1.0 / world),
               "trace_ms_per_step": round(trace_ms / max(1, steps), 4), "dt_local_ms": round(dt_local / steps * 1e3, 4), "segments_local": seg_local,
               "per_replica": per_replica}
        if reduce_launches:  # the one HBM-bound kernel of the path: the in-order per-pixel sample reduction (core.clj:52-53)
            red_s = reduce_ms / 1e3 / reduce_launches
            red_bytes = (counts["samples"] * 24.0 + counts["pixels"] * 24.0 * (2 * launches_per_step - 1)) / launches_per_step  # samples read once; running sums written / re-read between passes
            res["roofline"]["reduce_kernel"] = {"bound": "hbm", "achieved": round(red_bytes / red_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac": round(red_bytes / red_s / 1e9 / HBM_PEAK_GBS, 4), "ms": round(red_s * 1e3, 4),
                                                "bytes_per_launch": round(red_bytes), "share_of_step": round(reduce_ms / steps / (dt / steps * 1e3), 4),
                                                "note": "reduce_kernel: one thread per pixel sums its samples in sample order (24 B each, coalesced 1536-B rows per tile and sample); "
                                                        "duration = HIP events from the trace kernel's end to the reduction's end on the launch stream"}
        if count and accel == "bvh":
            res["aabb_tests_per_segment"] = round(counts["aabb_tests"] / max(1, counts["segments"]), 3)
            res["prim_tests_per_segment"] = round(counts["prim_tests"] / max(1, counts["segments"]), 3)
        if gather_ms is not None:
            res["gather_ms"] = round(gather_ms, 4)
            res["gather_path"] = drv.gather_path()
        return res

    drv = InLibrary(flat, nx, ny) if in_library else OneGpu(flat, nx, ny, max(1, args.frames_in_flight) if world == 1 else 1)
    res = measure(drv, args.accel, args.steps, args.warmup, cfg, nx, ny, ns, n_prims)

    # ---- N > 1, one process per GPU: a line that proves what ran -- every rank reports its device, its own kernel time, segments and tiles --------------
    ranks_info = None
    if multi_proc:
        from raytrace_clj_amd import dist as rd
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        ident = str(getattr(props, "uuid", "") or "") or str(getattr(props, "pci_bus_id", "") or "")
        if hasattr(props, "pci_bus_id"):
            ident = "%s pci %04x:%02x:%02x" % (ident, getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
        mine = {"rank": rank, "device_index": int(torch.cuda.current_device()), "device": ident.strip(), "device_name": props.name,
                "trace_ms": res["trace_ms_per_step"], "step_ms": res["dt_local_ms"], "segments": int(res["segments_local"]),
                "tiles": len(rd.local_tile_ids(nx, ny, rank, world)), "gather_ms": res.get("gather_ms")}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        tr = [e["trace_ms"] for e in everyone]
        sg = [e["segments"] for e in everyone]
        devices = [e["device"] or ("index %d" % e["device_index"]) for e in everyone]
        distinct = len(set(devices)) == world
        ranks_info = {"n": dist.get_world_size(), "backend": dist.get_backend(), "per_rank": everyone, "distinct_devices": distinct,
                      "trace_ms_min": round(min(tr), 4), "trace_ms_max": round(max(tr), 4),
                      "imbalance": round(max(tr) / (sum(tr) / len(tr)), 4) if sum(tr) > 0 else None,
                      "segments_imbalance": round(max(sg) / (sum(sg) / len(sg)), 4) if sum(sg) > 0 else None,
                      "gather_bytes": int((world - 1) * rd.tiles_per_rank(nx, ny, world) * 64 * 3 * 8),
                      "note": "imbalance = slowest rank's trace-kernel time per step / the mean over ranks; gather_bytes = what arrives at rank 0 per frame "
                              "(world - 1 records of tiles_per_rank x 64 x 3 doubles); gather_ms (top level) = rank 0's gather interval = transfer + wait for the slowest rank"}
        if not distinct and not rehearsal:
            if rank == 0:
                print(json.dumps({"error": "two ranks report the same device", "ranks": ranks_info}), flush=True)
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
    out = None
    if rank == 0:
        out = {
            "metric": "Msamples/sec (nx*ny*ns)", "value": res["value"], "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": workload_text(cfg, nx, ny, ns, n_prims, args.accel, world), "nx": nx, "ny": ny, "ns": ns, "spheres": n_prims,
                       "depth": 50, "accel": args.accel, "accel_ran": res["accel_ran"], "frames_in_flight": drv.slots, "segments_per_sample": res["segments_per_sample"],
                       "launch_form": "one process per GPU (torch.distributed, RCCL gather)" if multi_proc else
                                      ("one host process, rtmi_render_multi_device (in-library ncclGather)" if in_library else "one GPU")},
            "roofline": res["roofline"],
        }
        for k in ("aabb_tests_per_segment", "prim_tests_per_segment", "gather_ms", "gather_path"):
            if k in res:
                out[k] = res[k]
        if in_library:  # every replica's launches, not replica 0's
            per = res.get("per_replica") or []
            ms = [a / max(1, args.steps) for a, b in per]  # ms of trace kernel per step, per replica
            out["replicas"] = {"n": world, "devices": [i % visible for i in range(world)], "trace_ms_per_step": [round(x, 4) for x in ms],
                               "trace_ms_min": round(min(ms), 4) if ms else None, "trace_ms_max": round(max(ms), 4) if ms else None,
                               "imbalance": round(max(ms) / (sum(ms) / len(ms)), 4) if ms and sum(ms) > 0 else None}
        if multi_proc:
            out["ranks"] = ranks_info
        if world > 1:
            out["config"]["scaling_note"] = ("strong scaling of BASELINE configs[3] (the frame is fixed, its 8x8 tiles are dealt round-robin over the GPUs); the "
                                             "N = 1 point of this curve is the 'c4' key of the `--gpus 1` line (same frame on one GPU), not that line's C3 value")
        if rehearsal:
            out["rehearsal"] = True
            out["config"]["workload"] += " -- REHEARSAL: the %d replicas share %d device(s); control flow only, not a measurement" % (world, visible)
    # ---- extras (N = 1 only, outside the timed region) -------------------------------------------------------------------
    if world == 1 and extras and rank == 0:
        if drv.slots == 1:
            two = OneGpu(flat, nx, ny, 2)
            p = measure(two, args.accel, max(2, min(args.steps, 5)), 2, cfg, nx, ny, ns, n_prims, count=False)
            out["pipelined"] = {"value": p["value"], "ms_per_step": p["ms_per_step"], "frames_in_flight": 2,
                                "note": "the same steps with two frames in flight on two streams: a frame's first workgroups fill the CUs the "
                                        "previous frame's last deep paths leave idle; only for hosts that render frame after frame"}
            two.pl.close()
        for other in ("C2", "C4", "C5"):  # BASELINE configs[1], [3] (on one GPU) and [4] (the divergence-stress configuration)
            if other == cfg:
                continue
            onx, ony, ons = CONFIGS[other][:3]
            _, oflat = build_scene(r, other)
            od = OneGpu(oflat, onx, ony, 1)
            o = measure(od, args.accel, 10 if other == "C2" else 2, 2 if other == "C2" else 1, other, onx, ony, ons, oflat.n_prims)
            out[other.lower()] = {"workload": workload_text(other, onx, ony, ons, oflat.n_prims, args.accel, 1), "value": o["value"],
                                  "ms_per_step": o["ms_per_step"], "segments_per_sample": o["segments_per_sample"], "roofline": o["roofline"]}
            for k in ("aabb_tests_per_segment", "prim_tests_per_segment"):
                if k in o:
                    out[other.lower()][k] = o[k]
            od.pl.close()
        if cfg in ("C1", "C2", "C2m", "CB", "FINAL"):  # the other acceleration structure where it finishes in seconds
            other_accel = "flat" if args.accel == "bvh" else "bvh"
            o = measure(drv, other_accel, max(2, min(args.steps, 5)), 1, cfg, nx, ny, ns, n_prims)
            out["other_accel"] = {"accel": other_accel, "value": o["value"], "ms_per_step": o["ms_per_step"], "roofline": o["roofline"]}
    if rank == 0:
        if world == 1 and not args.no_e2e:  # section 8(d): scene upload reported separately and included in an end-to-end figure
            drv_close = getattr(getattr(drv, "pl", None), "close", None)
            if drv_close:
                drv_close()  # release the timed driver's workspace first: the one-frame host allocates its own
            e2e = one_frame_host(r, flat, nx, ny, ns, args.precision, args.accel)
            out["scene_create_ms"], out["upload_bytes"], out["end_to_end_ms"] = e2e["scene_create_ms"], e2e["upload_bytes"], e2e["end_to_end_ms"]
            out["one_frame_host"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, flat, nx, ny, ns)
        print(json.dumps(out), flush=True)
    if multi_proc:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
