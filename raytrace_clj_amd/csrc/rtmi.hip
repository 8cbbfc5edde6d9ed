// rtmi.hip -- kernels + C-ABI (include/rtmi.h) of the MI355X sampling path.  gfx950 only.
//
// Data flow of one render (all buffers in HBM):
//   scene SoA (static spheres {cx,cy,cz,r} | moving spheres | material / texture tables)
//     -> trace kernel: persistent 256-thread workgroups; the sphere SoA is staged into LDS as
//        {cx,cy,cz,r*r}; every lane owns one path (sample) at a time and runs one `color` iteration
//        (core.clj:17-41) per loop trip; lanes whose path ended are refilled from the workgroup's work
//        list (wave ballot + popcount prefix sum -> one LDS atomic per wave), so the sphere scan always
//        runs with full waves; a finished sample stores its colour at samples[tile][s][pixel]
//     -> reduce kernel: per pixel, sum over s IN SAMPLE ORDER (core.clj:52 reduce mat/add), * 1/ns
//     -> assemble kernel: tile-major -> dense frame, sqrt, *255.99, min, trunc (core.clj:54-56)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: librccl is opened at the first multi-device gather (dlopen), never linked
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>
#include <cmath>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rtmi.h"
#include "rtmi_device.h"

using namespace rtmi;

#define RTMI_EXPORT extern "C" __attribute__((visibility("default")))

// =====================================================================================================
// kernels
// =====================================================================================================
namespace {

constexpr int kBlock = RTMI_TRACE_BLOCK;      // (the probe kernels traverse the tree too: same workgroup size as the trace kernel, see RTMI_BVH_STRIDE)
constexpr int kTraceBlock = RTMI_TRACE_BLOCK; // threads per workgroup of the trace kernel
// Diagnostic build only (make stamps -> librtmi_stamps.so): per-phase wave ticks and lane-ticks (ph_stamp, rtmi_device.h), summed over the
// launch's waves into g_phase and printed to stderr after every render, with the workgroups' start / end times.  Never defined in the
// shipped library.
#ifdef RTMI_STAMPS
__device__ unsigned long long g_wg_t[2][4096]; // s_memrealtime (100 MHz) at workgroup start / end
__device__ inline unsigned long long real_now() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
#define RTMI_STAMP_DECL const unsigned long long st_wg0 = real_now(); \
    if (lane == 0) for (int k_ = 0; k_ < 3 * PH_SLOTS + 2; ++k_) g_ph[threadIdx.x >> 6][k_] = 0; \
    ph_stamp(-1); for (int k_ = 0; k_ < 32; ++k_) ph_stamp(PH_CAL);
#define RTMI_STAMP_FLUSH(cnt) if (lane == 0) for (int k_ = 0; k_ < 3 * PH_SLOTS; ++k_) if (g_ph[threadIdx.x >> 6][k_]) atomicAdd(&g_phase[k_], g_ph[threadIdx.x >> 6][k_]); \
    __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 4096) { g_wg_t[0][blockIdx.x] = st_wg0; g_wg_t[1][blockIdx.x] = real_now(); }
#else
#define RTMI_STAMP_DECL
#define RTMI_STAMP_FLUSH(cnt)
#endif
#ifndef RTMI_MIN_WAVES
#define RTMI_MIN_WAVES 4
#endif
#ifndef RTMI_EXT_MIN_WAVES
#define RTMI_EXT_MIN_WAVES RTMI_MIN_WAVES // waves per SIMD the mixed-kind (EXT) instantiations are compiled for
#endif

struct TraceParams {
    int nx, ny, depth;
    u64 seed;
    int tiles_x;
    int n_local_tiles;
    const int *tile_ids;   // [n_local_tiles] global tile index (row-major over tiles)
    int s_begin, s_count;  // samples [s_begin, s_begin + s_count) of every pixel in this pass
    void *samples;         // [n_local_tiles * s_count][64][3] real
    u64 *counters;         // [0] += ray segments (metrics total-rays, core.clj:24)
    int prims_per_tile;    // static spheres per LDS tile
    int n_ptiles;          // number of LDS tiles the static spheres are cut into
    unsigned *queue;       // work queue head of this pass (zeroed before the launch): next unclaimed work item
    unsigned total_items;  // n_local_tiles * s_count * 64
    unsigned qblock;       // work items a wave claims per queue access (a multiple of 64)
    int suspend_lanes;     // time-sliced traversal: hand the wave back when fewer lanes than this are still in the tree (0: never)
    int rx0, ry0, rx1, ry1; // output region (row 0 = top): pixels outside it are not traced (whole frame: 0, 0, nx, ny)
    u64 *trav;             // COUNT instantiations: [0] += AABB slab tests (metrics aabb.intersection.total, hitable.clj:39), [1] += exact primitive tests
    int susp_off;          // mixed-kind (EXT) BVH kernels: word offset of the parked cursors in LDS = stack levels of THIS scene's tree x RTMI_BVH_STRIDE
    int stash_off;         // mixed-kind kernels: word offset of the camera-ray stash in LDS (behind the stack and the parked cursors; 0 for the scan)
};
#ifndef RTMI_QUEUE_BLOCK
#define RTMI_QUEUE_BLOCK 256
#endif
#ifndef RTMI_EXT_NO_STASH
#define RTMI_EXT_NO_STASH 0 // 1: the EXT (f3 / f4) instantiations refill per trip (17 VGPRs fewer)
#endif
#ifndef RTMI_EXT_LDS_STASH
#define RTMI_EXT_LDS_STASH 1 // the mixed-kind instantiations keep the camera-ray stash in LDS (11 or 17 words per entry), not in 17 VGPRs: they stand at the 128-VGPR limit of 4 waves per SIMD
#endif
#ifndef RTMI_STASH
#define RTMI_STASH 1 // camera rays generated 64 at a time at full wave width into a register stash (0: per trip, for the dead lanes only)
#endif
constexpr unsigned kQueueBlock = RTMI_QUEUE_BLOCK; // work items a wave claims per queue access at least: 4 chunks = one tile x 4 consecutive samples

template <typename R> __device__ inline const R *stat4_of(SceneRef sc);
// Stage static spheres [first, first+count) into LDS as {cx, cy, cz, r*r} (hitable.clj:188: (* radius radius)).
template <typename R> __device__ inline void stage_prims(SceneRef sc, Prim4<R> *lds, int first, int count) {
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const R *g = stat4_of<R>(sc) + (size_t)(first + i) * 4;
        Prim4<R> p;
        p.cx = g[0]; p.cy = g[1]; p.cz = g[2]; p.r2 = g[3];
        lds[i] = p;
    }
}

template <typename R> __device__ inline const R *stat4_of(SceneRef sc);
template <> __device__ inline const double *stat4_of<double>(SceneRef sc) { return sc.stat4_d; }
template <> __device__ inline const float *stat4_of<float>(SceneRef sc) { return sc.stat4_f; }

// the FP32 cull exists for the FP64 path only; RTMI_F32 falls back to the plain scalar-cache scan
__device__ inline void scan_cull_dispatch(SceneRef sc, const Path<double> &P, double a, double tmin, double &best_t, int &best_i) {
    scan_all_cull(sc, P, a, tmin, best_t, best_i);
}
__device__ inline void scan_cull_dispatch(SceneRef sc, const Path<float> &P, float a, float tmin, float &best_t, int &best_i) {
    scan_static_pipe<float, false>(ScalarPrims<float>(sc.stat4_f), sc.n_static, 0, P, a, tmin, best_t, best_i);
    if (best_i >= 0) best_i = sc.stat_orig[best_i];
    if (sc.n_moving > 0) {
        int best_orig = best_i >= 0 ? best_i : 0x7fffffff, scan_i = -1;
        scan_moving<float>(sc, P, a, tmin, best_t, scan_i, best_orig);
        if (scan_i >= 0) best_i = best_orig;
    }
}


// hit? of the whole world for the lane's ray (closest hit, t in (t-min, t-max); core.clj:25 passes 0.001, Float/MAX_VALUE).
// MULTI (LDS variants only): the static spheres do not fit one LDS tile; every thread of the workgroup must call this.
// section 8(f3) scenes (FP64 only): BVH or culled flat scan over mixed primitive kinds with the any-order tie rule
// MSEQ: the instantiations for RTMI_MEDIA_HITLIST (1) and RTMI_MEDIA_NARROWED (2) worlds (their own kernels: the plain mixed-kind kernels keep their registers)
template <bool SLICED = false, bool COUNT = false, int MSEQ = 0>
__device__ inline void intersect_ext(SceneRef sc, int *stack, bool bvh, Path<double> &P, bool active, double tmin, double tmax, double &best_t, int &best_i,
                                     bool *mid = nullptr, int min_lanes = 0, unsigned *cnt = nullptr, int susp_off = RTMI_BVH_STACK * RTMI_BVH_STRIDE) {
    best_t = tmax; best_i = -1;
    if (!active) return;
    const double a = dot3(P.dx, P.dy, P.dz, P.dx, P.dy, P.dz);
    ExtHit H = {tmax, 0x7fffffff, -1};
    if (MSEQ == 2) { // (an instantiation of its own: with both list forms inlined side by side one of the kernels went to 248 registers and scratch)
        // RTMI_MEDIA_NARROWED: Hitlists holding media BELOW bvh-nodes (round 4).  A bvh-node hands its children the un-narrowed interval (hitable.clj:99-105), a Hitlist
        // hands every item the closest hit of the items before it (hitable.clj:15-26): call k of the media sequence sees the closest hit among the primitives
        // [media_lo[k], media_idx[k]) -- the items before it in its own (possibly nested) Hitlist -- or the caller's t-max when that range is empty.  The items of one
        // Hitlist are contiguous in the flattened order, so the narrowing state is a running closest hit over that list: its surfaces are scanned piece by piece
        // between its media (index-restricted scan: such lists are short), every medium's candidate joins it.  The world's closest hit is then the fold of ALL
        // surfaces (one traversal, below) and of the media's candidates -- in any order (ExtHit).
        MediumChord chord = medium_chord_begin(P);
        ExtHit Hrun = {tmax, 0x7fffffff, -1};
        int cur_lo = -1, scanned_to = 0;
        for (int k = 0; k < sc.n_media; ++k) {
            const int m = sc.media_idx[k], lo = sc.media_lo[k];
            if (lo >= m) { ext_medium_test(sc, m, P, tmin, tmax, H, chord, COUNT ? cnt : nullptr); continue; } // un-narrowed: a medium reached through bvh-nodes only
            if (lo != cur_lo) { Hrun.t = tmax; Hrun.F = 0x7fffffff; Hrun.W = -1; cur_lo = lo; scanned_to = lo; }
            if (m > scanned_to) scan_all_cull_ext(sc, P, a, tmin, Hrun, scanned_to, m);
            double tm = 0.0;
            if (ext_medium_test(sc, m, P, tmin, Hrun.any() ? Hrun.t : tmax, Hrun, chord, COUNT ? cnt : nullptr, &tm)) ext_update(H, tm, m, true);
            scanned_to = m + 1;
        }
        if (bvh) scan_bvh_ext<false, COUNT>(sc, stack, P, a, tmin, H, nullptr, false, 0, cnt);
        else if (sc.small_scan) scan_small_ext(sc, P, tmin, H);
        else scan_all_cull_ext(sc, P, a, tmin, H);
        best_i = ext_winner(H);
        if (best_i >= 0) best_t = H.t;
        return;
    }
    if (MSEQ == 1) {
        // RTMI_MEDIA_HITLIST: the world is a Hitlist (hitable.clj:15-26: (hit? item r t-min closest-so-far), item after item).  Surfaces may be
        // folded in any order (ExtHit reproduces the list's tie rule), so the list is scanned in pieces: the surfaces before the first medium,
        // that medium with the t-max the list would hand it -- the closest hit so far --, the surfaces up to the next medium, and so on.  (These
        // scenes run the instantiation without time-slicing: media_seq and SLICED never meet.)
        int prev = 0;
        MediumChord chord = medium_chord_begin(P);
        for (int k = 0; k <= sc.n_media; ++k) {
            const int m = k < sc.n_media ? sc.media_idx[k] : sc.n_all;
            if (m > prev) {
                if (bvh) scan_bvh_ext<false, COUNT>(sc, stack, P, a, tmin, H, nullptr, false, 0, cnt, prev, m);
                else if (sc.small_scan) scan_small_ext(sc, P, tmin, H, prev, m);
                else scan_all_cull_ext(sc, P, a, tmin, H, prev, m);
            }
            if (k < sc.n_media) ext_medium_test(sc, m, P, tmin, H.any() ? H.t : tmax, H, chord, COUNT ? cnt : nullptr);
            prev = m + 1;
        }
        best_i = ext_winner(H);
        if (best_i >= 0) best_t = H.t;
        return;
    }
    // Media FIRST (RTMI_MEDIA_DESCENT: every medium's hit? sees the caller's un-narrowed interval, hitable.clj:99-105, so its result -- and its draws, the only
    // draws of a hit? -- do not depend on the surfaces; ExtHit folds candidates in any order).  The surfaces' traversal then starts with the closest MEDIUM hit as
    // its bound: a path scattering inside make-final's subsurface sphere (mean free path 5; a third of that scene's segments) or ending in its haze no longer
    // walks the tree with an unbounded interval before its medium is asked.  The reference's call order is kept: primitive-index order = the order its descent
    // calls the media (and draws).  A resumed lane (time-sliced traversal) evaluated its media when its segment began; they ride in its parked hit state.
    if (!(SLICED && bvh && *mid)) {
        MediumChord chord = medium_chord_begin(P);
        for (int k = 0; k < sc.n_media; ++k) ext_medium_test(sc, sc.media_idx[k], P, tmin, tmax, H, chord, COUNT ? cnt : nullptr, nullptr, k);
        RTMI_PH(PH_MEDIA)
    }
    if (bvh) {
        if (SLICED) {
            const bool done = scan_bvh_ext<true, COUNT>(sc, stack, P, a, tmin, H, stack + susp_off, *mid, min_lanes, cnt);
            *mid = !done;
            if (!done) return;
        } else scan_bvh_ext<false, COUNT>(sc, stack, P, a, tmin, H, nullptr, false, 0, cnt);
    } else if (sc.small_scan) scan_small_ext(sc, P, tmin, H); // (wave-uniform)
    else scan_all_cull_ext(sc, P, a, tmin, H);
    RTMI_PH(PH_BVH_POST)
    best_i = ext_winner(H);
    if (best_i >= 0) best_t = H.t;
}
template <bool SLICED = false, bool COUNT = false, int MSEQ = 0>
__device__ inline void intersect_ext(SceneRef, int *, bool, Path<float> &, bool, float, float tmax, float &best_t, int &best_i, bool * = nullptr, int = 0, unsigned * = nullptr, int = 0) { best_t = tmax; best_i = -1; }

// NOGRID: the plain (not time-sliced) RENDER kernel -- never launched on a scene with an entry grid, so it carries no code for one (the probe
// kernels, also not time-sliced, do: they walk a long segment's pieces in a loop of their own)
template <typename R, bool MULTI, int VARIANT, bool EXT = false, bool COUNT = false, bool SLICED = false, int MSEQ = 0, bool NOGRID = false>
__device__ inline void intersect_world(SceneRef sc, Prim4<R> *lds, int prims_per_tile, int n_ptiles, Path<R> &P,
                                       bool active, R tmin, R tmax, R &best_t, int &best_i, unsigned *cnt = nullptr, bool *mid = nullptr, int min_lanes = 0,
                                       int susp_off = RTMI_BVH_STACK * RTMI_BVH_STRIDE) {
    if (EXT) { intersect_ext<SLICED, COUNT, MSEQ>(sc, reinterpret_cast<int *>(lds), VARIANT == SCAN_BVH, P, active, tmin, tmax, best_t, best_i, mid, min_lanes, cnt, susp_off); return; }
    best_t = tmax;
    best_i = -1;
    const R a = dot3(P.dx, P.dy, P.dz, P.dx, P.dy, P.dz);
    if (VARIANT == SCAN_BVH) { // RTMI_ACCEL_BVH; `lds` is the traversal stack, followed by the suspended lanes' state when the traversal is time-sliced
        int *stack = reinterpret_cast<int *>(lds);
        if (active) {
            if (SLICED) {
                const bool done = scan_bvh<R, COUNT, true>(sc, stack, P, a, tmin, best_t, best_i, [&]() { scan_cull_dispatch(sc, P, a, tmin, best_t, best_i); }, cnt,
                                                           stack + susp_off, *mid, min_lanes);
                *mid = !done;
            } else scan_bvh<R, COUNT, false, !NOGRID>(sc, stack, P, a, tmin, best_t, best_i, [&]() { scan_cull_dispatch(sc, P, a, tmin, best_t, best_i); }, cnt);
        }
        return;
    }
    if (VARIANT == SCAN_SGPR_CULL) { // all primitives, original order; returns the original index
        if (active) scan_cull_dispatch(sc, P, a, tmin, best_t, best_i);
        return;
    }
    if (VARIANT == SCAN_SGPR) {
        if (active) scan_static_pipe<R, false>(ScalarPrims<R>(stat4_of<R>(sc)), sc.n_static, 0, P, a, tmin, best_t, best_i);
    } else if (!MULTI) {
        if (active) {
            if (VARIANT == SCAN_LDS_PIPE) scan_static_pipe<R, true>(LdsPrims<R>{lds}, sc.n_static, 0, P, a, tmin, best_t, best_i);
            else scan_static<R>(lds, sc.n_static, 0, P, a, tmin, best_t, best_i);
        }
    } else {
        for (int tile = 0; tile < n_ptiles; ++tile) {
            const int first = tile * prims_per_tile;
            const int count = min(prims_per_tile, sc.n_static - first);
            __syncthreads();
            stage_prims<R>(sc, lds, first, count);
            __syncthreads();
            if (active) {
                if (VARIANT == SCAN_LDS_PIPE) scan_static_pipe<R, true>(LdsPrims<R>{lds}, count, first, P, a, tmin, best_t, best_i);
                else scan_static<R>(lds, count, first, P, a, tmin, best_t, best_i);
            }
        }
    }
    // static spheres were scanned in their own (original relative) order; convert to the original index, then the
    // moving spheres, with ties resolved by original index (first in Hitlist order wins, hitable.clj:20)
    if (active) {
        if (best_i >= 0) best_i = sc.stat_orig[best_i];
        if (sc.n_moving > 0) {
            int best_orig = best_i >= 0 ? best_i : 0x7fffffff, scan_i = -1;
            scan_moving<R>(sc, P, a, tmin, best_t, scan_i, best_orig);
            if (scan_i >= 0) best_i = best_orig;
        }
    }
}

// core.clj:43-51: jittered (u, v) for sample s of pixel (i, j), then the camera ray.
template <typename R> __device__ inline void start_sample(SceneRef sc, const TraceParams &tp, int i, int j, int s, Path<R> &P) {
    seed_stream(P, sample_key(tp.seed, (u64)j * (u64)tp.nx + (u64)i, (u64)s), 0u);
    const R u = ((R)(float)i + next_uniform(P)) / (R)tp.nx;
    const R v = ((R)(float)j + next_uniform(P)) / (R)tp.ny;
    get_ray<R>(sc, u, v, P);
    P.ar = P.ag = P.ab = R(1);
    P.depth = tp.depth;
}

// SLICE: time-sliced BVH traversal (section 5.1b of DESIGN.md); the plain instantiation is kept for scenes whose tree is too small to gain
// LST: the sphere (non-EXT) time-sliced BVH kernel with the camera-ray stash in LDS and stack columns sized by the scene's tree (launch site: when it fits)
template <typename R, bool MULTI, int VARIANT, bool EXT = false, bool COUNT = false, bool SLICE = true, int MSEQ = 0, bool LST = false>
// (the Hitlist-with-media (MSEQ) and the counting (COUNT) instantiations of the mixed-kind kernels need more than the 128 VGPRs of four waves per SIMD: rather
// than spill, they are compiled for three -- rare worlds and a diagnostic; the kernels the benchmarks run keep four)
__global__ void __launch_bounds__(kTraceBlock, EXT ? ((MSEQ != 0 || COUNT) ? 3 : RTMI_EXT_MIN_WAVES) : RTMI_MIN_WAVES) trace_kernel(ScenePtr scp, TraceParams tp) {
    SceneRef sc = *scp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Prim4<R> *lds = reinterpret_cast<Prim4<R> *>(smem);
    const int lane = threadIdx.x & 63;
    if (!MULTI && VARIANT < SCAN_SGPR) stage_prims<R>(sc, lds, 0, sc.n_static);
    __syncthreads();

    // Work distribution: one queue for the whole launch.  A work item is one sample of one pixel; item m belongs to chunk
    // m / 64 = (local tile m / (64 s_count), sample s_begin + (m / 64) % s_count), pixel m % 64 of that 8x8 tile.  Each WAVE
    // claims kQueueBlock consecutive items at a time with one global atomic and hands them to its dead lanes; no wave idles
    // while the queue holds work, whatever the other waves' paths do.  (Which wave renders an item never affects the result:
    // the RNG stream is a function of (seed, pixel, sample) only.)
    const unsigned total_items = tp.total_items;
    unsigned w_cur = 0, w_end = 0; // this wave's claimed range [w_cur, w_end): wave-uniform
    Path<R> P;
    P.ox = P.oy = P.oz = P.dx = P.dy = P.dz = P.time = R(0);
    P.ar = P.ag = P.ab = R(0);
    seed_stream(P, 0ull, 0u); P.depth = 0;
    bool alive = false;
    bool exhausted = (total_items == 0);
    unsigned out_item = 0; // work item of the lane's path: its colour goes to samples[out_item]
    constexpr bool LSTASH = (EXT && RTMI_STASH && RTMI_EXT_LDS_STASH && !RTMI_EXT_NO_STASH) || LST;
    constexpr bool STASH = RTMI_STASH && !(EXT && RTMI_EXT_NO_STASH) && !LSTASH;
    // LSTASH: entry e of wave w = words lst[k * kTraceBlock + e], k = 0..5 direction, 6..7 time, 8..9 stream state, 10 work item (-1: none), 11..16 origin
    // (only written / read when the camera's rays do not all start at one point).  Written and read by the same wave only.
    int *const lst = reinterpret_cast<int *>(smem) + tp.stash_off + (threadIdx.x & ~63);
    R st_ox = R(0), st_oy = R(0), st_oz = R(0), st_dx = R(0), st_dy = R(0), st_dz = R(0), st_time = R(0); // the stash: one generated camera ray per lane
    u64 st_rs = 0;
    unsigned st_item = 0xffffffffu;
    unsigned s_head = 64u; // wave-uniform: entries [s_head, 64) are unclaimed
    unsigned nrays = 0;
    unsigned ntrav[2] = {0u, 0u};
    const R tmin = R(0.001), tmax = Real<R>::tmax();
    constexpr bool SLICED = SLICE && VARIANT == SCAN_BVH && !MULTI;
    bool mid = false; // this lane's segment is suspended inside the tree (SLICED)

    RTMI_STAMP_DECL
    for (;;) {
        RTMI_PH(PH_LOOP) // loop overhead / tail
        // ---- refill dead lanes ------------------------------------------------------------------------------------------
        if (LSTASH) {
        for (;;) { // wave-uniform control flow; as the register stash below with the entries in LDS
            const u64 dead = __ballot(!alive);
            if (dead == 0) break;
            if (s_head == 64u) {
                if (exhausted) break;
                if (w_cur == w_end) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(tp.queue, tp.qblock);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= total_items) { exhausted = true; break; }
                    w_cur = base;
                    w_end = min(base + tp.qblock, total_items);
                }
                const unsigned m = w_cur + (unsigned)lane;
                w_cur += 64u;
                const unsigned chunk = m >> 6;
                const int l = (int)(m & 63u);
                const int tile_local = (int)(chunk / (unsigned)tp.s_count);
                const int s = tp.s_begin + (int)(chunk - (unsigned)tile_local * (unsigned)tp.s_count);
                const int gtile = tp.tile_ids[tile_local];
                const int x = (gtile % tp.tiles_x) * RTMI_TILE + (l & 7);
                const int y = (gtile / tp.tiles_x) * RTMI_TILE + (l >> 3);
                unsigned item = 0xffffffffu; // (an item outside the image / region: an empty entry; work items are 32-bit UNSIGNED -- C4 has 2.1e9 per pass)
                if (x >= tp.rx0 && x < tp.rx1 && y >= tp.ry0 && y < tp.ry1) {
                    Path<R> Q;
                    start_sample<R>(sc, tp, x, tp.ny - 1 - y, s, Q); // j = ny-1-y (core.clj:105)
                    int w0, w1;
                    best_to_words<R>(Q.dx, w0, w1); lst[lane] = w0; lst[kTraceBlock + lane] = w1;
                    best_to_words<R>(Q.dy, w0, w1); lst[2 * kTraceBlock + lane] = w0; lst[3 * kTraceBlock + lane] = w1;
                    best_to_words<R>(Q.dz, w0, w1); lst[4 * kTraceBlock + lane] = w0; lst[5 * kTraceBlock + lane] = w1;
                    best_to_words<R>(Q.time, w0, w1); lst[6 * kTraceBlock + lane] = w0; lst[7 * kTraceBlock + lane] = w1;
                    lst[8 * kTraceBlock + lane] = (int)(unsigned)Q.rs; lst[9 * kTraceBlock + lane] = (int)(unsigned)(Q.rs >> 32);
                    if (!sc.cam_fixed_origin) {
                        best_to_words<R>(Q.ox, w0, w1); lst[11 * kTraceBlock + lane] = w0; lst[12 * kTraceBlock + lane] = w1;
                        best_to_words<R>(Q.oy, w0, w1); lst[13 * kTraceBlock + lane] = w0; lst[14 * kTraceBlock + lane] = w1;
                        best_to_words<R>(Q.oz, w0, w1); lst[15 * kTraceBlock + lane] = w0; lst[16 * kTraceBlock + lane] = w1;
                    }
                    item = m;
                    RTMI_PH(PH_REFILL_GEN)
                }
                lst[10 * kTraceBlock + lane] = (int)item;
                s_head = 0u;
            }
            const unsigned avail = 64u - s_head, nd = (unsigned)__popcll(dead);
            const unsigned rank = (unsigned)__popcll(dead & ((1ull << lane) - 1ull));
            if (!alive && rank < avail) {
                const int e = (int)(s_head + rank);
                const unsigned item = (unsigned)lst[10 * kTraceBlock + e];
                if (item != 0xffffffffu) {
                    P.dx = best_from_words<R>(lst[e], lst[kTraceBlock + e]); P.dy = best_from_words<R>(lst[2 * kTraceBlock + e], lst[3 * kTraceBlock + e]);
                    P.dz = best_from_words<R>(lst[4 * kTraceBlock + e], lst[5 * kTraceBlock + e]); P.time = best_from_words<R>(lst[6 * kTraceBlock + e], lst[7 * kTraceBlock + e]);
                    P.rs = (u64)(unsigned)lst[8 * kTraceBlock + e] | ((u64)(unsigned)lst[9 * kTraceBlock + e] << 32);
                    if (sc.cam_fixed_origin) { P.ox = (R)sc.cam[0]; P.oy = (R)sc.cam[1]; P.oz = (R)sc.cam[2]; }
                    else {
                        P.ox = best_from_words<R>(lst[11 * kTraceBlock + e], lst[12 * kTraceBlock + e]); P.oy = best_from_words<R>(lst[13 * kTraceBlock + e], lst[14 * kTraceBlock + e]);
                        P.oz = best_from_words<R>(lst[15 * kTraceBlock + e], lst[16 * kTraceBlock + e]);
                    }
                    P.ar = P.ag = P.ab = R(1);
                    P.depth = tp.depth;
                    out_item = item;
                    alive = true;
                }
            }
            s_head += min(nd, avail);
        }
        } else if (STASH) {
        // Camera rays are generated 64 at a time by the WHOLE wave (key, jitter, lens disk loop, get-ray: start_sample at full
        // width) into a register stash, one entry per lane; dead lanes then pull entries across lanes (ds_bpermute): entry
        // s_head + (rank among the dead lanes).  Generating per trip for the dead lanes only ran start_sample at ~40 % width
        // every trip; now it runs at full width every ~2.5 trips.  Which lane traces an item never affects the result.
        for (;;) { // wave-uniform control flow
            const u64 dead = __ballot(!alive);
            if (dead == 0) break;
            if (s_head == 64u) {
                if (exhausted) break;
                if (w_cur == w_end) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(tp.queue, tp.qblock);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= total_items) { exhausted = true; break; }
                    w_cur = base;
                    w_end = min(base + tp.qblock, total_items); // total_items and qblock are multiples of 64
                }
                const unsigned m = w_cur + (unsigned)lane;
                w_cur += 64u;
                const unsigned chunk = m >> 6;
                const int l = (int)(m & 63u);
                const int tile_local = (int)(chunk / (unsigned)tp.s_count);
                const int s = tp.s_begin + (int)(chunk - (unsigned)tile_local * (unsigned)tp.s_count);
                const int gtile = tp.tile_ids[tile_local];
                const int x = (gtile % tp.tiles_x) * RTMI_TILE + (l & 7);
                const int y = (gtile / tp.tiles_x) * RTMI_TILE + (l >> 3);
                st_item = 0xffffffffu; // an item outside the image / region: an empty entry
                if (x >= tp.rx0 && x < tp.rx1 && y >= tp.ry0 && y < tp.ry1) {
                    Path<R> Q;
                    start_sample<R>(sc, tp, x, tp.ny - 1 - y, s, Q); // j = ny-1-y (core.clj:105)
                    st_ox = Q.ox; st_oy = Q.oy; st_oz = Q.oz; st_dx = Q.dx; st_dy = Q.dy; st_dz = Q.dz; st_time = Q.time; st_rs = Q.rs;
                    st_item = m;
                    RTMI_PH(PH_REFILL_GEN)
                }
                s_head = 0u;
            }
            const unsigned avail = 64u - s_head, nd = (unsigned)__popcll(dead);
            const unsigned rank = (unsigned)__popcll(dead & ((1ull << lane) - 1ull));
            const bool take = !alive && rank < avail;
            const int src = take ? (int)(s_head + rank) : lane;
            R f_ox, f_oy, f_oz;
            if (sc.cam_fixed_origin) { f_ox = (R)sc.cam[0]; f_oy = (R)sc.cam[1]; f_oz = (R)sc.cam[2]; } // wave-uniform: 6 cross-lane moves fewer per deal
            else { f_ox = __shfl(st_ox, src); f_oy = __shfl(st_oy, src); f_oz = __shfl(st_oz, src); }
            const R f_dx = __shfl(st_dx, src), f_dy = __shfl(st_dy, src), f_dz = __shfl(st_dz, src), f_time = __shfl(st_time, src);
            const u64 f_rs = __shfl(st_rs, src);
            const unsigned f_item = __shfl(st_item, src);
            if (take && f_item != 0xffffffffu) {
                P.ox = f_ox; P.oy = f_oy; P.oz = f_oz; P.dx = f_dx; P.dy = f_dy; P.dz = f_dz; P.time = f_time; P.rs = f_rs;
                P.ar = P.ag = P.ab = R(1);
                P.depth = tp.depth;
                out_item = f_item;
                alive = true;
            }
            s_head += min(nd, avail);
        }
        } else
        while (!exhausted) { // every lane of the wave takes part: the loop conditions are wave-uniform
            const u64 dead = __ballot(!alive);
            if (dead == 0) break;
            if (w_cur == w_end) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(tp.queue, kQueueBlock);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= total_items) { exhausted = true; break; }
                w_cur = base;
                w_end = min(base + kQueueBlock, total_items);
            }
            const unsigned avail = w_end - w_cur;
            const unsigned rank = (unsigned)__popcll(dead & ((1ull << lane) - 1ull));
            if (!alive && rank < avail) {
                const unsigned m = w_cur + rank;
                const unsigned chunk = m >> 6;
                const int l = (int)(m & 63u);
                const int tile_local = (int)(chunk / (unsigned)tp.s_count);
                const int s = tp.s_begin + (int)(chunk - (unsigned)tile_local * (unsigned)tp.s_count);
                const int gtile = tp.tile_ids[tile_local];
                const int x = (gtile % tp.tiles_x) * RTMI_TILE + (l & 7);
                const int y = (gtile / tp.tiles_x) * RTMI_TILE + (l >> 3);
                if (x >= tp.rx0 && x < tp.rx1 && y >= tp.ry0 && y < tp.ry1) {
                    start_sample<R>(sc, tp, x, tp.ny - 1 - y, s, P); // j = ny-1-y (core.clj:105)
                    out_item = m;
                    alive = true;
                }
            }
            w_cur += min((unsigned)__popcll(dead), avail);
        }
        if (MULTI) { if (!__syncthreads_or(alive ? 1 : 0)) break; }
        else { if (!__any(alive ? 1 : 0)) break; }
        RTMI_PH(PH_REFILL_DEAL) // refill: claims, dealing stash entries to dead lanes

        // ---- one iteration of `color` for every live lane ---------------------------------------------
        // SLICED (BVH kernels): the traversal hands the wave back as soon as fewer than tp.suspend_lanes lanes are still in the tree
        // (a few rays of a wave visit ten times the nodes the others do: 40 % of the node-visit trips served < 8 lanes); those lanes
        // are `mid` segment -- they sit out the shading below and resume where they stopped in the next trip, next to the new
        // segments of the others.  Once the queue is empty nothing is gained by handing back early (suspend_lanes 0).
        R best_t; int best_i;
        intersect_world<R, MULTI, VARIANT, EXT, COUNT, SLICED, MSEQ, !SLICED>(sc, lds, tp.prims_per_tile, tp.n_ptiles, P, alive, tmin, tmax, best_t, best_i, ntrav,
                                                                     &mid, exhausted ? 0 : tp.suspend_lanes, (EXT || LST) ? tp.susp_off : RTMI_BVH_STACK * RTMI_BVH_STRIDE);
        RTMI_PH(PH_BVH_POST) // intersection: what the phases inside did not book (suspend bookkeeping, call overhead)
        if (!SLICED || __any(alive && !mid)) { // a trip in which no lane finished its segment has nothing to shade
        if (alive) {
            if (!mid) ++nrays;
            R emit[3];
            const bool scat = shade_segment<R, EXT>(sc, P, best_t, best_i, nullptr, emit, mid); // a `mid` lane is a passenger: it changes nothing
            if (!mid && !scat) {
                R *out = reinterpret_cast<R *>(tp.samples) + (size_t)out_item * 3;
                out[0] = emit[0]; out[1] = emit[1]; out[2] = emit[2];
                alive = false;
                RTMI_PH(PH_STORE)
            }
        }
        }
        RTMI_PH(PH_STORE) // shading: what its phases did not book
    }
    RTMI_STAMP_FLUSH(tp.counters)
    // total-rays: wave reduction, one atomic per wave
    unsigned n = nrays;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if (lane == 0 && n) atomicAdd(tp.counters, (u64)n);
    if (COUNT) {
        u64 a = ntrav[0], b = ntrav[1];
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
        if (lane == 0) { atomicAdd(tp.trav, a); atomicAdd(tp.trav + 1, b); }
    }
}

// core.clj:52-53: (reduce mat/add) over the samples IN ORDER, then (mul (/ 1.0 nr)) on the last pass.
template <typename R>
__global__ void __launch_bounds__(kBlock) reduce_kernel(const R *__restrict__ samples, R *__restrict__ accum, double *__restrict__ tiles_linear,
                                                        const int *__restrict__ tile_ids, int tiles_x, int nx, int ny, int n_local_tiles,
                                                        int s_begin, int s_count, int ns, u64 *counters, u64 n_valid_pixels,
                                                        int rx0, int ry0, int rx1, int ry1) {
    // A (tile, sample) row is 64 pixels x 3 channels = 192 consecutive values; the sum over the samples is element-wise, so lane l of the
    // tile's wave owns elements l, l + 64, l + 128 of the row (pixel j / 3, channel j % 3): every load of the wave is one contiguous
    // 64-element run (a thread per PIXEL read its 3 values at a 24-byte stride, three passes over the same cache lines).  Per element
    // the additions are still s = 0, 1, 2, ... in order.
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0 && counters && s_begin + s_count >= ns) counters[1] = n_valid_pixels; // metrics total-pixels, core.clj:47
    if (gid >= (long long)n_local_tiles * 64) return;
    const int tile_local = (int)(gid >> 6), l = (int)(gid & 63);
    const int gtile = tile_ids[tile_local];
    const int tx = (gtile % tiles_x) * RTMI_TILE, ty = (gtile / tiles_x) * RTMI_TILE;
    R acc[3];
    bool valid[3];
    const size_t tile_base = (size_t)tile_local * 192;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = (l + 64 * k) / 3; // pixel of the tile this element belongs to
        const int x = tx + (p & 7), y = ty + (p >> 3);
        valid[k] = x >= rx0 && x < rx1 && y >= ry0 && y < ry1;
        acc[k] = (valid[k] && s_begin > 0) ? accum[tile_base + l + 64 * k] : R(0);
    }
    const R *row = samples + (size_t)tile_local * s_count * 192 + l;
    for (int s = 0; s < s_count; ++s, row += 192) {
        const R v0 = row[0], v1 = row[64], v2 = row[128];
        if (s_begin + s == 0) { acc[0] = v0; acc[1] = v1; acc[2] = v2; } // the fold starts FROM the first sample (not 0 + first)
        else { acc[0] = acc[0] + v0; acc[1] = acc[1] + v1; acc[2] = acc[2] + v2; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t o = tile_base + l + 64 * k;
        if (s_begin + s_count < ns) accum[o] = valid[k] ? acc[k] : R(0);
        else tiles_linear[o] = valid[k] ? (double)(acc[k] * (R(1.0) / (R)ns)) : 0.0;
    }
}

// core.clj:54-56 + the y-flipped store of core.clj:105-106 (tiles already hold output rows).
// gathered[r][k][64][3]: rank r's k-th tile is global tile r + k*world.
template <typename R>
__global__ void __launch_bounds__(kBlock) assemble_kernel(const double *__restrict__ gathered, int world, size_t rank_stride, int tiles_x,
                                                          int nx, int ny, double *__restrict__ out_linear, unsigned char *__restrict__ out_rgb8) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)nx * ny) return;
    const int x = (int)(gid % nx), y = (int)(gid / nx);
    const int gtile = (y / RTMI_TILE) * tiles_x + (x / RTMI_TILE);
    const int r = gtile % world, k = gtile / world;
    const int l = (y % RTMI_TILE) * RTMI_TILE + (x % RTMI_TILE);
    const double *p = gathered + (size_t)r * rank_stride + ((size_t)k * 64 + l) * 3; // rank_stride: doubles per rank record
    for (int c = 0; c < 3; ++c) {
        const double m = p[c];
        if (out_linear) out_linear[gid * 3 + c] = m;
        if (out_rgb8) {
            const R q = Real<R>::sqrt_((R)m) * R(255.99);
            // (int (min 255.99 q)): clojure.core/min propagates NaN and (int NaN) = 0
            unsigned char o = 0;
            if (q == q) { const R mq = q < R(255.99) ? q : R(255.99); o = (unsigned char)(int)mq; }
            out_rgb8[gid * 3 + c] = o;
        }
    }
}

// The same for a rectangular window of tiles [tx0, tx0+wtx) x [ty0, ...) rendered for the output region [x0, x0+w) x [y0, y0+h):
// tiles[k][64][3], k row-major over the window; writes the dense w x h region.
template <typename R>
__global__ void __launch_bounds__(kBlock) assemble_region_kernel(const double *__restrict__ tiles, int tx0, int ty0, int wtx, int x0, int y0, int w, int h,
                                                                 double *__restrict__ out_linear, unsigned char *__restrict__ out_rgb8) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)w * h) return;
    const int x = x0 + (int)(gid % w), y = y0 + (int)(gid / w);
    const int k = (y / RTMI_TILE - ty0) * wtx + (x / RTMI_TILE - tx0);
    const int l = (y % RTMI_TILE) * RTMI_TILE + (x % RTMI_TILE);
    const double *p = tiles + ((size_t)k * 64 + l) * 3;
    for (int c = 0; c < 3; ++c) {
        const double m = p[c];
        if (out_linear) out_linear[gid * 3 + c] = m;
        if (out_rgb8) {
            const R q = Real<R>::sqrt_((R)m) * R(255.99);
            unsigned char o = 0;
            if (q == q) { const R mq = q < R(255.99) ? q : R(255.99); o = (unsigned char)(int)mq; }
            out_rgb8[gid * 3 + c] = o;
        }
    }
}

// multi-device render: every rank's record ends with its two metrics counters; out = their sums
__global__ void sum_counters_kernel(const u64 *gathered, int world, size_t rank_stride_u64, size_t off_u64, u64 *out) {
    if (threadIdx.x < 2) {
        u64 acc = 0;
        for (int r = 0; r < world; ++r) acc += gathered[(size_t)r * rank_stride_u64 + off_u64 + threadIdx.x];
        out[threadIdx.x] = acc;
    }
}

// ---- probe kernels (one protocol call per thread; same device functions as trace_kernel) -------------
template <typename R> __device__ inline void load_ray(const double *q, Path<R> &P) {
    P.ox = (R)q[0]; P.oy = (R)q[1]; P.oz = (R)q[2]; P.dx = (R)q[3]; P.dy = (R)q[4]; P.dz = (R)q[5]; P.time = (R)q[6];
    P.ar = P.ag = P.ab = R(1); seed_stream(P, 0ull, 0u); P.depth = 0;
}

template <typename R, int VARIANT, bool EXT = false, int MSEQ = 0>
__global__ void __launch_bounds__(kBlock) probe_hit_kernel(ScenePtr scp, int prims_per_tile, int n_ptiles, int n, const double *rays, double tmin, double tmax, double *out) {
    SceneRef sc = *scp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Prim4<R> *lds = reinterpret_cast<Prim4<R> *>(smem);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = k < n;
    Path<R> P;
    load_ray<R>(rays + (size_t)(active ? k : 0) * 7, P);
    R best_t; int best_i;
    intersect_world<R, true, VARIANT, EXT, false, false, MSEQ>(sc, lds, prims_per_tile, n_ptiles, P, active, (R)tmin, (R)tmax, best_t, best_i);
    if (!active) return;
    double *o = out + (size_t)k * 11;
    for (int c = 0; c < 11; ++c) o[c] = 0.0;
    if (best_i < 0) return;
    HitRec<R> h;
    resolve_any<R, EXT>(sc, P, best_t, best_i, h);
    o[0] = 1.0; o[1] = h.orig; o[2] = h.t; o[3] = h.px; o[4] = h.py; o[5] = h.pz;
    o[6] = h.nx; o[7] = h.ny; o[8] = h.nz; o[9] = h.u; o[10] = h.v;
}

template <typename R, int VARIANT, bool EXT = false, int MSEQ = 0>
__global__ void __launch_bounds__(kBlock) probe_paths_kernel(ScenePtr scp, int prims_per_tile, int n_ptiles, int n, const double *rays, const u64 *keys, u64 ctr0,
                                                             int depth, double *out_rgb, u64 *out_nseg, double *log, int max_seg, int *out_nlog) {
    SceneRef sc = *scp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Prim4<R> *lds = reinterpret_cast<Prim4<R> *>(smem);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    bool alive = k < n;
    Path<R> P;
    load_ray<R>(rays + (size_t)(alive ? k : 0) * 7, P);
    seed_stream(P, alive ? keys[k] : 0ull, (unsigned)ctr0); P.depth = depth;
    SegLog lg = {log ? log + (size_t)(alive ? k : 0) * max_seg * RTMI_SEG_REC : nullptr, max_seg, 0};
    u64 nseg = 0;
    const R tmin = R(0.001), tmax = Real<R>::tmax();
    R rgb[3] = {R(0), R(0), R(0)};
    while (__syncthreads_or(alive ? 1 : 0)) {
        R best_t; int best_i;
        intersect_world<R, true, VARIANT, EXT, false, false, MSEQ>(sc, lds, prims_per_tile, n_ptiles, P, alive, tmin, tmax, best_t, best_i);
        if (alive) {
            ++nseg;
            R emit[3];
            alive = shade_segment<R, EXT>(sc, P, best_t, best_i, log ? &lg : nullptr, emit);
            if (!alive) { rgb[0] = emit[0]; rgb[1] = emit[1]; rgb[2] = emit[2]; }
        }
    }
    if (k < n) {
        out_rgb[3 * k] = rgb[0]; out_rgb[3 * k + 1] = rgb[1]; out_rgb[3 * k + 2] = rgb[2];
        if (out_nseg) out_nseg[k] = nseg;
        if (out_nlog) out_nlog[k] = lg.n;
    }
}

template <typename R> __global__ void probe_camera_kernel(ScenePtr scp, int n, const double *uv, const u64 *keys, double *out) {
    SceneRef sc = *scp;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Path<R> P;
    seed_stream(P, keys[k], 0u); P.depth = 0;
    get_ray<R>(sc, (R)uv[2 * k], (R)uv[2 * k + 1], P);
    double *o = out + (size_t)k * 8;
    o[0] = P.ox; o[1] = P.oy; o[2] = P.oz; o[3] = P.dx; o[4] = P.dy; o[5] = P.dz; o[6] = P.time; o[7] = (double)P.ctr;
}

template <typename R, bool F4 = false> __global__ void probe_texture_kernel(ScenePtr scp, int tex, int n, const double *uvp, double *out) {
    SceneRef sc = *scp;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double *q = uvp + (size_t)k * 5;
    R r, g, b;
    tex_sample<R, F4>(sc, tex, (R)q[0], (R)q[1], (R)q[2], (R)q[3], (R)q[4], r, g, b);
    out[3 * k] = r; out[3 * k + 1] = g; out[3 * k + 2] = b;
}

// Shader.scatter (shader.clj) on an explicit hit record {p, normal, u, v}: the same scatter_emit the render kernel runs.
template <typename R, bool F4 = false>
__global__ void probe_scatter_kernel(ScenePtr scp, int mat, int n, const double *rays, const double *hits, const u64 *keys, double *out) {
    SceneRef sc = *scp;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Path<R> P;
    load_ray<R>(rays + (size_t)k * 7, P);
    seed_stream(P, keys[k], 0u); P.depth = 1;
    const double *hq = hits + (size_t)k * 8;
    HitRec<R> h;
    h.t = R(0); h.px = (R)hq[0]; h.py = (R)hq[1]; h.pz = (R)hq[2]; h.nx = (R)hq[3]; h.ny = (R)hq[4]; h.nz = (R)hq[5];
    h.u = (R)hq[6]; h.v = (R)hq[7]; h.orig = -1; h.kind = RTMI_PRIM_SPHERE; h.mat = mat;
    R att[3] = {R(0), R(0), R(0)}, emit[3];
    const bool scat = scatter_emit<R, F4>(sc, P, h, att, emit);
    double *o = out + (size_t)k * 9;
    o[0] = scat ? 1.0 : 0.0;
    o[1] = scat ? P.dx : 0; o[2] = scat ? P.dy : 0; o[3] = scat ? P.dz : 0;
    o[4] = scat ? att[0] : 0; o[5] = scat ? att[1] : 0; o[6] = scat ? att[2] : 0;
    o[7] = scat ? P.time : 0; o[8] = (double)P.ctr;
}

template <typename R> __global__ void probe_rng_kernel(u64 key, u64 d0, int n, u64 *bits, double *real) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const u64 z = draw_bits(key, d0 + (u64)k);
    bits[k] = z;
    real[k] = (double)Real<R>::uniform(z);
}

__global__ void probe_arith_kernel(int n, const double *abc, double *out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double a = abc[3 * k], b = abc[3 * k + 1], c = abc[3 * k + 2];
    out[3 * k] = a / b;
    out[3 * k + 1] = ::sqrt(::fabs(a));
    out[3 * k + 2] = a * b + c; // must stay unfused (-ffp-contract=off)
}

} // namespace

// =====================================================================================================
// host side: C-ABI
// =====================================================================================================
namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(RTMI_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// the multi-device entries switch the calling thread's current device; hosts that track it themselves (PyTorch) get it back
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return RTMI_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (hipMalloc(&p, need) != hipSuccess) { p = nullptr; return fail(RTMI_E_NOMEM, "hipMalloc(%zu bytes) failed", need); }
        bytes = need;
        return RTMI_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

} // namespace

struct rtmi_ctx {
    uint32_t magic = 0x52544d49u;
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    int cus = 0;
    int lds_per_cu = 0;
    size_t hbm = 0;
    std::string arch;
    int blocks_per_cu = 8; // workgroups per CU in the persistent grid (4 resident; the rest start as others drain: shorter tail)
    int64_t workspace_bytes = (int64_t)64 << 30; // sample-buffer budget (HBM is 288 GB; allocated as needed): 1920x1080x256 (12.7 GB of samples) renders in one pass, 3840x2160x512 in two
    int accel = RTMI_ACCEL_BVH; // bit-identical to the flat Hitlist scan and what every reference scene builds (scene.clj:332: make-bvh)
    int scan_variant = SCAN_SGPR_CULL;
    int max_lds_bytes = 64 * 1024 - 64; // static-sphere LDS tile budget per workgroup
    // workspace
    DevBuf samples, accum, tiles, tile_ids, counters, scratch_lin;
    DevBuf multi; // rtmi_render_multi*: this replica's record (tiles + counters); on replica 0 the gathered records of all replicas
    hipEvent_t ev_done = nullptr, ev_g0 = nullptr, ev_g1 = nullptr; // multi-device: render finished / gather interval on replica 0
    bool have_gather = false;
    int last_gather_path = RTMI_GATHER_NONE; // how the last rtmi_render_multi* on this context (as replica 0) gathered
    // copy-branch gather: replica 0's stream copies OUT of this replica's record; the event (created on replica 0's device, recorded on
    // its stream after the copy) is what this replica's stream waits for before it renders into the record again
    hipEvent_t ev_consumed = nullptr;
    int ev_consumed_device = -1;
    bool consume_pending = false;
    int fail_next_render = 0;  // option "test_fail_next_render" (test hook): the next render on this context fails before it launches anything
    int fail_allocs = 0;       // option "test_fail_allocs" (test hook): the next n sample-buffer allocations fail as if HBM were exhausted
    int last_accel = -1;       // RTMI_ACCEL_* the most recent render ran (rtmi_last_accel): option "flat_below" can answer a request for the tree with the scan
    int last_passes = 0;       // sample passes of the most recent render (rtmi_last_passes)
    int last_grid = 0; // workgroups of the last trace launch (diagnostics)
    std::map<std::pair<const void *, size_t>, int> occupancy; // resident workgroups per CU by (kernel, dynamic LDS bytes)
    std::vector<int> tile_ids_host;
    int tile_key[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    std::vector<hipEvent_t> events_r; // RTMI_FLAG_TIMING: after the reduction that follows launch k
    double last_reduce_ms = 0.0;      // of the window rtmi_last_trace_ms closed last
    int last_reduce_launches = 0;
    int count_traversal = 0;      // option "count_traversal": run the COUNT instantiation of the BVH kernels
    int flat_below = 24;          // option "flat_below": mixed-kind scenes with fewer primitives answer accel = BVH with the flat scan (same image; 3 - 10 % faster there)
    bool suspend_lanes_set = false; // the option was set by the host (else mixed-kind trees take their own default, see the launch)
    int suspend_lanes = 8;        // option "suspend_lanes": threshold of the time-sliced BVH traversal (0 = plain while-while loop); 6 .. 12 within 0.4 % (C3 69.3 / 69.2 / 69.5 ms at 6 / 10 / 12; 18: 70.7, 24: 73.6)
    hipStream_t last_stream = nullptr; // stream of the most recent render (rtmi_last_traversal_counters synchronises on it)
    long long tile_valid_pixels = 0;
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    int events_used = 0;
};

struct rtmi_scene {
    uint32_t magic = 0x52545343u;
    rtmi_ctx *ctx = nullptr;
    DevScene dev{};             // host copy of the descriptor
    ScenePtr d_dev = nullptr;   // the descriptor in HBM (what the kernels read)
    std::vector<void *> allocs;
    size_t device_bytes = 0;  // HBM the scene occupies = what its creation uploads (rtmi_scene_device_bytes)
    int n_prims = 0, n_mats = 0, n_tex = 0;
    int bvh_node_count = 0;   // inner nodes of the device's tree
    int bvh_depth = 0;        // deepest leaf of the device's tree(s)
    bool uses_perlin = false; // a Perlin texture is present: rtmi_scene_set_perlin must have been called before rendering
    int max_image = -1;       // highest ImageMap index: rtmi_scene_set_images must cover it
    bool have_perlin = false;
    std::vector<int> host_kind; // primitive kinds (boundary flag removed), for argument checks
    std::map<int, std::array<double, 5>> media_fast_of; // medium primitive -> {density, c.xyz, r*r} when it and its boundary are one plain sphere without wrappers (DevScene::media_fast)
    // the caller's arrays, copied at creation (the library keeps no host POINTERS): what rtmi_scene_clone replicates
    struct Args {
        std::vector<int32_t> prim_kind, prim_mat, mat_kind, mat_tex, tex_kind, tex_child, prim_flip, prim_xform, xform_kind, perm, media_calls, media_lo, image_wh;
        std::vector<double> prim_geom, mat_param, tex_param, cam, xform_param, perlin_vec;
        std::vector<uint8_t> image_rgb;
        int cam_kind = 0;
        int media_mode = 0;
        bool has_media_calls = false;
    } args;
};

namespace {

bool ctx_ok(rtmi_ctx *c) { return c && c->magic == 0x52544d49u; }
bool scene_ok(rtmi_scene *s) { return s && s->magic == 0x52545343u && ctx_ok(s->ctx); }

template <typename T> int upload(rtmi_scene *s, const std::vector<T> &v, const T **out) {
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(RTMI_E_NOMEM, "hipMalloc(%zu) failed", bytes);
    s->allocs.push_back(p);
    s->device_bytes += bytes;
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T *>(p);
    return RTMI_OK;
}

// LDS tiling of the static spheres for a given precision
void lds_plan(const rtmi_ctx *c, int n_static, size_t real_bytes, int *prims_per_tile, int *n_ptiles, size_t *lds_bytes) {
    const size_t rec = 4 * real_bytes;
    int cap = (int)((size_t)c->max_lds_bytes / rec);
    int ppt = std::max(1, std::min(std::max(n_static, 1), cap));
    *prims_per_tile = ppt;
    *n_ptiles = std::max(1, (n_static + ppt - 1) / ppt);
    *lds_bytes = (size_t)ppt * rec + 16;
}

int tiles_x_of(int nx) { return (nx + RTMI_TILE - 1) / RTMI_TILE; }
int tiles_y_of(int ny) { return (ny + RTMI_TILE - 1) / RTMI_TILE; }

// Local tile list of a render: global tiles first, first+stride, ... that intersect the output region rg = {x0, y0, x1, y1}.
int ensure_tile_ids(rtmi_ctx *c, int nx, int ny, int first, int stride, const int *rg, hipStream_t st, int *n_local) {
    const int ntiles = tiles_x_of(nx) * tiles_y_of(ny);
    const int key[8] = {nx, ny, first, stride, rg[0], rg[1], rg[2], rg[3]};
    if (!std::memcmp(key, c->tile_key, sizeof key)) { *n_local = (int)c->tile_ids_host.size(); return RTMI_OK; }
    const int tx_n = tiles_x_of(nx);
    const bool whole = rg[0] <= 0 && rg[1] <= 0 && rg[2] >= nx && rg[3] >= ny;
    c->tile_ids_host.clear();
    long long valid = 0;
    for (int g = first; g < ntiles; g += stride) {
        const int px0 = (g % tx_n) * RTMI_TILE, py0 = (g / tx_n) * RTMI_TILE;
        const int ax0 = std::max(px0, rg[0]), ay0 = std::max(py0, rg[1]);
        const int ax1 = std::min(std::min(px0 + RTMI_TILE, nx), rg[2]), ay1 = std::min(std::min(py0 + RTMI_TILE, ny), rg[3]);
        if (ax1 <= ax0 || ay1 <= ay0) { if (whole) c->tile_ids_host.push_back(g); continue; } // (cannot happen for the whole frame)
        c->tile_ids_host.push_back(g);
        valid += (long long)(ax1 - ax0) * (ay1 - ay0);
    }
    const int nl = (int)c->tile_ids_host.size();
    *n_local = nl;
    c->tile_valid_pixels = valid;
    int rc = c->tile_ids.ensure((size_t)std::max(nl, 1) * sizeof(int));
    if (rc) return rc;
    if (nl) HIP_TRY(hipMemcpyAsync(c->tile_ids.p, c->tile_ids_host.data(), (size_t)nl * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st)); // tile_ids_host may be rewritten by the next call
    std::memcpy(c->tile_key, key, sizeof key);
    return RTMI_OK;
}

int next_event_pair(rtmi_ctx *c, hipEvent_t *a, hipEvent_t *b) {
    if (c->events_used == (int)c->events.size()) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        c->events.emplace_back(e0, e1);
    }
    *a = c->events[(size_t)c->events_used].first;
    *b = c->events[(size_t)c->events_used].second;
    c->events_used++;
    return RTMI_OK;
}

template <typename R>
int render_tiles_impl(rtmi_scene *s, int nx, int ny, int ns, int depth, uint64_t seed, int first, int stride, const int *rg, void *d_tiles_linear,
                      void *d_counters, hipStream_t st) {
    rtmi_ctx *c = s->ctx;
    int n_local = 0;
    const int whole[4] = {0, 0, nx, ny};
    if (!rg) rg = whole;
    int rc = ensure_tile_ids(c, nx, ny, first, stride, rg, st, &n_local);
    if (rc) return rc;
    c->last_stream = st;
    if (c->fail_next_render) { c->fail_next_render = 0; return fail(RTMI_E_DEVICE, "render failed (injected by the test hook test_fail_next_render)"); }
    rc = c->counters.ensure(8 * sizeof(u64)); // [0..1] the metrics when the caller passes no buffer, [2] the work-queue head, [3..4] traversal counters
    if (rc) return rc;
    if (d_counters) HIP_TRY(hipMemsetAsync(d_counters, 0, 2 * sizeof(u64), st));
    HIP_TRY(hipMemsetAsync(reinterpret_cast<u64 *>(c->counters.p) + 3, 0, 2 * sizeof(u64), st));
    if (n_local == 0) return RTMI_OK;
    u64 *cnt = d_counters ? reinterpret_cast<u64 *>(d_counters) : reinterpret_cast<u64 *>(c->counters.p);
    unsigned *queue = reinterpret_cast<unsigned *>(reinterpret_cast<u64 *>(c->counters.p) + 2);

    // sample-buffer passes: samples [s_begin, s_begin+s_count) of every local pixel per pass
    const size_t per_sample = (size_t)n_local * 64 * 3 * sizeof(R);
    int s_per_pass = (int)std::max<int64_t>(1, std::min<int64_t>(ns, c->workspace_bytes / (int64_t)per_sample));
    // the work queue is indexed with 32 bits: items per pass (+ one claim per wave past the end) must stay below 2^32
    while (s_per_pass > 1 && (long long)n_local * s_per_pass * 64 >= 0xf0000000ll) s_per_pass /= 2;
    if ((long long)n_local * s_per_pass * 64 >= 0xf0000000ll) return fail(RTMI_E_ARG, "frame too large for one pass: %d tiles per rank", n_local);
    if (per_sample * (size_t)s_per_pass > c->samples.bytes) {
        // The buffer has to grow.  The budget is what the option allows AND what the device can give: at most kFreeShare of the HBM that
        // is free right now (plus what this context's buffer already holds) -- a host that shares the GPU (PyTorch's caching allocator, a
        // second render slot) gets more passes instead of RTMI_E_NOMEM.  If the allocation still fails, halve the pass and retry.
        constexpr double kFreeShare = 0.8;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double avail = kFreeShare * (double)free_b + (double)c->samples.bytes;
            const int fit = (int)std::max<double>(1.0, std::min<double>((double)s_per_pass, avail / (double)per_sample));
            s_per_pass = std::min(s_per_pass, fit);
        }
        for (;;) {
            if (c->fail_allocs > 0) { c->fail_allocs--; c->samples.release(); rc = fail(RTMI_E_NOMEM, "hipMalloc(%zu bytes) failed (injected by the test hook)", per_sample * (size_t)s_per_pass); }
            else rc = c->samples.ensure(per_sample * (size_t)s_per_pass);
            if (!rc) break;
            (void)hipGetLastError(); // a failed hipMalloc leaves its error sticky
            if (s_per_pass == 1) return rc;
            s_per_pass = (s_per_pass + 1) / 2;
        }
    }
    c->last_passes = (ns + s_per_pass - 1) / s_per_pass;
    if (s_per_pass < ns) {
        rc = c->accum.ensure((size_t)n_local * 64 * 3 * sizeof(R));
        if (rc) return rc;
    }
    int ppt, nptiles;
    size_t lds_bytes;
    lds_plan(c, s->dev.n_static, sizeof(R), &ppt, &nptiles, &lds_bytes);
    const bool multi = nptiles > 1;

    for (int s_begin = 0; s_begin < ns; s_begin += s_per_pass) {
        const int s_count = std::min(s_per_pass, ns - s_begin);
        TraceParams tp;
        tp.nx = nx; tp.ny = ny; tp.depth = depth; tp.seed = seed; tp.tiles_x = tiles_x_of(nx);
        tp.n_local_tiles = n_local; tp.tile_ids = reinterpret_cast<const int *>(c->tile_ids.p);
        tp.s_begin = s_begin; tp.s_count = s_count; tp.samples = c->samples.p; tp.counters = cnt;
        tp.prims_per_tile = (c->scan_variant >= SCAN_SGPR || s->dev.has_ext) ? 0 : ppt; tp.n_ptiles = nptiles;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if ((c->flags & RTMI_FLAG_TIMING) && c->events_used < 8192) {
            rc = next_event_pair(c, &e0, &e1);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(e0, st));
        }
        tp.queue = queue; tp.total_items = (unsigned)((long long)n_local * s_count * 64);
        { // claim size: one global atomic per claim -- 256 items on small launches (a short tail matters more), up to 1024 when a wave has
          // thousands of claims ahead of it (C3: 129 000 items per wave)
            const long long per_wave = (long long)tp.total_items / std::max(1, c->cus * 16);
            // (mixed-kind scenes: 192 -- their launches are short and end in a long die-off of deep paths through the media; make-final 64 / 128 / 192 / 256 / 320 / 384 / 512 items:
            // 18.2 / 17.3 / 17.1 - 17.2 / 17.4 - 17.5 / 17.9 / 18.3 / 19.4 ms, the Cornell box indifferent)
            unsigned qb = s->dev.has_ext ? 192u : kQueueBlock;
            while (qb < 1024u && per_wave >= (long long)qb * 128) qb *= 2;
            if (const char *e = std::getenv("RTMI_QUEUE_BLOCK_RT")) qb = std::max(64, std::atoi(e) / 64 * 64);
            tp.qblock = qb;
        }
        tp.rx0 = std::max(rg[0], 0); tp.ry0 = std::max(rg[1], 0); tp.rx1 = std::min(rg[2], nx); tp.ry1 = std::min(rg[3], ny);
        tp.trav = reinterpret_cast<u64 *>(c->counters.p) + 3;
        HIP_TRY(hipMemsetAsync(queue, 0, sizeof(unsigned), st));
        int variant = c->accel == RTMI_ACCEL_BVH ? SCAN_BVH : c->scan_variant;
        { // a tree over a handful of mixed-kind primitives costs more than scanning them: a Cornell box's 18 (six of them too big for the tree anyway)
          // trace 8 % faster through the scalar-cache scan, the 3 - 8 of the small f3 / f4 scenes 3 - 10 %.  The two paths are bit-identical (tested
          // scene by scene), so the request for the tree is answered with the scan -- unless the tree's traversal counters were asked for.
            int below = c->flat_below;
            if (const char *e = std::getenv("RTMI_FLAT_BELOW")) below = std::atoi(e);
            if (variant == SCAN_BVH && s->dev.has_ext && !c->count_traversal && s->dev.n_all < below) variant = SCAN_SGPR_CULL;
            c->last_accel = variant == SCAN_BVH ? RTMI_ACCEL_BVH : RTMI_ACCEL_FLAT;
        }
        void (*kern)(ScenePtr, TraceParams) = nullptr;
        size_t dyn_lds = 0;
        const size_t bvh_lds = (size_t)(RTMI_BVH_STACK + RTMI_BVH_SUSPEND_WORDS) * kTraceBlock * sizeof(int); // stack columns + suspended cursors
        tp.suspend_lanes = c->suspend_lanes;
        if (s->dev.has_ext && !c->suspend_lanes_set) tp.suspend_lanes = 12; // make-final, 20 frames each, thresholds 8 / 10 / 12 / 14: 16.22 - 16.30 / 16.16 - 16.21 / 16.12 - 16.20 / 16.21 - 16.22 ms
        if (const char *e = std::getenv("RTMI_SUSPEND_LANES")) tp.suspend_lanes = std::max(0, std::min(64, std::atoi(e)));
        // Mixed-kind kernels: LDS = stack columns for THIS scene's tree (its depth is known: the sphere kernels use the compile-time RTMI_BVH_STACK for their
        // immediate ds_ offsets) + the parked cursors (time-sliced instantiation) + the camera-ray stash (11 words per entry, 17 when the rays' origins differ)
        tp.susp_off = RTMI_BVH_STACK * RTMI_BVH_STRIDE; tp.stash_off = 0;
        const char *lst_env = std::getenv("RTMI_SPHERE_LDS_STASH");
        constexpr bool kExtLdsStash = RTMI_STASH && RTMI_EXT_LDS_STASH && !RTMI_EXT_NO_STASH;
        const int ext_levels = std::max(4, std::min(RTMI_BVH_STACK, s->bvh_depth + 2));
        const int stash_words = kExtLdsStash ? (s->dev.cam_fixed_origin ? 11 : 17) : 0;
        auto ext_lds = [&](bool bvh, int susp_words) {
            const int levels = bvh ? ext_levels : 0;
            tp.susp_off = levels * RTMI_BVH_STRIDE;
            tp.stash_off = (levels + susp_words) * RTMI_BVH_STRIDE;
            return (size_t)(levels + susp_words + stash_words) * kTraceBlock * sizeof(int);
        };
        if (s->dev.has_ext && s->dev.media_seq) { // a Hitlist world holding media (RTMI_MEDIA_HITLIST): its own instantiations, never time-sliced
            const bool nar = s->dev.media_seq == 2; // RTMI_MEDIA_NARROWED: Hitlists holding media below bvh-nodes (MSEQ = 2)
            if (variant == SCAN_BVH) {
                if (nar) kern = c->count_traversal ? trace_kernel<double, false, SCAN_BVH, true, true, false, 2> : trace_kernel<double, false, SCAN_BVH, true, false, false, 2>;
                else kern = c->count_traversal ? trace_kernel<double, false, SCAN_BVH, true, true, false, 1> : trace_kernel<double, false, SCAN_BVH, true, false, false, 1>;
                dyn_lds = ext_lds(true, 0);
            } else { kern = nar ? trace_kernel<double, false, SCAN_SGPR_CULL, true, false, true, 2> : trace_kernel<double, false, SCAN_SGPR_CULL, true, false, true, 1>; dyn_lds = ext_lds(false, 0); }
        } else if (s->dev.has_ext) { // section 8(f3) scenes: FP64 kernels with the mixed-kind intersectors
            if (variant == SCAN_BVH) { // a Cornell box's 20-primitive tree loses 5 % to the time-slicing machinery, make-final's 3400 gain 8 %
                const bool slice = s->bvh_node_count >= 128 && tp.suspend_lanes > 0;
                if (c->count_traversal) kern = slice ? trace_kernel<double, false, SCAN_BVH, true, true> : trace_kernel<double, false, SCAN_BVH, true, true, false>;
                else kern = slice ? trace_kernel<double, false, SCAN_BVH, true> : trace_kernel<double, false, SCAN_BVH, true, false, false>;
                dyn_lds = ext_lds(true, slice ? RTMI_BVH_SUSPEND_WORDS_EXT : 0);
            }
            else { kern = trace_kernel<double, false, SCAN_SGPR_CULL, true>; dyn_lds = ext_lds(false, 0); }
        } else
        switch (variant) {
        case SCAN_BVH: // suspend_lanes = 0 or a small tree (< 128 inner nodes) selects the instantiation without the time-slicing machinery (the plain while-while loop);
                       // a scene with an entry grid always runs the time-sliced one (threshold 0 = never park early): the piecewise walk of long segments lives there
            if ((tp.suspend_lanes > 0 && s->bvh_node_count >= 128) || s->dev.grid_n > 0) kern = c->count_traversal ? trace_kernel<R, false, SCAN_BVH, false, true> : trace_kernel<R, false, SCAN_BVH>;
            else kern = c->count_traversal ? trace_kernel<R, false, SCAN_BVH, false, true, false> : trace_kernel<R, false, SCAN_BVH, false, false, false>;
            dyn_lds = bvh_lds;
            // The time-sliced sphere kernel keeps its camera-ray stash in LDS too when the scene's tree leaves room for it beside the stack columns (a dead lane reads
            // its entry with 6 ds_read instead of 11 ds_bpermute, and 14 VGPRs come free): C3 69.30 -> 68.76 ms, C2 3.067 -> 3.040 (RTMI_SPHERE_LDS_STASH=0: the
            // register stash, which deeper trees -- more than 21 levels with their grid entries -- keep anyway: a fifth kilobyte-row would cost the fourth workgroup per CU)
            if (!(lst_env && lst_env[0] == '0') && !c->count_traversal && kern == (void (*)(ScenePtr, TraceParams))trace_kernel<R, false, SCAN_BVH>) {
                const int levels = std::max(4, std::min(RTMI_BVH_STACK, s->bvh_depth + 2));
                const int words = s->dev.cam_fixed_origin ? 11 : 17;
                if ((size_t)(levels + RTMI_BVH_SUSPEND_WORDS + words) * kTraceBlock * sizeof(int) <= 40 * 1024) { // four workgroups per CU still fit
                    kern = trace_kernel<R, false, SCAN_BVH, false, false, true, false, true>;
                    tp.susp_off = levels * RTMI_BVH_STRIDE;
                    tp.stash_off = (levels + RTMI_BVH_SUSPEND_WORDS) * RTMI_BVH_STRIDE;
                    dyn_lds = (size_t)(levels + RTMI_BVH_SUSPEND_WORDS + words) * kTraceBlock * sizeof(int);
                }
            }
            break;
        case SCAN_SGPR_CULL: kern = trace_kernel<R, false, SCAN_SGPR_CULL>; break;
        case SCAN_SGPR: kern = trace_kernel<R, false, SCAN_SGPR>; break;
        case SCAN_LDS_PIPE: kern = multi ? trace_kernel<R, true, SCAN_LDS_PIPE> : trace_kernel<R, false, SCAN_LDS_PIPE>; dyn_lds = lds_bytes; break;
        default: kern = multi ? trace_kernel<R, true, SCAN_LDS_LITERAL> : trace_kernel<R, false, SCAN_LDS_LITERAL>; dyn_lds = lds_bytes;
        }
        // persistent launch: as many workgroups as stay resident (at most blocks_per_cu per CU); the queue feeds them
        int resident = 0;
        { // the occupancy query is a runtime call per launch and replica: asked once per (kernel, LDS bytes) and kept on the context
            const std::pair<const void *, size_t> key(reinterpret_cast<const void *>(kern), dyn_lds);
            auto it = c->occupancy.find(key);
            if (it == c->occupancy.end()) {
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, kTraceBlock, dyn_lds));
                c->occupancy.emplace(key, resident);
            } else resident = it->second;
        }
        const int grid_trace = std::max(1, c->cus * std::max(1, std::min(c->blocks_per_cu * (256 / kTraceBlock), resident)));
        if (c->last_grid != grid_trace && std::getenv("RTMI_DEBUG"))
            fprintf(stderr, "[rtmi] trace launch: %d workgroups of %d threads (%d resident per CU by the occupancy query, cap %d), %zu B LDS each\n",
                    grid_trace, kTraceBlock, resident, c->blocks_per_cu * (256 / kTraceBlock), dyn_lds);
        c->last_grid = grid_trace;
        hipLaunchKernelGGL(kern, dim3(grid_trace), dim3(kTraceBlock), dyn_lds, st, s->d_dev, tp);
        HIP_TRY(hipGetLastError());
        if (e1) HIP_TRY(hipEventRecord(e1, st));
        const long long npx = (long long)n_local * 64;
        hipLaunchKernelGGL((reduce_kernel<R>), dim3((unsigned)((npx + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           reinterpret_cast<const R *>(c->samples.p), reinterpret_cast<R *>(c->accum.p),
                           reinterpret_cast<double *>(d_tiles_linear), reinterpret_cast<const int *>(c->tile_ids.p), tiles_x_of(nx), nx, ny,
                           n_local, s_begin, s_count, ns, d_counters ? cnt : nullptr, (u64)c->tile_valid_pixels, tp.rx0, tp.ry0, tp.rx1, tp.ry1);
        HIP_TRY(hipGetLastError());
        if (e1) { // timing: the reduction is the interval from the trace kernel's end event to this one
            const size_t k = (size_t)c->events_used - 1;
            while (c->events_r.size() <= k) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); c->events_r.push_back(e); }
            HIP_TRY(hipEventRecord(c->events_r[k], st));
        }
    }
#ifdef RTMI_STAMPS
    {
        const int grid_trace_dbg = c->last_grid;
        HIP_TRY(hipStreamSynchronize(st));
        unsigned long long h[3 * PH_SLOTS] = {0}, z[3 * PH_SLOTS] = {0};
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)));
        {
            static unsigned long long wt[2][4096];
            HIP_TRY(hipMemcpyFromSymbol(wt, HIP_SYMBOL(g_wg_t), sizeof(wt)));
            const int g = std::min(4096, grid_trace_dbg);
            unsigned long long t0 = ~0ull; for (int k = 0; k < g; ++k) t0 = std::min(t0, wt[0][k]);
            std::vector<double> e(g), b(g); for (int k = 0; k < g; ++k) { e[k] = (wt[1][k] - t0) * 1e-5; b[k] = (wt[0][k] - t0) * 1e-5; }
            std::sort(e.begin(), e.end()); std::sort(b.begin(), b.end());
            fprintf(stderr, "[stamps] workgroup start (ms after first): median %.3f max %.3f | end: min %.3f p10 %.3f median %.3f p90 %.3f max %.3f (last pass, %d workgroups)\n",
                    b[g / 2], b[g - 1], e[0], e[g / 10], e[g / 2], e[g * 9 / 10], e[g - 1], g);
        }
        static const char *names[PH_N] = {"loop/tail", "refill: generate 64 camera rays", "refill: claim + deal", "bvh: ray setup / resume", "bvh: big primitives (exact)",
                                          "bvh: descent (node visits)", "bvh: leaf exact tests", "bvh: loop control / park", "shade: hit record", "shade: |d| normalise",
                                          "shade: rand-in-unit-sphere", "shade: material record + directions", "shade: texture", "shade: store / rest", "shade: sphere uv", "(stamp calibration)",
                                          "bvh: grid entry / next piece of the walk", "media: chords, draws, log"};
        // every interval begins with the bookkeeping of the stamp that opened it: subtract the cost of one stamp (PH_CAL: back-to-back stamps) per stamp
        const double per_stamp = h[2 * PH_SLOTS + PH_CAL] ? (double)h[PH_CAL] / (double)h[2 * PH_SLOTS + PH_CAL] : 0.0;
        double tk[PH_N], lk[PH_N], tot = 0, totl = 0, raw = 0;
        for (int k = 0; k < PH_N; ++k) {
            raw += (double)h[k];
            const double t = (double)h[k], c = std::min(t, per_stamp * (double)h[2 * PH_SLOTS + k]);
            tk[k] = k == PH_CAL ? 0.0 : t - c;
            lk[k] = t > 0 ? (double)h[PH_SLOTS + k] * (tk[k] / t) : 0.0;
            tot += tk[k]; totl += lk[k];
        }
        fprintf(stderr, "[phases] one stamp = %.0f ticks; stamps took %.1f %% of the %.4g wave-ticks of this (diagnostic) launch and are subtracted below\n", per_stamp, 100 * (raw - tot) / raw, raw);
        fprintf(stderr, "[phases] %-36s %8s %8s %10s %10s %12s\n", "phase", "ticks %", "lanes", "masked %", "useful %", "stamps");
        for (int k = 0; k < PH_N; ++k) {
            if (!h[k] || k == PH_CAL) continue;
            const double t = tk[k], l = lk[k];
            fprintf(stderr, "[phases] %-36s %8.2f %8.1f %10.2f %10.2f %12llu\n", names[k], 100 * t / tot, t > 0 ? l / t : 0.0, 100 * (64 * t - l) / (64 * tot), 100 * l / (64 * tot), h[2 * PH_SLOTS + k]);
        }
        fprintf(stderr, "[phases] %-36s %8.2f %8.1f %10.2f %10.2f   (%.4g wave-ticks)\n", "total", 100.0, totl / tot, 100 * (64 * tot - totl) / (64 * tot), 100 * totl / (64 * tot), tot);
    }
#endif
    return RTMI_OK;
}

// ---- RTMI_ACCEL_BVH host build ----------------------------------------------------------------------------------------
// Binned-SAH binary BVH over the primitives' boxes, one primitive per leaf, each node carrying its two children's boxes
// (one 64-byte fetch per step).  Boxes are FLOAT, rounded outward and inflated by 2^-21 * obound (see slab_hit): the
// traversal is only a conservative filter in front of the exact FP64 sphere test, so the tree's shape affects speed, never
// results.  Primitives whose radius is a large fraction of the scene (sky dome, ground) are kept out of the tree.
struct BvhBox { double lo[3], hi[3]; };
struct BvhItem { BvhBox b; double cen[3]; int idx; };

inline void box_grow(BvhBox &a, const BvhBox &b) { for (int k = 0; k < 3; ++k) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); } }
inline BvhBox box_empty() { BvhBox b; for (int k = 0; k < 3; ++k) { b.lo[k] = 1e300; b.hi[k] = -1e300; } return b; }
inline double box_area(const BvhBox &b) { const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2]; return x < 0 ? 0.0 : 2.0 * (x * y + y * z + z * x); }
inline float f_down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -INFINITY); return f; }
inline float f_up(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, INFINITY); return f; }

#include <unistd.h> // getpid (the team below must not be used by a forked child)
// Host-side scene preparation (the device's trees) runs on a small TEAM of threads created once per process and kept: on the GPU boxes of this pool creating a
// thread costs ~0.3 ms, a team of 16 per call cost more than the 11 025 rectangle trees it built.  run(fn): the caller and every worker execute fn() once.
static std::atomic<int> g_build_single{0}; // test hook (rtmi_test_build_tree): build on the calling thread only
inline unsigned team_size() {
    unsigned n = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    if (const char *e = std::getenv("RTMI_BUILD_THREADS")) n = (unsigned)std::max(1, std::min(64, std::atoi(e)));
    return n;
}
inline unsigned build_threads() { return g_build_single.load() ? 1u : team_size(); }
class WorkTeam {
    std::vector<std::thread> th;
    std::mutex mu, use_mu;
    std::condition_variable cv, done_cv;
    const std::function<void()> *fn = nullptr;
    unsigned long gen = 0;
    unsigned pending = 0;
    bool stop = false;
    pid_t owner;
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void()> *f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = fn;
            }
            (*f)();
            { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) done_cv.notify_all(); }
        }
    }
public:
    explicit WorkTeam(unsigned n) : owner(getpid()) { for (unsigned t = 1; t < n; ++t) th.emplace_back([this] { loop(); }); }
    ~WorkTeam() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (std::thread &t : th) t.join(); }
    // one team per process, never destroyed (its threads wait on the condition variable until the process exits: no join in a static destructor, which a host
    // that unloads libraries in its own order -- a JVM, an interpreter -- could run while they still wait); a forked child has the object but not the threads
    static WorkTeam &get() { static WorkTeam *team = new WorkTeam(team_size()); return *team; }
    void run(const std::function<void()> &f) {
        std::unique_lock<std::mutex> use(use_mu, std::try_to_lock);
        if (!use.owns_lock() || th.empty() || g_build_single.load() || getpid() != owner) { f(); return; } // the team is busy with another host thread's scene: this one builds alone
        { std::lock_guard<std::mutex> lk(mu); fn = &f; pending = (unsigned)th.size(); ++gen; }
        cv.notify_all();
        f();
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
};
// fn(begin, end) over [0, n) in blocks taken from a shared counter; `first` (optional) is one more job some thread of the team takes before the blocks
template <typename F> void parallel_blocks(size_t n, size_t block, F fn, const std::function<void()> *first = nullptr) {
    std::atomic<size_t> next{0};
    std::atomic<bool> first_taken{first == nullptr};
    const std::function<void()> worker = [&]() {
        if (!first_taken.exchange(true)) (*first)();
        for (size_t b = next.fetch_add(block); b < n; b = next.fetch_add(block)) fn(b, std::min(n, b + block));
    };
    if (n < 4 * block && !first) { worker(); return; }
    WorkTeam::get().run(worker);
}
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct BvhBuilder {
    std::vector<BvhItem> items;
    std::vector<float> nodes; // 16 floats per node
    std::vector<char> moving; // by original primitive index
    std::vector<char> box6;   // by original primitive index: the first of six rectangles that form a Box (one leaf: RTMI_LEAF_BOX)
    double delta = 0.0;
    const BvhBuilder *flags = nullptr; // a per-job builder (entry-grid rectangle trees) reads the flag vectors of the scene's builder instead of copying them
    int leaf_code(int idx) const {
        const BvhBuilder &f = flags ? *flags : *this;
        return ~(idx | (f.moving[(size_t)idx] ? 0x40000000 : 0) | (!f.box6.empty() && f.box6[(size_t)idx] ? RTMI_LEAF_BOX : 0));
    }
    int sah_depth = 8, max_depth = 0;
    double min_frac = 0.0;    // experiments: RTMI_BVH_MIN_FRAC = smallest share of a node's primitives a child may get (balance)
    int sweep_max = 0;  // subtrees up to this many primitives: exact sweep SAH; above: 32 bins (build time)
    int sah_levels = 1 << 20; // experiments: RTMI_BVH_SAH_LEVELS = number of top levels split by SAH (median below)
    BvhBox bounds(int b, int e) const { BvhBox r = box_empty(); for (int i = b; i < e; ++i) box_grow(r, items[(size_t)i].b); return r; }
    // node record (16 floats): l.lo.xy l.hi.xy | r.lo.xy r.hi.xy | l.lo.z l.hi.z r.lo.z r.hi.z | left, right, 0, 0
    void put_box(int node, int side, const BvhBox &b) {
        float *q = &nodes[(size_t)node * 16];
        for (int k = 0; k < 2; ++k) { q[side * 4 + k] = f_down(b.lo[k] - delta); q[side * 4 + 2 + k] = f_up(b.hi[k] + delta); }
        q[8 + side * 2] = f_down(b.lo[2] - delta); q[8 + side * 2 + 1] = f_up(b.hi[2] + delta);
    }
    void put_empty_box(int node, int side) {
        float *q = &nodes[(size_t)node * 16];
        for (int k = 0; k < 2; ++k) { q[side * 4 + k] = INFINITY; q[side * 4 + 2 + k] = -INFINITY; }
        q[8 + side * 2] = INFINITY; q[8 + side * 2 + 1] = -INFINITY;
    }
    int build(int b, int e, int depth) { // returns the child code of the subtree over items [b, e): byte offset of the node, or a leaf code
        max_depth = std::max(max_depth, depth);
        if (e - b == 1) return leaf_code(items[(size_t)b].idx);
        const int node = (int)(nodes.size() / 16);
        nodes.resize(nodes.size() + 16, 0.0f);
        // split: binned SAH (32 bins) over all three axes, the cheapest split wins; median split on the longest axis as the fallback
        // (also beyond sah_depth, to bound the stack)
        double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
        for (int i = b; i < e; ++i) for (int k = 0; k < 3; ++k) { clo[k] = std::min(clo[k], items[(size_t)i].cen[k]); chi[k] = std::max(chi[k], items[(size_t)i].cen[k]); }
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
        int mid = (b + e) / 2;
        bool done = false;
        // SAH wherever the subtree can still be finished by median splits within the stack's depth: depth + ceil(log2(count)) + 1
        // levels at most (a global cap on the SAH depth left the deep, crowded parts of large scenes to median splits)
        int lgc = 1;
        while ((1 << lgc) < e - b) ++lgc;
        if (depth + lgc + 1 < sah_depth && e - b > 2 && depth < sah_levels && e - b <= sweep_max) {
            // exact sweep SAH: for each axis sort by centroid, try every split position (suffix boxes, then one forward pass)
            double best = 1e300; int best_pos = -1;
            const int n = e - b;
            std::vector<BvhItem> tmp((size_t)n), best_order;
            std::vector<double> suffix((size_t)n + 1);
            for (int ax = 0; ax < 3; ++ax) {
                if (!(chi[ax] - clo[ax] > 0)) continue;
                std::copy(items.begin() + b, items.begin() + e, tmp.begin());
                std::sort(tmp.begin(), tmp.end(), [&](const BvhItem &x, const BvhItem &y) { return x.cen[ax] < y.cen[ax] || (x.cen[ax] == y.cen[ax] && x.idx < y.idx); });
                BvhBox acc = box_empty();
                for (int i = n - 1; i > 0; --i) { box_grow(acc, tmp[(size_t)i].b); suffix[(size_t)i] = box_area(acc); }
                acc = box_empty();
                bool improved = false;
                for (int i = 0; i < n - 1; ++i) { // left = [0, i], right = [i+1, n)
                    box_grow(acc, tmp[(size_t)i].b);
                    const double cost = box_area(acc) * (i + 1) + suffix[(size_t)i + 1] * (n - 1 - i);
                    if (std::min(i + 1, n - 1 - i) < min_frac * n) continue;
                    if (cost < best) { best = cost; best_pos = i + 1; improved = true; }
                }
                if (improved) best_order = tmp;
            }
            if (best_pos > 0) {
                std::copy(best_order.begin(), best_order.end(), items.begin() + b);
                mid = b + best_pos;
                done = true;
            }
        } else if (depth + lgc + 1 < sah_depth && e - b > 2 && depth < sah_levels) {
            const int NB = 32;
            double best = 1e300; int best_k = -1, best_axis = -1;
            for (int ax = 0; ax < 3; ++ax) {
                const double ext = chi[ax] - clo[ax];
                if (!(ext > 0)) continue;
                // Only OCCUPIED bins matter: between two occupied bins the two sides of a split -- boxes and counts -- do not change, so every split position of
                // such a run costs the same and the strict `cost < best` keeps the run's first, which is the occupied bin itself.  Walking the occupied bins (at
                // most e - b of them) instead of all 32 gives the same split for a fraction of the work on the small sets of the entry grid's rectangle trees
                // (3 - 14 primitives each, 11 025 trees at C3).
                BvhBox bb[NB]; int cnt[NB];
                unsigned occ = 0;
                for (int i = b; i < e; ++i) {
                    const int q = std::min(NB - 1, std::max(0, (int)((items[(size_t)i].cen[ax] - clo[ax]) / ext * NB)));
                    if (!((occ >> q) & 1u)) { bb[q] = box_empty(); cnt[q] = 0; occ |= 1u << q; }
                    box_grow(bb[q], items[(size_t)i].b); cnt[q]++;
                }
                int list[NB], m = 0;
                for (int q = 0; q < NB; ++q) if ((occ >> q) & 1u) list[m++] = q;
                BvhBox right[NB]; int rc[NB]; // right[j] / rc[j]: the occupied bins list[j], list[j + 1], ...
                BvhBox acc = box_empty(); int n = 0;
                for (int j = m - 1; j > 0; --j) { box_grow(acc, bb[list[j]]); n += cnt[list[j]]; right[j] = acc; rc[j] = n; }
                acc = box_empty(); n = 0;
                for (int j = 0; j + 1 < m; ++j) { // split after bin k = list[j]
                    box_grow(acc, bb[list[j]]); n += cnt[list[j]];
                    if (std::min(n, rc[j + 1]) < min_frac * (e - b)) continue;
                    const double cost = box_area(acc) * n + box_area(right[j + 1]) * rc[j + 1];
                    if (cost < best) { best = cost; best_k = list[j]; best_axis = ax; }
                }
            }
            if (best_k >= 0) {
                const double ext = chi[best_axis] - clo[best_axis], lo = clo[best_axis];
                auto it = std::partition(items.begin() + b, items.begin() + e, [&](const BvhItem &x) {
                    return std::min(NB - 1, std::max(0, (int)((x.cen[best_axis] - lo) / ext * NB))) <= best_k; });
                mid = (int)(it - items.begin());
                done = mid > b && mid < e;
            }
        }
        if (!done) {
            mid = (b + e) / 2;
            std::nth_element(items.begin() + b, items.begin() + mid, items.begin() + e,
                             [&](const BvhItem &x, const BvhItem &y) { return x.cen[axis] < y.cen[axis] || (x.cen[axis] == y.cen[axis] && x.idx < y.idx); });
        }
        const BvhBox lb = bounds(b, mid), rb = bounds(mid, e);
        const int l = build(b, mid, depth + 1);
        const int r = build(mid, e, depth + 1);
        put_box(node, 0, lb);
        put_box(node, 1, rb);
        std::memcpy(&nodes[(size_t)node * 16 + 12], &l, 4);
        std::memcpy(&nodes[(size_t)node * 16 + 13], &r, 4);
        return node * 64;
    }
};

// World-space box of primitive i in double (with a little slack): the local box of the innermost record, then each
// instance wrapper's outward map applied to its 8 corners, innermost wrapper first (RotateY: hitable.clj:441-443; Translate: 396).
// MovingSpheres: the sweep over the shutter interval.  Returns false when the primitive cannot be bounded.
bool prim_world_box(int kind, const double *g, const int32_t *xf_kind, const double *xf_param, int xf_first, int xf_count,
                    double t_lo, double t_hi, BvhBox &out) {
    BvhBox b;
    if (kind <= RTMI_PRIM_MOVING) {
        const double r = std::fabs(g[3]);
        if (kind == RTMI_PRIM_MOVING) {
            const double f0 = (t_lo - g[7]) / (g[8] - g[7]), f1 = (t_hi - g[7]) / (g[8] - g[7]);
            if (!std::isfinite(f0) || !std::isfinite(f1)) return false;
            for (int k = 0; k < 3; ++k) {
                const double a0 = g[k] * (1.0 - f0) + g[4 + k] * f0, a1 = g[k] * (1.0 - f1) + g[4 + k] * f1;
                b.lo[k] = std::min(a0, a1) - r; b.hi[k] = std::max(a0, a1) + r;
            }
        } else for (int k = 0; k < 3; ++k) { b.lo[k] = g[k] - r; b.hi[k] = g[k] + r; }
    } else if (kind <= RTMI_PRIM_RECT_YZ) {
        const int ax = kind == RTMI_PRIM_RECT_XY ? 2 : (kind == RTMI_PRIM_RECT_XZ ? 1 : 0);
        const int ua = kind == RTMI_PRIM_RECT_YZ ? 1 : 0, va = kind == RTMI_PRIM_RECT_XY ? 1 : 2;
        b.lo[ua] = std::min(g[0], g[2]); b.hi[ua] = std::max(g[0], g[2]);
        b.lo[va] = std::min(g[1], g[3]); b.hi[va] = std::max(g[1], g[3]);
        b.lo[ax] = g[4]; b.hi[ax] = g[4];
    } else {
        for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(g[k], std::min(g[3 + k], g[6 + k])); b.hi[k] = std::max(g[k], std::max(g[3 + k], g[6 + k])); }
    }
    if ((kind == RTMI_PRIM_SPHERE || kind == RTMI_PRIM_UVSPHERE) && xf_count > 0) {
        // A sphere under Translate / RotateY wrappers is a sphere of the same radius about the mapped centre: its world box is centre +- r, not the box of the
        // eight rotated corners of its local box (a RotateY of 15 degrees grows that one by a fifth per side -- half again the area -- and make-final's
        // thousand spheres sit behind one).  The wrappers' own rounding moves a hit point by ~1e-13 of a coordinate; the slack below and the tree's 2^-21 obound cover it.
        double c[3] = {g[0], g[1], g[2]};
        const double r = std::fabs(g[3]);
        for (int q = xf_count - 1; q >= 0; --q) {
            const double *p = xf_param + (size_t)(xf_first + q) * 3;
            if (xf_kind[xf_first + q] == RTMI_XFORM_TRANSLATE) { c[0] += p[0]; c[1] += p[1]; c[2] += p[2]; }
            else { const double sn = p[0], cs = p[1]; const double rx = cs * c[0] + sn * c[2], rz = -(sn * c[0]) + cs * c[2]; c[0] = rx; c[2] = rz; }
        }
        for (int k = 0; k < 3; ++k) { const double pad = 1e-9 * (std::fabs(c[k]) + r) + 1e-12; b.lo[k] = c[k] - r - pad; b.hi[k] = c[k] + r + pad; }
    } else
    for (int q = xf_count - 1; q >= 0; --q) {
        const double *p = xf_param + (size_t)(xf_first + q) * 3;
        BvhBox nb = box_empty();
        for (int c = 0; c < 8; ++c) {
            double x = (c & 1) ? b.hi[0] : b.lo[0], y = (c & 2) ? b.hi[1] : b.lo[1], z = (c & 4) ? b.hi[2] : b.lo[2];
            if (xf_kind[xf_first + q] == RTMI_XFORM_TRANSLATE) { x += p[0]; y += p[1]; z += p[2]; }
            else { const double sn = p[0], cs = p[1]; const double rx = cs * x + sn * z, rz = -(sn * x) + cs * z; x = rx; z = rz; }
            const double pt[3] = {x, y, z};
            for (int k = 0; k < 3; ++k) { nb.lo[k] = std::min(nb.lo[k], pt[k]); nb.hi[k] = std::max(nb.hi[k], pt[k]); }
        }
        b = nb;
    }
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(b.lo[k]) || !std::isfinite(b.hi[k]) || std::fabs(b.lo[k]) > 1e15 || std::fabs(b.hi[k]) > 1e15) return false;
        const double slack = 1e-9 * (std::fabs(b.lo[k]) + std::fabs(b.hi[k])) + 1e-12;
        b.lo[k] -= slack; b.hi[k] += slack;
    }
    out = b;
    return true;
}

// the IEEE half at or beyond x in the given direction (up: >= x, else <= x); beyond the half range: +-inf.  Integer arithmetic on the float's bits (directed
// rounding of the magnitude: toward zero by truncation, away from zero by truncation + 1 when inexact): a host without F16C converts _Float16 in software, and a
// scene's tree has twelve planes per node (C3: 1.4 million conversions there and back).  rtmi_test_half_outward exposes it to the CPU test against numpy.
static uint16_t half_outward(float x, bool up) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    const uint32_t sign = u >> 31, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(0x7e00u | (sign << 15)); // NaN
    const bool away = up != (sign != 0); // the magnitude rounds away from zero
    uint32_t m;
    bool inexact;
    if (a >= 0x47800000u) { m = a == 0x7f800000u ? 0x7c00u : 0x7bffu; inexact = a != 0x7f800000u; } // >= 2^16: the largest half (65504) toward zero, inf away
    else if (a >= 0x38800000u) { m = (((a >> 23) - 112u) << 10) | ((a & 0x7fffffu) >> 13); inexact = (a & 0x1fffu) != 0; } // normal halves: 2^-14 <= |x| < 2^16
    else { const float sc = std::fabs(x) * 16777216.0f; m = (uint32_t)sc; inexact = (float)m != sc; } // half subnormals: units of 2^-24 (the scaling is exact)
    if (inexact && away) m += 1; // (carries into the exponent: 0x03ff + 1 = the smallest normal, 0x7bff + 1 = inf)
    return (uint16_t)(m | (sign << 15));
}
// fills d.bvh_* ; returns the node array to upload.  wbox[i] / bounded[i]: prim_world_box of every primitive.
// box_first[i] != 0: primitives i .. i + 5 are the six faces of one Box (detected at scene creation) -- one leaf, unless the box is too large for the tree
std::vector<float> build_bvh(DevScene &d, int n_prims, const int *prim_kind, const std::vector<BvhBox> &wbox, const std::vector<char> &bounded, const double *cam,
                             bool want_grid, std::vector<int> &grid_cells, const std::vector<char> &box_first, int *out_depth = nullptr, const std::vector<BvhBox> *media_boxes = nullptr) {
    BvhBuilder B;
    struct DepthOut { BvhBuilder &b; int *o; ~DepthOut() { if (o) *o = b.max_depth; } } depth_out{B, out_depth};
    std::vector<BvhItem> all;
    double obound = 0.0;
    for (int k = 0; k < 3; ++k) obound = std::max(obound, std::fabs(cam[k]));
    for (int i = 0; i < n_prims; ++i) {
        if (prim_kind[i] == RTMI_PRIM_MEDIUM) continue; // media are not surfaces (ext_medium_test)
        BvhItem it; it.idx = i;
        if (bounded[(size_t)i]) {
            it.b = wbox[(size_t)i];
            for (int k = 0; k < 3; ++k) it.cen[k] = 0.5 * (it.b.lo[k] + it.b.hi[k]);
        } else { for (int k = 0; k < 3; ++k) { it.b.lo[k] = -1e15; it.b.hi[k] = 1e15; it.cen[k] = 0; } } // unbounded: goes to the big list below
        for (int k = 0; k < 3; ++k) obound = std::max(obound, std::max(std::fabs(it.b.lo[k]), std::fabs(it.b.hi[k])));
        all.push_back(it);
    }
    obound = std::min(obound, 1e15) * 1.001 + 1e-30;
    d.n_big = 0;
    B.box6.assign((size_t)std::max(n_prims, 1), 0);
    for (size_t a = 0; a < all.size(); ++a) {
        BvhItem it = all[a];
        if (!box_first.empty() && box_first[(size_t)it.idx] && a + 5 < all.size() && all[a + 5].idx == it.idx + 5) { // a Box: the union of its six faces, if that fits the tree
            BvhBox u = it.b;
            bool ok = bounded[(size_t)it.idx] != 0;
            for (int k = 1; k < 6; ++k) { box_grow(u, all[a + (size_t)k].b); ok = ok && bounded[(size_t)it.idx + (size_t)k]; }
            const double uext = std::max(u.hi[0] - u.lo[0], std::max(u.hi[1] - u.lo[1], u.hi[2] - u.lo[2]));
            if (ok && uext < 0.25 * obound) {
                it.b = u;
                for (int k = 0; k < 3; ++k) it.cen[k] = 0.5 * (u.lo[k] + u.hi[k]);
                B.box6[(size_t)it.idx] = 1;
                B.items.push_back(it);
                a += 5;
                continue;
            }
        }
        const double ext = std::max(it.b.hi[0] - it.b.lo[0], std::max(it.b.hi[1] - it.b.lo[1], it.b.hi[2] - it.b.lo[2]));
        if (ext >= 0.25 * obound && d.n_big < 16) d.big_idx[d.n_big++] = it.idx; // ascending index order
        else B.items.push_back(it);
    }
    B.delta = obound * (1.0 / 2097152.0); // 2^-21 * obound
    B.moving.assign((size_t)std::max(n_prims, 1), 0);
    for (int i = 0; i < n_prims; ++i) B.moving[(size_t)i] = prim_kind[i] == RTMI_PRIM_MOVING;
    d.bvh_obound = f_down(obound);
    double cbound = 0.0;
    for (const BvhItem &it : B.items) for (int k = 0; k < 3; ++k) cbound = std::max(cbound, std::max(std::fabs(it.b.lo[k]), std::fabs(it.b.hi[k])));
    d.bvh_cbound = f_up(cbound);
    std::vector<BvhItem> grid_items;
    std::function<void()> whole_job; // the whole tree's build, when it is deferred to run beside the grid's jobs
    if (B.items.empty()) d.bvh_root = RTMI_BVH_EMPTY;
    else if (B.items.size() == 1) { // a lone primitive: a node whose right child is an empty box
        B.nodes.assign(16, 0.0f);
        B.put_box(0, 0, B.items[0].b);
        B.put_empty_box(0, 1);
        const int l = B.leaf_code(B.items[0].idx), r = l;
        std::memcpy(&B.nodes[12], &l, 4); std::memcpy(&B.nodes[13], &r, 4);
        d.bvh_root = 0;
    } else {
        int lg = 1;
        while ((1u << lg) < B.items.size()) ++lg;
        if (const char *e = std::getenv("RTMI_BVH_SAH_LEVELS")) B.sah_levels = std::atoi(e);
        if (const char *e = std::getenv("RTMI_BVH_SWEEP_MAX")) B.sweep_max = std::atoi(e);
        if (const char *e = std::getenv("RTMI_BVH_MIN_FRAC")) B.min_frac = std::atof(e);
        B.sah_depth = RTMI_BVH_STACK - 2; // depth budget: a node at depth d over k primitives may use SAH while d + ceil(log2 k) + 1 < budget
        (void)lg;
        // The whole tree and the entry grid's rectangle trees are independent: when a grid will be tried, the whole tree is built on a thread of its own
        // (into B.nodes, which the grid's jobs do not touch: they build into builders of their own and are appended after the join)
        grid_items.assign(B.items.begin(), B.items.end()); // (build() reorders B.items: the grid works on a copy taken before)
        auto build_whole = [&B, &d]() {
            const double tb0 = now_ms();
            d.bvh_root = B.build(0, (int)B.items.size(), 0);
            if (std::getenv("RTMI_DEBUG")) fprintf(stderr, "[rtmi] build: whole tree over %zu primitives %.2f ms\n", B.items.size(), now_ms() - tb0);
        };
        if (want_grid && B.items.size() >= 256 && build_threads() > 1) whole_job = build_whole; // deferred: the team's first job, beside the grid's rectangle trees
        else build_whole();
    }
    auto join_whole = [&]() {
        if (whole_job) { whole_job(); whole_job = nullptr; } // (no grid was built after all: build it here)
        if (d.bvh_root >= 0 && (B.max_depth >= RTMI_BVH_STACK - 1 || B.nodes.size() / 16 >= (1u << 25))) { // cannot happen by construction / node byte offsets are 31-bit
            d.bvh_root = RTMI_BVH_EMPTY; d.n_big = 0; d.bvh_obound = -1.0f; // obound < 0: every ray takes the exact flat scan
            B.nodes.clear();
        }
    };
    // ---- entry grid: a BVH per x-z cell over the primitives whose boxes overlap the cell (DevScene::grid_*) -------------------------------------
    d.grid_n = 0; d.grid_tall = RTMI_BVH_EMPTY; d.grid_kmax = 4; d.grid_walk = 0;
    grid_cells.clear();
    const char *grid_env = std::getenv("RTMI_GRID"); // "0": off; "n": n x n cells (experiments)
    const double tg0 = now_ms();
    struct GridTimer { double t0; ~GridTimer() { if (std::getenv("RTMI_DEBUG")) fprintf(stderr, "[rtmi] build: entry grid + node formats %.2f ms\n", now_ms() - t0); } } grid_timer{tg0};
    if (want_grid && !(grid_env && grid_env[0] == '0') && grid_items.size() >= 256) { // (the whole tree may still be in the making: nothing below touches B.nodes / B.items before join_whole())
        const size_t n_items = grid_items.size();
        const std::vector<BvhItem> &world = grid_items; // (any order will do)
        // the layer: every primitive except the few much taller than the typical one (the cover scene's three big spheres among 10 000 small ones)
        std::vector<double> hts(n_items);
        for (size_t i = 0; i < n_items; ++i) hts[i] = world[i].b.hi[1] - world[i].b.lo[1];
        std::vector<double> sorted_h(hts);
        std::nth_element(sorted_h.begin(), sorted_h.begin() + (long)(n_items / 2), sorted_h.end());
        const double tall_h = 3.0 * sorted_h[n_items / 2] + 1e-300;
        std::vector<BvhItem> layer, tall;
        for (size_t i = 0; i < n_items; ++i) (hts[i] > tall_h ? tall : layer).push_back(world[i]);
        BvhBox lb = box_empty();
        for (const BvhItem &it : layer) box_grow(lb, it.b);
        const double ex = lb.hi[0] - lb.lo[0], ez = lb.hi[2] - lb.lo[2], ey = lb.hi[1] - lb.lo[1];
        // ~3.5 primitives per cell (C3: 53 x 53 cells, C2: 12 x 12).  Before long segments were walked in pieces (RTMI_GRID_CHUNK) ~10 per cell was best (C3, 16 .. 48
        // cells per side: 84.1 / 82.3 / 83.3 ms: a finer grid sent more rays to the root of the whole tree); with the walk 32 / 40 / 48 / 56 / 64 / 72 cells: 77.2 / 76.2 /
        // 76.4 / 76.1 / 76.7 / 78.5 ms, C2 7 / 10 / 14 / 20 cells: 3.39 / 3.37 / 3.33 / 3.49 ms
        // ... and cells no smaller than ~4.5 x the layer's height: a ray crosses the layer over a horizontal distance of height / tan(elevation), so a
        // taller layer (the moving cover scene: its spheres sweep up to 0.5 upwards, the layer is 0.9 instead of 0.4 high) at the same cell size means more
        // cells per segment, i.e. more pieces (C2 moving, 6 / 8 / 10 / 12 / 16 cells per side: 3.55 / 3.60 / 3.68 / 3.75 / 4.01 ms; for the static scenes both
        // rules give the same cell)
        int G = (int)std::lround(std::min(std::sqrt((double)layer.size() / 3.5), std::min(ex, ez) / (4.5 * std::max(ey, 1e-300))));
        if (grid_env && std::atoi(grid_env) > 1) G = std::atoi(grid_env);
        G = std::max(2, std::min(G, 256)); // (90 000 spheres: 160 x 160 cells)
        // worth it for a flat, wide layer of many primitives with few tall outliers
        if (layer.size() >= 256 && tall.size() * 20 <= n_items && ex > 0 && ez > 0 && ey < 0.25 * std::min(ex, ez)) {
            const double eps = 4.0 * B.delta; // cells claim the primitives whose (already inflated) boxes come this close; the device grows a ray's cell rectangle by its own position error
            const double csx = ex / G, csz = ez / G;
            std::vector<std::vector<int>> cell_items((size_t)G * G);
            for (size_t i = 0; i < layer.size(); ++i) {
                const BvhBox &b = layer[i].b;
                const int i0 = std::max(0, std::min(G - 1, (int)std::floor((b.lo[0] - 2 * eps - lb.lo[0]) / csx))), i1 = std::max(0, std::min(G - 1, (int)std::floor((b.hi[0] + 2 * eps - lb.lo[0]) / csx)));
                const int j0 = std::max(0, std::min(G - 1, (int)std::floor((b.lo[2] - 2 * eps - lb.lo[2]) / csz))), j1 = std::max(0, std::min(G - 1, (int)std::floor((b.hi[2] + 2 * eps - lb.lo[2]) / csz)));
                for (int j = j0; j <= j1; ++j) for (int ii = i0; ii <= i1; ++ii) cell_items[(size_t)j * G + ii].push_back((int)i);
            }
            size_t claimed = 0;
            for (const std::vector<int> &ci : cell_items) claimed += ci.size();
            if (claimed > 4 * layer.size()) cell_items.clear(); // primitives that each span many cells (long sweeps, slabs): the per-cell trees would multiply them -- no grid
            const int depth0 = 3; // stack entries a grid start may already hold: the rectangle's tree under the tall tree (+ margin)
            auto subtree = [&](const std::vector<BvhItem> &its) -> int { // child code of a tree over `its`, appended to B.nodes
                if (its.empty()) return RTMI_BVH_EMPTY;
                const int b0 = (int)B.items.size();
                B.items.insert(B.items.end(), its.begin(), its.end());
                if (its.size() == 1) { // a lone primitive: a node whose right child is an empty box (a bare leaf code would skip the box test)
                    const int node = (int)(B.nodes.size() / 16);
                    B.nodes.resize(B.nodes.size() + 16, 0.0f);
                    B.put_box(node, 0, its[0].b);
                    B.put_empty_box(node, 1);
                    const int l = B.leaf_code(its[0].idx);
                    std::memcpy(&B.nodes[(size_t)node * 16 + 12], &l, 4); std::memcpy(&B.nodes[(size_t)node * 16 + 13], &l, 4);
                    return node * 64;
                }
                return B.build(b0, b0 + (int)its.size(), depth0);
            };
            // One tree per RECTANGLE of cells a segment can touch -- 1 x 1, 2 x 1, 1 x 2, 2 x 2 (four families, each indexed by the rectangle's low corner:
            // grid_cells[(wi + 2 wj) G G + j0 G + i0]) -- over the union of the cells' primitives: a segment starts at ONE root (no root per cell to push and
            // to visit), and a primitive two cells of the rectangle share is in the tree once (it used to be tested exactly once per cell).
            if (!cell_items.empty()) grid_cells.assign((size_t)4 * G * G, RTMI_BVH_EMPTY);
            // The 4 G^2 rectangle trees (C3: 11 025 of them) are independent: every job builds its tree in a builder of its own, worker threads take jobs from a
            // shared counter, and the results are appended to the node array IN JOB ORDER with their node offsets rebased -- the same array whatever the thread
            // count (scene creation at C3: 59 ms single-threaded, most of it here).
            if (!cell_items.empty()) {
                struct RectTree { std::vector<float> nodes; int root = RTMI_BVH_EMPTY; int depth = 0; };
                const size_t n_jobs = (size_t)4 * G * G;
                std::vector<RectTree> trees(n_jobs);
                auto job = [&](size_t jb) {
                    const int fam = (int)(jb / ((size_t)G * G)), j = (int)((jb / (size_t)G) % (size_t)G), i = (int)(jb % (size_t)G);
                    const int wi = fam & 1, wj = fam >> 1;
                    if (j + wj >= G || i + wi >= G) return;
                    std::vector<int> uni;
                    for (int dj = 0; dj <= wj; ++dj) for (int di = 0; di <= wi; ++di) { const std::vector<int> &ci = cell_items[(size_t)(j + dj) * G + i + di]; uni.insert(uni.end(), ci.begin(), ci.end()); }
                    std::sort(uni.begin(), uni.end());
                    uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
                    if (uni.empty()) return;
                    BvhBuilder L;
                    L.flags = &B; L.delta = B.delta; L.sah_depth = B.sah_depth; L.min_frac = B.min_frac; L.sweep_max = B.sweep_max; L.sah_levels = B.sah_levels;
                    for (int k : uni) L.items.push_back(layer[(size_t)k]);
                    RectTree &T = trees[jb];
                    if (L.items.size() == 1) { // a lone primitive: a node whose right child is an empty box (a bare leaf code would skip the box test)
                        L.nodes.assign(16, 0.0f);
                        L.put_box(0, 0, L.items[0].b);
                        L.put_empty_box(0, 1);
                        const int l = L.leaf_code(L.items[0].idx);
                        std::memcpy(&L.nodes[12], &l, 4); std::memcpy(&L.nodes[13], &l, 4);
                        T.root = 0; T.depth = depth0 + 1;
                    } else { T.root = L.build(0, (int)L.items.size(), depth0); T.depth = L.max_depth; }
                    T.nodes.swap(L.nodes);
                };
                const double tr0 = now_ms();
                { // the whole tree (if deferred) is one more job of the same team
                    const std::function<void()> first = whole_job;
                    whole_job = nullptr;
                    parallel_blocks(n_jobs, 16, [&](size_t b, size_t e) { for (size_t q = b; q < e; ++q) job(q); }, first ? &first : nullptr);
                }
                if (std::getenv("RTMI_DEBUG")) fprintf(stderr, "[rtmi] build: %zu rectangle trees on %u threads %.2f ms\n", n_jobs, build_threads(), now_ms() - tr0);
                join_whole();
                // append in job order, node offsets rebased: every job's place in the array is the sum of the sizes before it, so the copies run on the team too
                std::vector<size_t> at(n_jobs + 1, B.nodes.size());
                for (size_t jb = 0; jb < n_jobs; ++jb) { at[jb + 1] = at[jb] + trees[jb].nodes.size(); B.max_depth = std::max(B.max_depth, trees[jb].depth); }
                B.nodes.resize(at[n_jobs]);
                parallel_blocks(n_jobs, 64, [&](size_t b, size_t e) {
                    for (size_t jb = b; jb < e; ++jb) {
                        RectTree &T = trees[jb];
                        if (T.root == RTMI_BVH_EMPTY) continue;
                        const int base = (int)(at[jb] / 16) * 64;
                        float *dst = &B.nodes[at[jb]];
                        std::memcpy(dst, T.nodes.data(), T.nodes.size() * sizeof(float));
                        for (size_t nd = 0; nd < T.nodes.size() / 16; ++nd) {
                            int c[2];
                            std::memcpy(c, dst + nd * 16 + 12, 8);
                            for (int k = 0; k < 2; ++k) if (c[k] >= 0) c[k] += base; // inner node: byte offset of its record (leaf codes and RTMI_BVH_EMPTY are negative)
                            std::memcpy(dst + nd * 16 + 12, c, 8);
                        }
                        grid_cells[jb] = T.root + base;
                    }
                });
            }
            join_whole();
            if (d.bvh_root < 0) cell_items.clear(); // (the whole tree did not fit the stack: every ray takes the flat scan, no grid either)
            if (!cell_items.empty()) d.grid_tall = subtree(tall);
            if (cell_items.empty() || B.max_depth >= RTMI_BVH_STACK - 1 || B.nodes.size() / 16 >= (1u << 25)) { // too deep for the stack: no grid (the whole tree above stays valid)
                grid_cells.clear(); d.grid_tall = RTMI_BVH_EMPTY;
            } else {
                d.grid_n = G;
                if (const char *e = std::getenv("RTMI_GRID_KMAX")) d.grid_kmax = std::max(1, std::min(4, std::atoi(e)));
                d.grid_walk = d.grid_kmax >= 4;
                if (const char *e = std::getenv("RTMI_GRID_WALK")) d.grid_walk = d.grid_walk && e[0] != '0'; // (tests: the same grid without the piecewise walk)
                d.grid_lo_x = (float)lb.lo[0]; d.grid_lo_z = (float)lb.lo[2];
                d.grid_inv_x = (float)(1.0 / csx); d.grid_inv_z = (float)(1.0 / csz);
                for (int k = 0; k < 3; ++k) { d.grid_box[k] = f_down(lb.lo[k] - B.delta - eps); d.grid_box[3 + k] = f_up(lb.hi[k] + B.delta + eps); }
                d.grid_eps = 0.0f;
                BvhBox tb = box_empty();
                for (const BvhItem &it : tall) box_grow(tb, it.b);
                for (int k = 0; k < 3; ++k) { d.grid_tall_box[k] = tall.empty() ? 0.0f : f_down(tb.lo[k] - B.delta - eps); d.grid_tall_box[3 + k] = tall.empty() ? 0.0f : f_up(tb.hi[k] + B.delta + eps); }
                for (int k = 0; k < 3; ++k) { // what the device tests: {lo, hi} half pairs, rounded outward once more
                    d.grid_box_h[k] = (unsigned)half_outward(d.grid_box[k], false) | ((unsigned)half_outward(d.grid_box[3 + k], true) << 16);
                    d.grid_tall_box_h[k] = (unsigned)half_outward(d.grid_tall_box[k], false) | ((unsigned)half_outward(d.grid_tall_box[3 + k], true) << 16);
                }
            }
        }
    }
    join_whole();
    // ---- neighbourhood trees of the media (DevScene::mloc_*) ------------------------------------------------------------------------------------------------
    d.n_mloc = 0;
    // (measured on make-final: node visits per segment 10.2 -> 8.3, frame 17.43 vs 17.46 ms -- the segments it shortens finish early and wait for their wave's
    // long ones; off unless RTMI_MLOC=1)
    const char *mloc_env = std::getenv("RTMI_MLOC");
    if (media_boxes && d.bvh_root >= 0 && mloc_env && mloc_env[0] == '1') {
        const size_t n_tree = B.items.size(); // (the grid is never built for a scene with media: the items are the whole tree's)
        std::vector<BvhItem> world(B.items.begin(), B.items.end());
        for (const BvhBox &mb : *media_boxes) {
            if (d.n_mloc >= 4) break;
            BvhBox R = mb; // the boundary's box, a hundredth larger per side
            for (int k = 0; k < 3; ++k) { const double pad = 0.01 * (mb.hi[k] - mb.lo[k]) + 4.0 * B.delta; R.lo[k] -= pad; R.hi[k] += pad; }
            std::vector<BvhItem> its;
            for (const BvhItem &it : world) { // every primitive with a surface point inside R: its box (the tree inflates it by delta once more) reaches into R
                bool hit = true;
                for (int k = 0; k < 3; ++k) hit = hit && it.b.hi[k] + 2.0 * B.delta >= R.lo[k] && it.b.lo[k] - 2.0 * B.delta <= R.hi[k];
                if (hit) its.push_back(it);
            }
            if (its.size() * 2 > n_tree) continue; // a medium that holds most of the scene (make-final's haze): nothing to gain
            int root = RTMI_BVH_EMPTY;
            if (its.size() == 1) { // a lone primitive: a node whose right child is an empty box
                const int node = (int)(B.nodes.size() / 16);
                B.nodes.resize(B.nodes.size() + 16, 0.0f);
                B.put_box(node, 0, its[0].b);
                B.put_empty_box(node, 1);
                const int l = B.leaf_code(its[0].idx);
                std::memcpy(&B.nodes[(size_t)node * 16 + 12], &l, 4); std::memcpy(&B.nodes[(size_t)node * 16 + 13], &l, 4);
                root = node * 64;
            } else if (!its.empty()) {
                const int b0 = (int)B.items.size();
                B.items.insert(B.items.end(), its.begin(), its.end());
                root = B.build(b0, b0 + (int)its.size(), 1);
            }
            if (B.max_depth >= RTMI_BVH_STACK - 1 || B.nodes.size() / 16 >= (1u << 25)) break; // (cannot happen: a subset of a tree that fitted)
            d.mloc_root[d.n_mloc] = root;
            for (int k = 0; k < 3; ++k) { d.mloc_box[d.n_mloc][k] = f_up(R.lo[k]); d.mloc_box[d.n_mloc][3 + k] = f_down(R.hi[k]); }
            d.n_mloc++;
        }
    }
    d.bvh_node16 = 0;
    if (d.bvh_root != RTMI_BVH_EMPTY) { // 32-byte records (Node16) when rounding the planes to half costs little: 12 halves + 2 child codes
        auto half_bits = [](float x, bool up) { return half_outward(x, up); };
        auto half_val = [](uint16_t b) { // the half's value (integer decode: no software _Float16 conversion)
            const int e = (b >> 10) & 31, m = b & 1023;
            const double v = e == 0 ? std::ldexp((double)m, -24) : (e == 31 ? (m ? (double)NAN : (double)INFINITY) : std::ldexp((double)(1024 + m), e - 25));
            return (b >> 15) ? -v : v;
        };
        std::vector<float> out(B.nodes.size() / 2, 0.0f);
        double area32 = 0.0, area16 = 0.0;
        const size_t n_nodes = B.nodes.size() / 16;
        std::vector<double> part32((n_nodes + 2047) / 2048 + 1, 0.0), part16(part32.size(), 0.0); // per block, summed in block order: the same sums whatever the thread count
        parallel_blocks(n_nodes, 2048, [&](size_t nb, size_t ne) {
            double a32 = 0.0, a16 = 0.0;
            for (size_t n = nb; n < ne; ++n) {
                const float *q = &B.nodes[n * 16];
                uint16_t h[12];
                for (int side = 0; side < 2; ++side) {
                    const float lo[3] = {q[side * 4], q[side * 4 + 1], q[8 + side * 2]}, hi[3] = {q[side * 4 + 2], q[side * 4 + 3], q[8 + side * 2 + 1]};
                    double e32[3], e16[3];
                    for (int k = 0; k < 3; ++k) {
                        h[side * 6 + k * 2] = half_bits(lo[k], false); h[side * 6 + k * 2 + 1] = half_bits(hi[k], true);
                        e32[k] = (double)hi[k] - lo[k]; e16[k] = half_val(h[side * 6 + k * 2 + 1]) - half_val(h[side * 6 + k * 2]);
                    }
                    if (e32[0] >= 0 && std::isfinite(e32[0] + e32[1] + e32[2])) { // (the lone primitive's empty sibling is +inf / -inf)
                        a32 += e32[0] * e32[1] + e32[1] * e32[2] + e32[2] * e32[0];
                        a16 += std::isfinite(e16[0] + e16[1] + e16[2]) ? e16[0] * e16[1] + e16[1] * e16[2] + e16[2] * e16[0] : INFINITY;
                    }
                }
                int c[2];
                std::memcpy(c, &q[12], 8);
                for (int k = 0; k < 2; ++k) if (c[k] >= 0 && c[k] != RTMI_BVH_EMPTY) c[k] /= 2; // byte offsets of 32-byte records
                std::memcpy(reinterpret_cast<char *>(&out[n * 8]), h, 24);
                std::memcpy(reinterpret_cast<char *>(&out[n * 8]) + 24, c, 8);
            }
            part32[nb / 2048] = a32; part16[nb / 2048] = a16;
        });
        for (size_t k = 0; k < part32.size(); ++k) { area32 += part32[k]; area16 += part16[k]; }
        const char *force = std::getenv("RTMI_NODE16"); // "0" / "1": override the choice (tests)
        const bool use16 = force ? force[0] == '1' : (area16 <= 1.25 * area32);
        if (use16) {
            d.bvh_node16 = 1;
            d.bvh_root = d.bvh_root >= 0 ? d.bvh_root / 2 : d.bvh_root;
            for (int &c : grid_cells) if (c >= 0) c /= 2;
            for (int k = 0; k < d.n_mloc; ++k) if (d.mloc_root[k] >= 0) d.mloc_root[k] /= 2;
            if (d.grid_tall >= 0) d.grid_tall /= 2;
            return out;
        }
    }
    return B.nodes;
}

int check_render_args(rtmi_scene *s, int nx, int ny, int ns, int depth, int precision) {
    if (!scene_ok(s)) return fail(RTMI_E_STATE, "invalid scene handle");
    if (nx <= 0 || ny <= 0 || ns <= 0 || depth < 0) return fail(RTMI_E_ARG, "nx, ny, ns must be > 0 and depth >= 0 (got %d %d %d %d)", nx, ny, ns, depth);
    if ((long long)nx * ny > (1ll << 30)) return fail(RTMI_E_ARG, "frame too large");
    if (precision != RTMI_F64 && precision != RTMI_F32) return fail(RTMI_E_ARG, "precision must be RTMI_F64 or RTMI_F32");
    if (precision == RTMI_F32 && s->dev.has_ext) return fail(RTMI_E_UNSUPPORTED, "rectangles / triangles / instances / procedural textures are rendered by the FP64 kernels only");
    if (s->uses_perlin && !s->have_perlin) return fail(RTMI_E_STATE, "the scene holds a Perlin texture: call rtmi_scene_set_perlin first");
    if (s->max_image >= s->dev.n_images) return fail(RTMI_E_STATE, "the scene holds an ImageMap with index %d: call rtmi_scene_set_images first", s->max_image);
    return RTMI_OK;
}

} // namespace

// ---- library / context -------------------------------------------------------------------------------
// test hook, host code only (no device): the device's tree (+ entry grid) over n spheres as rtmi_scene_create builds it, on the team or (threads = 1) on the
// calling thread alone.  out_hash = FNV-1a of the node array and the grid's root codes, out_info = {node records, depth, grid cells per side, big primitives}
RTMI_EXPORT int rtmi_test_build_tree(int32_t n, const double *geom, const double *cam, int32_t threads, uint64_t *out_hash, int32_t *out_info, double *out_ms) {
    if (n <= 0 || !geom || !cam || !out_hash || !out_info || !out_ms) return fail(RTMI_E_ARG, "bad arguments");
    std::vector<int> kind((size_t)n, RTMI_PRIM_SPHERE);
    std::vector<BvhBox> wbox((size_t)n);
    std::vector<char> bounded((size_t)n, 0), box_first((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        const double g[9] = {geom[4 * (size_t)i], geom[4 * (size_t)i + 1], geom[4 * (size_t)i + 2], geom[4 * (size_t)i + 3], 0, 0, 0, 0, 1};
        bounded[(size_t)i] = prim_world_box(RTMI_PRIM_SPHERE, g, nullptr, nullptr, 0, 0, 0.0, 1.0, wbox[(size_t)i]);
    }
    DevScene d{};
    std::vector<int> grid_cells;
    int depth = 0;
    g_build_single.store(threads == 1 ? 1 : 0);
    const double t0 = now_ms();
    const std::vector<float> nodes = build_bvh(d, n, kind.data(), wbox, bounded, cam, true, grid_cells, box_first, &depth);
    *out_ms = now_ms() - t0;
    g_build_single.store(0);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t bytes) { const unsigned char *q = (const unsigned char *)p; for (size_t k = 0; k < bytes; ++k) { h ^= q[k]; h *= 1099511628211ull; } };
    mix(nodes.data(), nodes.size() * sizeof(float));
    mix(grid_cells.data(), grid_cells.size() * sizeof(int));
    mix(&d.bvh_root, sizeof(int)); mix(&d.grid_tall, sizeof(int)); mix(&d.bvh_node16, sizeof(int));
    *out_hash = h;
    out_info[0] = (int32_t)(nodes.size() / (d.bvh_node16 ? 8 : 16)); out_info[1] = depth; out_info[2] = d.grid_n; out_info[3] = d.n_big;
    return RTMI_OK;
}
RTMI_EXPORT int rtmi_test_half_outward(double x, int32_t up) { return (int)half_outward((float)x, up != 0); } // test hook (host arithmetic only: no device needed)
RTMI_EXPORT const char *rtmi_last_error(void) { return g_err.c_str(); }
RTMI_EXPORT const char *rtmi_backend_name(void) { return "hip-gfx950"; }
RTMI_EXPORT int rtmi_version(void) { return 204; } // 204: rtmi_probe_math2 (explicit slot count; rtmi_probe_math writes 8 values per triple again)
RTMI_EXPORT uint64_t rtmi_sample_key(uint64_t seed, uint64_t pixel, uint64_t sample) { return sample_key(seed, pixel, sample); }

RTMI_EXPORT int rtmi_init(int device, uint32_t flags, rtmi_ctx **out_ctx) {
    if (!out_ctx) return fail(RTMI_E_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RTMI_E_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RTMI_E_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RTMI_E_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
    rtmi_ctx *c = new (std::nothrow) rtmi_ctx();
    if (!c) return fail(RTMI_E_NOMEM, "out of host memory");
    c->device = device;
    c->flags = flags;
    c->cus = prop.multiProcessorCount;
    c->lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    c->hbm = prop.totalGlobalMem;
    c->arch = prop.gcnArchName;
    if (const char *e = std::getenv("RTMI_BLOCKS_PER_CU")) c->blocks_per_cu = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("RTMI_SCAN_VARIANT")) c->scan_variant = std::min(3, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("RTMI_LDS_TILE_BYTES")) c->max_lds_bytes = std::min(64 * 1024 - 64, std::max(1024, std::atoi(e)));
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(RTMI_E_DEVICE, "hipStreamCreate failed"); }
    *out_ctx = c;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_shutdown(rtmi_ctx *c) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
#ifdef RTMI_HIST
    { unsigned long long h[200]; if (hipMemcpyFromSymbol(h, HIP_SYMBOL(rtmi::g_hist), sizeof h) == hipSuccess) { for (int k = 0; k < 3; ++k) { fprintf(stderr, "HIST%d", k); for (int i = 0; i <= 64; ++i) fprintf(stderr, " %llu", h[64 * k + i]); fprintf(stderr, "\n"); } } }
#endif
    c->samples.release(); c->accum.release(); c->tiles.release(); c->tile_ids.release(); c->counters.release(); c->scratch_lin.release(); c->multi.release();
    for (hipEvent_t e : {c->ev_done, c->ev_g0, c->ev_g1}) if (e) (void)hipEventDestroy(e);
    if (c->ev_consumed) { (void)hipSetDevice(c->ev_consumed_device); (void)hipEventDestroy(c->ev_consumed); (void)hipSetDevice(c->device); }
    for (auto &e : c->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (hipEvent_t e : c->events_r) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    c->magic = 0;
    delete c;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_set_option(rtmi_ctx *c, const char *name, int64_t value) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (!name) return fail(RTMI_E_ARG, "name is NULL");
    if (!std::strcmp(name, "blocks_per_cu")) { if (value < 1 || value > 64) return fail(RTMI_E_ARG, "blocks_per_cu must be 1..64"); c->blocks_per_cu = (int)value; return RTMI_OK; }
    if (!std::strcmp(name, "workspace_bytes")) { if (value < (1 << 20)) return fail(RTMI_E_ARG, "workspace_bytes must be >= 1 MiB"); c->workspace_bytes = value; return RTMI_OK; }
    if (!std::strcmp(name, "lds_tile_bytes")) { if (value < 1024 || value > 64 * 1024 - 64) return fail(RTMI_E_ARG, "lds_tile_bytes out of range"); c->max_lds_bytes = (int)value; return RTMI_OK; }
    if (!std::strcmp(name, "timing")) { if (value) c->flags |= RTMI_FLAG_TIMING; else c->flags &= ~RTMI_FLAG_TIMING; return RTMI_OK; }
    if (!std::strcmp(name, "scan_variant")) { if (value < 0 || value > 3) return fail(RTMI_E_ARG, "scan_variant must be 0..3"); c->scan_variant = (int)value; return RTMI_OK; }
    if (!std::strcmp(name, "count_traversal")) { c->count_traversal = value ? 1 : 0; return RTMI_OK; }
    if (!std::strcmp(name, "test_fail_next_render")) { c->fail_next_render = value ? 1 : 0; return RTMI_OK; }
    if (!std::strcmp(name, "test_fail_allocs")) { c->fail_allocs = (int)std::max<int64_t>(0, std::min<int64_t>(value, 64)); return RTMI_OK; }
    if (!std::strcmp(name, "flat_below")) { if (value < 0 || value > (1 << 20)) return fail(RTMI_E_ARG, "flat_below must be 0..2^20"); c->flat_below = (int)value; return RTMI_OK; }
    if (!std::strcmp(name, "suspend_lanes")) { if (value < 0 || value > 64) return fail(RTMI_E_ARG, "suspend_lanes must be 0..64"); c->suspend_lanes = (int)value; c->suspend_lanes_set = true; return RTMI_OK; }
    if (!std::strcmp(name, "accel")) {
        if (value == RTMI_ACCEL_FLAT || value == RTMI_ACCEL_BVH) { c->accel = (int)value; return RTMI_OK; }
        return fail(RTMI_E_UNSUPPORTED, "accel %lld is not available in this build", (long long)value);
    }
    return fail(RTMI_E_ARG, "unknown option '%s'", name);
}

RTMI_EXPORT int rtmi_device_info(rtmi_ctx *c, int32_t *cus, int32_t *lds, int64_t *hbm, char *arch, int32_t arch_len) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (cus) *cus = c->cus;
    if (lds) *lds = c->lds_per_cu;
    if (hbm) *hbm = (int64_t)c->hbm;
    if (arch && arch_len > 0) { std::strncpy(arch, c->arch.c_str(), (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
    return RTMI_OK;
}

// ---- scene ---------------------------------------------------------------------------------------------
RTMI_EXPORT int rtmi_scene_create(rtmi_ctx *c, int32_t n_prims, const int32_t *prim_kind, const double *prim_geom, const int32_t *prim_mat,
                                  int32_t n_mats, const int32_t *mat_kind, const int32_t *mat_tex, const double *mat_param,
                                  int32_t n_tex, const int32_t *tex_kind, const double *tex_param, const int32_t *tex_child,
                                  int32_t cam_kind, const double *cam, rtmi_scene **out_scene) {
    return rtmi_scene_create_ex(c, n_prims, prim_kind, prim_geom, prim_mat, n_mats, mat_kind, mat_tex, mat_param, n_tex, tex_kind, tex_param, tex_child,
                                cam_kind, cam, nullptr, nullptr, 0, nullptr, nullptr, out_scene);
}

// DevScene::media_fast follows the media call sequence
static void fill_media_fast(rtmi_scene *s) {
    for (int k = 0; k < 8; ++k) {
        double *q = s->dev.media_fast[k];
        for (int j = 0; j < 8; ++j) q[j] = 0.0;
        if (k >= s->dev.n_media) continue;
        const auto it = s->media_fast_of.find(s->dev.media_idx[k]);
        if (it == s->media_fast_of.end()) continue;
        q[0] = 1.0;
        for (int j = 0; j < 5; ++j) q[1 + j] = it->second[(size_t)j];
    }
}
RTMI_EXPORT int rtmi_scene_create_ex(rtmi_ctx *c, int32_t n_prims, const int32_t *prim_kind, const double *prim_geom, const int32_t *prim_mat,
                                     int32_t n_mats, const int32_t *mat_kind, const int32_t *mat_tex, const double *mat_param,
                                     int32_t n_tex, const int32_t *tex_kind, const double *tex_param, const int32_t *tex_child,
                                     int32_t cam_kind, const double *cam, const int32_t *prim_flip, const int32_t *prim_xform,
                                     int32_t n_xforms, const int32_t *xform_kind, const double *xform_param, rtmi_scene **out_scene) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    const double t_create0 = now_ms();
    if (n_xforms < 0 || (n_xforms > 0 && (!xform_kind || !xform_param || !prim_xform))) return fail(RTMI_E_ARG, "xform arrays are NULL");
    for (int k = 0; k < n_xforms; ++k)
        if (xform_kind[k] != RTMI_XFORM_TRANSLATE && xform_kind[k] != RTMI_XFORM_ROTATE_Y) return fail(RTMI_E_UNSUPPORTED, "xform %d: kind %d unsupported on GPU path", k, xform_kind[k]);
    if (!out_scene) return fail(RTMI_E_ARG, "out_scene is NULL");
    *out_scene = nullptr;
    if (n_prims < 0 || n_mats < 0 || n_tex < 0) return fail(RTMI_E_ARG, "negative count");
    if (n_prims > 0 && (!prim_kind || !prim_geom || !prim_mat)) return fail(RTMI_E_ARG, "primitive arrays are NULL");
    if (n_mats > 0 && (!mat_kind || !mat_tex || !mat_param)) return fail(RTMI_E_ARG, "material arrays are NULL");
    if (n_tex > 0 && (!tex_kind || !tex_param || !tex_child)) return fail(RTMI_E_ARG, "texture arrays are NULL");
    if (!cam) return fail(RTMI_E_ARG, "cam is NULL");
    if (cam_kind != RTMI_CAM_PINHOLE && cam_kind != RTMI_CAM_THINLENS) return fail(RTMI_E_UNSUPPORTED, "camera kind %d unsupported on GPU path", cam_kind);
    // validate: this is where "unknown record type -> explicit unsupported error" surfaces
    for (int t = 0; t < n_tex; ++t) {
        if (tex_kind[t] < RTMI_TEX_CONSTANT || tex_kind[t] > RTMI_TEX_IMAGE) return fail(RTMI_E_UNSUPPORTED, "texture %d: kind %d unsupported on GPU path", t, tex_kind[t]);
        if (tex_kind[t] == RTMI_TEX_FLIP_U || tex_kind[t] == RTMI_TEX_FLIP_V) {
            const int ch = tex_child[2 * t];
            if (ch < 0 || ch >= n_tex || ch == t) return fail(RTMI_E_ARG, "texture %d: wrapped texture %d invalid", t, ch);
        }
        if (tex_kind[t] == RTMI_TEX_CHECKER)
            for (int k = 0; k < 2; ++k) {
                const int ch = tex_child[2 * t + k];
                if (ch < 0 || ch >= n_tex || ch == t) return fail(RTMI_E_ARG, "texture %d: checker child %d invalid", t, ch);
            }
    }
    for (int m = 0; m < n_mats; ++m) {
        if (mat_kind[m] < RTMI_MAT_LAMBERTIAN || mat_kind[m] > RTMI_MAT_ISOTROPIC) return fail(RTMI_E_UNSUPPORTED, "material %d: kind %d unsupported on GPU path", m, mat_kind[m]);
        if (mat_kind[m] != RTMI_MAT_DIELECTRIC && (mat_tex[m] < 0 || mat_tex[m] >= n_tex)) return fail(RTMI_E_ARG, "material %d: texture index %d invalid", m, mat_tex[m]);
    }
    std::vector<double> stat_geom, mov_geom, stat4_d;
    std::vector<float> stat4_f;
    bool has_ext = false, uses_perlin = false;
    int n_world = 0, n_media = 0, media[16];
    int max_image = -1;
    for (int t = 0; t < n_tex; ++t) {
        if (tex_kind[t] > RTMI_TEX_CHECKER) has_ext = true; // section 8(f4) textures live in the EXT kernels only
        if (tex_kind[t] >= RTMI_TEX_PERLIN_NOISE && tex_kind[t] <= RTMI_TEX_MARBLE) uses_perlin = true;
        if (tex_kind[t] == RTMI_TEX_IMAGE) {
            const double im = tex_param[(size_t)t * RTMI_TEX_STRIDE];
            if (!(im >= 0 && im < 1e6 && im == std::floor(im))) return fail(RTMI_E_ARG, "texture %d: image index invalid", t);
            max_image = std::max(max_image, (int)im);
        }
    }
    std::vector<int> stat_orig, mov_orig, pk((size_t)n_prims), pm((size_t)n_prims);
    for (int i = 0; i < n_prims; ++i) {
        const int kind = prim_kind[i] & ~RTMI_PRIM_BOUNDARY;
        const bool is_boundary = (prim_kind[i] & RTMI_PRIM_BOUNDARY) != 0;
        if (kind < RTMI_PRIM_SPHERE || kind > RTMI_PRIM_MEDIUM) return fail(RTMI_E_UNSUPPORTED, "primitive %d: kind %d unsupported on GPU path", i, prim_kind[i]);
        if (is_boundary && kind == RTMI_PRIM_MEDIUM) return fail(RTMI_E_UNSUPPORTED, "primitive %d: a medium inside a medium's boundary is unsupported", i);
        if (!is_boundary && n_world != i) return fail(RTMI_E_ARG, "primitive %d: boundary primitives must come after all world primitives", i);
        if (!is_boundary) n_world = i + 1;
        if (prim_mat[i] < 0 || prim_mat[i] >= n_mats) return fail(RTMI_E_ARG, "primitive %d: material index %d invalid", i, prim_mat[i]);
        pk[(size_t)i] = kind; pm[(size_t)i] = prim_mat[i];
        if (mat_kind[prim_mat[i]] == RTMI_MAT_ISOTROPIC) has_ext = true; // Isotropic.scatter (shader.clj:129-138) is compiled into the EXT kernels only
        if (kind == RTMI_PRIM_MEDIUM) {
            const double *mg = prim_geom + (size_t)i * RTMI_PRIM_STRIDE;
            const int fb = (int)mg[1], nb = (int)mg[2];
            if (!(mg[0] == mg[0]) || fb < 0 || nb <= 0 || fb + nb > n_prims) return fail(RTMI_E_ARG, "medium %d: boundary range [%d, %d) invalid", i, fb, fb + nb);
            for (int q = fb; q < fb + nb; ++q) if (!(prim_kind[q] & RTMI_PRIM_BOUNDARY)) return fail(RTMI_E_ARG, "medium %d: primitive %d is not flagged RTMI_PRIM_BOUNDARY", i, q);
            if (mat_kind[prim_mat[i]] != RTMI_MAT_ISOTROPIC) return fail(RTMI_E_ARG, "medium %d: the phase function must be RTMI_MAT_ISOTROPIC", i);
            if (n_media >= 16) return fail(RTMI_E_UNSUPPORTED, "more than 16 ConstantMedium records in one scene");
            media[n_media++] = i;
            has_ext = true;
            continue;
        }
        if (is_boundary) { has_ext = true; continue; }
        const double *g = prim_geom + (size_t)i * RTMI_PRIM_STRIDE;
        const int xf_first = prim_xform ? prim_xform[2 * i] : 0, xf_count = prim_xform ? prim_xform[2 * i + 1] : 0;
        if (xf_count < 0 || xf_first < 0 || xf_first + xf_count > n_xforms) return fail(RTMI_E_ARG, "primitive %d: xform range [%d, %d) invalid", i, xf_first, xf_first + xf_count);
        if (kind > RTMI_PRIM_MOVING || xf_count > 0 || (prim_flip && prim_flip[i])) { has_ext = true; continue; }
        if (kind == RTMI_PRIM_MOVING) {
            mov_geom.insert(mov_geom.end(), g, g + RTMI_PRIM_STRIDE);
            mov_orig.push_back(i);
        } else {
            stat_geom.insert(stat_geom.end(), g, g + 4);
            stat_orig.push_back(i);
            // {cx, cy, cz, r*r}: hitable.clj:188 (* radius radius), one IEEE multiply in the precision the kernel computes in
            const volatile double r2d = g[3] * g[3];
            const volatile float rf = (float)g[3];
            const volatile float r2f = rf * rf;
            stat4_d.insert(stat4_d.end(), {g[0], g[1], g[2], (double)r2d});
            stat4_f.insert(stat4_f.end(), {(float)g[0], (float)g[1], (float)g[2], (float)r2f});
        }
    }
    HIP_TRY(hipSetDevice(c->device));
    rtmi_scene *s = new (std::nothrow) rtmi_scene();
    if (!s) return fail(RTMI_E_NOMEM, "out of host memory");
    s->ctx = c; s->n_prims = n_prims; s->n_mats = n_mats; s->n_tex = n_tex;
    s->uses_perlin = uses_perlin; s->max_image = max_image;
    DevScene &d = s->dev;
    d.n_static = (int)stat_orig.size(); d.n_moving = (int)mov_orig.size(); d.n_tex = n_tex; d.cam_kind = cam_kind;
    std::memcpy(d.cam, cam, 24 * sizeof(double));
    { // get-ray's origin is cam origin + lens offset; with aperture 0 the offset is (+-0, +-0, +-0) (camera.clj:39-44: lens-radius * rand-in-unit-disk), and
      // x + (+-0) = x bit for bit for every x except -0 (whose sum with +0 is +0): then, and for the pinhole camera, all rays share one origin
        bool fixed = cam_kind == RTMI_CAM_PINHOLE || cam[21] == 0.0;
        for (int k = 0; k < 3; ++k) fixed = fixed && !(cam[k] == 0.0 && std::signbit(cam[k])) && std::isfinite(cam[k]);
        for (int k = 12; k < 18; ++k) fixed = fixed && std::isfinite(cam[k]);
        d.cam_fixed_origin = fixed ? 1 : 0;
    }
    std::vector<int> mk(mat_kind, mat_kind + n_mats), mt(mat_tex, mat_tex + n_mats), tk(tex_kind, tex_kind + n_tex), tc(tex_child, tex_child + 2 * (size_t)n_tex);
    std::vector<double> mp(mat_param, mat_param + n_mats), tpv(tex_param, tex_param + (size_t)n_tex * RTMI_TEX_STRIDE);
    int rc = RTMI_OK;
    if (!rc) rc = upload(s, stat_geom, &d.stat_geom);
    if (!rc) rc = upload(s, stat_orig, &d.stat_orig);
    if (!stat_orig.empty()) { // pad to round_up(n,8)+8 records with copies of the last sphere (see scan_static_pipe)
        const size_t n4 = (stat_orig.size() + 7) / 8 * 8 + 8;
        const double ld[4] = {stat4_d[stat4_d.size() - 4], stat4_d[stat4_d.size() - 3], stat4_d[stat4_d.size() - 2], stat4_d[stat4_d.size() - 1]};
        const float lf[4] = {stat4_f[stat4_f.size() - 4], stat4_f[stat4_f.size() - 3], stat4_f[stat4_f.size() - 2], stat4_f[stat4_f.size() - 1]};
        while (stat4_d.size() < n4 * 4) stat4_d.insert(stat4_d.end(), ld, ld + 4);
        while (stat4_f.size() < n4 * 4) stat4_f.insert(stat4_f.end(), lf, lf + 4);
    }
    // ---- scan variant SCAN_SGPR_CULL: all primitives in Hitlist order ----
    // exact12[i] = c0.xyz, r*r, c1.xyz, t0, t1, moving?, r, 0 ; cull20 = FP32 bounding data per group of 4.
    // A MovingSphere's cull entry bounds its sweep over the camera's shutter interval [t_lo, t_hi] (rays outside that
    // interval bypass the cull, make_cull_ray): centre = midpoint of the two extreme centres, radius = r + half the
    // distance between them, both inflated for the float rounding of the centre.
    const double t_lo = cam_kind == RTMI_CAM_THINLENS ? std::min(cam[22], cam[23]) : 0.0;
    const double t_hi = cam_kind == RTMI_CAM_THINLENS ? std::max(cam[22], cam[23]) : 0.0;
    std::vector<double> exact12;
    std::vector<float> cull_c, cull_r2, cull_w; // per primitive: centre (3), r2, w
    std::vector<BvhBox> wbox((size_t)n_prims);
    std::vector<char> bounded((size_t)n_prims, 0);
    std::vector<int> ext_info;
    std::vector<double> ext_xf;
    for (int k = 0; k < n_xforms; ++k) {
        const double *p = xform_param + (size_t)k * 3;
        const double rec[4] = {xform_kind[k] == RTMI_XFORM_TRANSLATE ? 0.0 : 1.0, p[0], p[1], p[2]};
        ext_xf.insert(ext_xf.end(), rec, rec + 4);
    }
    for (int i = 0; i < n_prims; ++i) {
        const double *g = prim_geom + (size_t)i * RTMI_PRIM_STRIDE;
        const int kind = prim_kind[i] & ~RTMI_PRIM_BOUNDARY;
        const int xf_first = prim_xform ? prim_xform[2 * i] : 0, xf_count = prim_xform ? prim_xform[2 * i + 1] : 0;
        const int info[4] = {kind, prim_flip ? (prim_flip[i] & 1) : 0, xf_first, xf_count};
        ext_info.insert(ext_info.end(), info, info + 4);
        if (kind == RTMI_PRIM_MEDIUM) { // not a surface: no box, a neutral cull entry (ext_prim_test ignores it), evaluated by ext_medium_test
            exact12.insert(exact12.end(), {g[0], g[1], g[2], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0});
            cull_c.insert(cull_c.end(), {0.0f, 0.0f, 0.0f});
            cull_r2.push_back(0.0f);
            cull_w.push_back(0.0f);
            continue;
        }
        bounded[(size_t)i] = prim_world_box(kind, g, xform_kind, xform_param, xf_first, xf_count, t_lo, t_hi, wbox[(size_t)i]);
        const bool moving = kind == RTMI_PRIM_MOVING;
        const volatile double r2d = g[3] * g[3];
        if (kind > RTMI_PRIM_MOVING || xf_count > 0) { // f3 primitive or instanced sphere: cull by the sphere around its world box
            if (kind > RTMI_PRIM_MOVING) exact12.insert(exact12.end(), {g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], 0.0, 0.0, 0.0});
            else exact12.insert(exact12.end(), {g[0], g[1], g[2], (double)r2d, g[4], g[5], g[6], g[7], g[8], moving ? 1.0 : 0.0, g[3], 0.0});
            float cf[3] = {0, 0, 0};
            double r2b = 3.0e38, w = 3.0e38;
            if (bounded[(size_t)i]) {
                const BvhBox &b = wbox[(size_t)i];
                double slack = 0.0, rb2 = 0.0, cn = 0.0;
                for (int k = 0; k < 3; ++k) {
                    const double cm = 0.5 * (b.lo[k] + b.hi[k]);
                    cf[k] = (float)cm;
                    slack += std::fabs(cm - (double)cf[k]);
                    rb2 += 0.25 * (b.hi[k] - b.lo[k]) * (b.hi[k] - b.lo[k]);
                    cn += std::fabs((double)cf[k]);
                }
                const double rb = (std::sqrt(rb2) + slack + 1e-4) * (1.0 + 1e-6); // 1e-4: the reference's own rect/triangle bbox padding scale
                r2b = rb * rb * (1.0 + 1e-6);
                w = (2.0 * (cn + slack) * (cn + slack) + r2b) * 1.0001;
                if (!std::isfinite(w) || w > 1e37) { w = 3.0e38; r2b = 3.0e38; }
            }
            cull_c.insert(cull_c.end(), cf, cf + 3);
            cull_r2.push_back((float)std::min(r2b * (1.0 + 1e-6), 3.0e38));
            cull_w.push_back((float)std::min(w, 3.0e38));
            continue;
        }
        const double rec[12] = {g[0], g[1], g[2], (double)r2d, g[4], g[5], g[6], g[7], g[8], moving ? 1.0 : 0.0, g[3], 0.0};
        exact12.insert(exact12.end(), rec, rec + 12);
        double cm[3] = {g[0], g[1], g[2]}, rb = std::fabs(g[3]);
        bool unbounded = false;
        if (moving) {
            const double f0 = (t_lo - g[7]) / (g[8] - g[7]), f1 = (t_hi - g[7]) / (g[8] - g[7]);
            if (!std::isfinite(f0) || !std::isfinite(f1)) unbounded = true;
            else {
                double half2 = 0.0;
                for (int k = 0; k < 3; ++k) {
                    const double a0 = g[k] * (1.0 - f0) + g[4 + k] * f0, a1 = g[k] * (1.0 - f1) + g[4 + k] * f1;
                    cm[k] = 0.5 * (a0 + a1);
                    half2 += 0.25 * (a1 - a0) * (a1 - a0);
                }
                rb += std::sqrt(half2) * (1.0 + 1e-9);
            }
        }
        float cf[3];
        double slack = 0.0;
        for (int k = 0; k < 3; ++k) { cf[k] = (float)cm[k]; slack += std::fabs(cm[k] - (double)cf[k]); }
        if (moving) rb = (rb + slack) * (1.0 + 1e-6); // the bounding sphere is defined around the FLOAT centre
        const double r2b = moving ? rb * rb * (1.0 + 1e-6) : (double)r2d;
        const double cn = std::fabs((double)cf[0]) + std::fabs((double)cf[1]) + std::fabs((double)cf[2]) + slack;
        double w = (2.0 * cn * cn + r2b) * 1.0001;
        if (unbounded || !std::isfinite(w) || w > 1e37) w = 3.0e38; // tol = inf: always passes to the exact test
        cull_c.insert(cull_c.end(), cf, cf + 3);
        cull_r2.push_back(unbounded ? 3.0e38f : (float)std::min(r2b * (moving ? 1.0 + 1e-6 : 1.0), 3.0e38));
        cull_w.push_back((float)w);
    }
    // Box = six consecutive rectangles RectXY z1, RectXY z0, RectXZ y1, RectXZ y0, RectYZ x1, RectYZ x0 over one (x0 y0 z0) - (x1 y1 z1) and one instance
    // chain (hitable.clj:500-511, spliced in by the flattener): the tree gets one leaf for the six (ext_box_test); z0 goes to slot 5 of the first record
    std::vector<char> box_first((size_t)std::max(n_prims, 1), 0);
    if (const char *e = std::getenv("RTMI_BOX_LEAF"); e && e[0] == '1') // (measured: make-final 22.1 ms with box leaves against 21.2 without -- six face tests per leaf cost more than the 1.8 node visits they save; kept for experiments)
    for (int i = 0; i + 5 < n_world; ++i) {
        static const int want[6] = {RTMI_PRIM_RECT_XY, RTMI_PRIM_RECT_XY, RTMI_PRIM_RECT_XZ, RTMI_PRIM_RECT_XZ, RTMI_PRIM_RECT_YZ, RTMI_PRIM_RECT_YZ};
        bool ok = true;
        for (int k = 0; k < 6 && ok; ++k) {
            ok = pk[(size_t)i + k] == want[k];
            if (prim_xform) ok = ok && prim_xform[2 * (i + k)] == prim_xform[2 * i] && prim_xform[2 * (i + k) + 1] == prim_xform[2 * i + 1];
        }
        if (!ok) continue;
        const double *q = prim_geom + (size_t)i * RTMI_PRIM_STRIDE;
        const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], z1 = q[4], z0 = q[RTMI_PRIM_STRIDE + 4];
        const double expect[6][5] = {{x0, y0, x1, y1, z1}, {x0, y0, x1, y1, z0}, {x0, z0, x1, z1, y1}, {x0, z0, x1, z1, y0}, {y0, z0, y1, z1, x1}, {y0, z0, y1, z1, x0}};
        for (int k = 0; k < 6 && ok; ++k) for (int c = 0; c < 5; ++c) ok = ok && std::memcmp(&q[(size_t)k * RTMI_PRIM_STRIDE + c], &expect[k][c], sizeof(double)) == 0; // bit for bit
        if (!ok) continue;
        box_first[(size_t)i] = 1;
        exact12[(size_t)i * 12 + 5] = z0;
        i += 5;
    }
    const size_t n_pad = n_prims > 0 ? ((size_t)n_prims + 7) / 8 * 8 + 8 : 0;
    if (n_prims > 0) {
        const std::vector<double> last12(exact12.begin() + (size_t)(n_prims - 1) * 12, exact12.begin() + (size_t)n_prims * 12);
        exact12.resize((size_t)n_prims * 12);
        for (size_t i = (size_t)n_prims; i < n_pad; ++i) {
            exact12.insert(exact12.end(), last12.begin(), last12.end());
            cull_c.push_back(cull_c[(size_t)(n_prims - 1) * 3]); cull_c.push_back(cull_c[(size_t)(n_prims - 1) * 3 + 1]); cull_c.push_back(cull_c[(size_t)(n_prims - 1) * 3 + 2]);
            cull_r2.push_back(cull_r2[(size_t)n_prims - 1]);
            cull_w.push_back(cull_w[(size_t)n_prims - 1]);
        }
    }
    std::vector<float> cull20; // per group of 4 (padded) primitives: cx[4] cy[4] cz[4] r2[4] w[4]
    for (size_t g = 0; g + 3 < n_pad; g += 4) {
        float rec[20];
        for (int k = 0; k < 4; ++k) {
            rec[k] = cull_c[(g + k) * 3]; rec[4 + k] = cull_c[(g + k) * 3 + 1]; rec[8 + k] = cull_c[(g + k) * 3 + 2];
            rec[12 + k] = cull_r2[g + k]; rec[16 + k] = cull_w[g + k];
        }
        cull20.insert(cull20.end(), rec, rec + 20);
    }
    d.n_all = n_world; d.cull_t_lo = t_lo; d.cull_t_hi = t_hi; // the scans walk the world; boundary primitives are reached only through their medium
    d.n_media = n_media;
    for (int k = 0; k < n_media; ++k) { d.media_idx[k] = media[k]; d.media_lo[k] = media[k]; }
    s->host_kind = pk;
    std::vector<int> grid_cells;
    const double t_create1 = now_ms();
    std::vector<BvhBox> media_boxes; // per ConstantMedium: the box of its boundary (if every boundary primitive can be bounded)
    for (int k = 0; k < n_media; ++k) {
        const double *mg = prim_geom + (size_t)media[k] * RTMI_PRIM_STRIDE;
        const int fb = (int)mg[1], nb = (int)mg[2];
        BvhBox u = box_empty();
        bool ok = true;
        for (int q = fb; q < fb + nb; ++q) { ok = ok && bounded[(size_t)q]; if (ok) box_grow(u, wbox[(size_t)q]); }
        if (ok) media_boxes.push_back(u);
    }
    const std::vector<float> bvh_nodes = build_bvh(d, n_world, pk.data(), wbox, bounded, cam, !has_ext, grid_cells, box_first, &s->bvh_depth, &media_boxes);
    s->bvh_node_count = (int)(bvh_nodes.size() / (d.bvh_node16 ? 8 : 16));
    if (std::getenv("RTMI_DEBUG"))
        fprintf(stderr, "[rtmi] tree: %d node records of %d bytes (%.2f MB), depth %d, %d big primitives, %d box leaves; entry grid %d x %d cells, %zu rectangle trees\n", s->bvh_node_count,
                d.bvh_node16 ? 32 : 64, s->bvh_node_count * (d.bvh_node16 ? 32.0 : 64.0) / 1e6, s->bvh_depth, d.n_big, (int)std::count(box_first.begin(), box_first.end(), (char)1), d.grid_n, d.grid_n, grid_cells.size());
    if (std::getenv("RTMI_DEBUG") && d.n_mloc) fprintf(stderr, "[rtmi] %d medium neighbourhood tree(s)\n", d.n_mloc);
    const double t_create2 = now_ms();
    if (!rc) rc = upload(s, bvh_nodes, &d.bvh_nodes);
    if (!rc) rc = upload(s, grid_cells, &d.grid_cells);
    std::vector<int> moving_all;
    for (int i = 0; i < n_world; ++i) if (pk[(size_t)i] == RTMI_PRIM_MOVING) moving_all.push_back(i);
    d.n_moving_all = (int)moving_all.size();
    if (!rc) rc = upload(s, moving_all, &d.moving_all);
    d.has_ext = has_ext ? 1 : 0;
    { const char *e = std::getenv("RTMI_SMALL_SCAN"); d.small_scan = (has_ext && n_world <= RTMI_SMALL_SCAN_MAX && !(e && e[0] == '0')) ? 1 : 0; }
    { // LeafRec (rtmi_device.h: ext_leaf_test): one 112-byte record per world primitive
        std::vector<double> leaf_rec((size_t)std::max(n_prims, 1) * RTMI_LEAF_REC_DOUBLES, 0.0);
        for (int i = 0; i < n_prims; ++i) {
            double *q = &leaf_rec[(size_t)i * RTMI_LEAF_REC_DOUBLES];
            const int kind = pk[(size_t)i];
            const int xf_first = prim_xform ? prim_xform[2 * i] : 0, xf_count = prim_xform ? prim_xform[2 * i + 1] : 0;
            int hdr[4] = {kind | ((prim_flip && (prim_flip[i] & 1)) ? 0x100 : 0), 0, 0, 0}; // bit 8: FlipNormals parity (resolve_hit_ext)
            const bool simple = (kind == RTMI_PRIM_SPHERE || kind == RTMI_PRIM_UVSPHERE || (kind >= RTMI_PRIM_RECT_XY && kind <= RTMI_PRIM_RECT_YZ)) && xf_count <= 2;
            if (!simple) hdr[1] = 1; // generic: ext_prim_test
            else {
                for (int c = 0; c < 5; ++c) q[2 + c] = exact12[(size_t)i * 12 + c]; // sphere: c r*r (slot 4 unused) | rectangle: u0 v0 u1 v1 k
                for (int k = 0; k < xf_count; ++k) {
                    const double *xp = xform_param + (size_t)(xf_first + k) * 3;
                    hdr[2 + k] = xform_kind[xf_first + k] == RTMI_XFORM_TRANSLATE ? 1 : 2;
                    q[7 + 3 * k] = xp[0]; q[8 + 3 * k] = xp[1]; q[9 + 3 * k] = xp[2];
                }
                if (box_first[(size_t)i]) q[13] = exact12[(size_t)i * 12 + 5]; // z0 of the Box whose first face this rectangle is
            }
            std::memcpy(q, hdr, sizeof hdr);
        }
        if (!rc) rc = upload(s, leaf_rec, &d.leaf_rec);
    }
    ext_info.insert(ext_info.end(), {RTMI_PRIM_MEDIUM, 0, 0, 0}); // one record past the end: scan_small_ext requests primitive i + 1's records while it tests primitive i
    if (!rc) rc = upload(s, ext_info, &d.ext_info);
    if (!rc) rc = upload(s, ext_xf, &d.ext_xf);
    if (!rc) rc = upload(s, cull20, &d.cull20);
    if (!rc) rc = upload(s, exact12, &d.exact12);
    if (!rc) rc = upload(s, stat4_d, &d.stat4_d);
    if (!rc) rc = upload(s, stat4_f, &d.stat4_f);
    if (!rc) rc = upload(s, mov_geom, &d.mov_geom);
    if (!rc) rc = upload(s, mov_orig, &d.mov_orig);
    { // device copy of prim_kind: + RTMI_PRIM_NEEDS_U / _V where a UVSphere's material texture reads that coordinate (texture.clj: UVGradient --
      // per coordinate: a gradient whose corner colours do not vary along u never reads u --, ImageMap, through Checkerboard / FlipTexture
      // children); Constant, Checkerboard itself and the Perlin family read p only
        std::vector<char> uses((size_t)std::max(n_tex, 1), 0); // bit 0: reads u, bit 1: reads v
        bool changed = true;
        for (int pass = 0; pass <= n_tex && changed; ++pass) { // children may come after their parents: iterate to the fixed point (a pass that changes nothing ends it:
            changed = false;                                  // one texture per sphere made the unconditional n_tex passes 5.8 of the 6.4 s a 90 000-sphere scene took to create)
            for (int t = 0; t < n_tex; ++t) {
                const int k = tex_kind[t];
                char u = k == RTMI_TEX_IMAGE ? 3 : 0;
                if (k == RTMI_TEX_UVGRADIENT) { // co cu cv cuv: a = cu (1-u) + co u, b = cuv (1-u) + cv u, out = b (1-v) + a v  (texture.clj:26-34)
                    const double *tp = tex_param + (size_t)t * RTMI_TEX_STRIDE;
                    bool var_u = false, var_v = false;
                    // "does not vary" is a comparison of BITS (memcmp), not of values: only then is the lerp of the two colours at u = 1/2 the colour itself
                    // whatever it holds (c/2 + c/2 = c exactly; +0 against -0, or two different NaNs, count as varying and keep the real coordinate)
                    auto same = [&](int a, int b) { return std::memcmp(&tp[a], &tp[b], sizeof(double)) == 0; };
                    for (int c = 0; c < 3; ++c) {
                        var_u = var_u || !same(c, 3 + c) || !same(6 + c, 9 + c);     // co != cu or cv != cuv
                        var_v = var_v || !same(c, 6 + c) || !same(3 + c, 9 + c);     // co != cv or cu != cuv
                    }
                    u = (char)((var_u ? 1 : 0) | (var_v ? 2 : 0));
                }
                if (k == RTMI_TEX_CHECKER || k == RTMI_TEX_FLIP_U || k == RTMI_TEX_FLIP_V)
                    for (int c = 0; c < (k == RTMI_TEX_CHECKER ? 2 : 1); ++c) {
                        const int ch = tex_child[2 * (size_t)t + c];
                        if (ch >= 0 && ch < n_tex) u |= uses[(size_t)ch];
                    }
                if (uses[(size_t)t] != u) { uses[(size_t)t] = u; changed = true; }
            }
        }
        std::vector<int> pk_dev(pk);
        for (int i = 0; i < n_prims; ++i)
            if (pk[(size_t)i] == RTMI_PRIM_UVSPHERE) {
                const int m = pm[(size_t)i], t = (m >= 0 && m < n_mats) ? mat_tex[m] : -1;
                const int bits = (t < 0 || t >= n_tex) ? 3 : uses[(size_t)t];
                if (bits & 1) pk_dev[(size_t)i] |= RTMI_PRIM_NEEDS_U;
                if (bits & 2) pk_dev[(size_t)i] |= RTMI_PRIM_NEEDS_V;
            } else if (pk[(size_t)i] >= RTMI_PRIM_RECT_XY && pk[(size_t)i] <= RTMI_PRIM_TRIANGLE) {
                // a rectangle's uv is two IEEE divisions per hit (hitable.clj:283-284), a triangle's a second Moeller-Trumbore: computed only where the material's
                // texture reads uv at all (both coordinates then: no coordinate is ever replaced here, so nothing deviates) -- a Cornell box's walls never do
                const int m = pm[(size_t)i], t = (m >= 0 && m < n_mats) ? mat_tex[m] : -1;
                if (t < 0 || t >= n_tex || uses[(size_t)t]) pk_dev[(size_t)i] |= RTMI_PRIM_NEEDS_UV;
            }
        if (!rc) rc = upload(s, pk_dev, &d.prim_kind);
        std::vector<int> km((size_t)std::max(n_prims, 1) * 2, 0);
        for (int i = 0; i < n_prims; ++i) { km[2 * (size_t)i] = pk_dev[(size_t)i]; km[2 * (size_t)i + 1] = pm[(size_t)i]; }
        if (!rc) rc = upload(s, km, &d.prim_km);
        std::vector<double> mrec((size_t)std::max(n_mats, 1) * 12, 0.0), mgrad((size_t)std::max(n_mats, 1) * 12, 0.0);
        for (int m = 0; m < n_mats; ++m) {
            MatRec r;
            std::memset(&r, 0, sizeof(r));
            r.mat_kind = mat_kind[m]; r.tex = mat_tex[m]; r.param = mat_param[m];
            if (mat_kind[m] == RTMI_MAT_DIELECTRIC) { // one IEEE operation each, as the kernel would evaluate them per scatter
                const volatile double ri = mat_param[m];
                const volatile double inv = 1.0 / ri, num = 1.0 - ri, den = 1.0 + ri;
                const volatile double q = num / den;
                const volatile double r0 = q * q;
                r.inv_ri = inv; r.r0 = r0;
            }
            r.tex_kind = (r.tex >= 0 && r.tex < n_tex) ? tex_kind[r.tex] : -1;
            if (r.tex_kind == RTMI_TEX_CONSTANT) { const double *tp = tex_param + (size_t)r.tex * RTMI_TEX_STRIDE; r.r = tp[0]; r.g = tp[1]; r.b = tp[2]; }
            if (r.tex_kind == RTMI_TEX_UVGRADIENT) { // texture.clj:26-34: co cu cv cuv travel with the material
                std::memcpy(&mgrad[(size_t)m * 12], tex_param + (size_t)r.tex * RTMI_TEX_STRIDE, 12 * sizeof(double));
                r.tex_kind = RTMI_TEX_GRADIENT_REC;
            }
            if (r.tex_kind == RTMI_TEX_CHECKER) { // both children Constant: the whole texture fits the record
                const int c0 = tex_child[2 * (size_t)r.tex], c1 = tex_child[2 * (size_t)r.tex + 1];
                if (c0 >= 0 && c0 < n_tex && c1 >= 0 && c1 < n_tex && tex_kind[c0] == RTMI_TEX_CONSTANT && tex_kind[c1] == RTMI_TEX_CONSTANT) {
                    const double *t0 = tex_param + (size_t)c0 * RTMI_TEX_STRIDE, *t1 = tex_param + (size_t)c1 * RTMI_TEX_STRIDE;
                    r.tex_kind = RTMI_TEX_CHECKER2;
                    r.scale = tex_param[(size_t)r.tex * RTMI_TEX_STRIDE];
                    r.r = t0[0]; r.g = t0[1]; r.b = t0[2]; r.c1r = t1[0]; r.c1g = t1[1]; r.c1b = t1[2];
                }
            }
            static_assert(sizeof(MatRec) == 96, "MatRec is twelve doubles");
            std::memcpy(&mrec[(size_t)m * 12], &r, sizeof(r));
        }
        if (!rc) rc = upload(s, mrec, &d.mat_rec);
        if (!rc) rc = upload(s, mgrad, &d.mat_grad);
    }
    if (!rc) rc = upload(s, pm, &d.prim_mat);
    if (!rc) rc = upload(s, mk, &d.mat_kind);
    if (!rc) rc = upload(s, mt, &d.mat_tex);
    if (!rc) rc = upload(s, mp, &d.mat_param);
    if (!rc) rc = upload(s, tk, &d.tex_kind);
    if (!rc) rc = upload(s, tpv, &d.tex_param);
    if (!rc) rc = upload(s, tc, &d.tex_child);
    for (int k = 0; k < n_media; ++k) { // media whose boundary is one plain sphere, neither under wrappers: their operands go into the descriptor (media_fast)
        const int m = media[k];
        const int fb = (int)prim_geom[(size_t)m * RTMI_PRIM_STRIDE + 1], nb = (int)prim_geom[(size_t)m * RTMI_PRIM_STRIDE + 2];
        if (nb != 1 || fb < 0 || fb >= n_prims) continue;
        if (pk[(size_t)fb] != RTMI_PRIM_SPHERE && pk[(size_t)fb] != RTMI_PRIM_UVSPHERE) continue;
        if (prim_xform && (prim_xform[2 * m + 1] != 0 || prim_xform[2 * fb + 1] != 0)) continue;
        s->media_fast_of[m] = {exact12[(size_t)m * 12], exact12[(size_t)fb * 12], exact12[(size_t)fb * 12 + 1], exact12[(size_t)fb * 12 + 2], exact12[(size_t)fb * 12 + 3]};
    }
    fill_media_fast(s);
    if (!rc) {
        std::vector<DevScene> one(1, d);
        const DevScene *dp = nullptr;
        rc = upload(s, one, &dp);
        s->d_dev = (ScenePtr)dp;
    }
    if (rc) { rtmi_scene_destroy(s); return rc; }
    if (std::getenv("RTMI_DEBUG"))
        fprintf(stderr, "[rtmi] scene create: records %.2f ms, trees %.2f ms, tables + upload (%zu allocations, %.2f MB) %.2f ms\n", t_create1 - t_create0, t_create2 - t_create1,
                s->allocs.size(), (double)s->device_bytes / 1e6, now_ms() - t_create2);
    {
        rtmi_scene::Args &A = s->args;
        A.prim_kind.assign(prim_kind, prim_kind + n_prims); A.prim_mat.assign(prim_mat, prim_mat + n_prims);
        A.prim_geom.assign(prim_geom, prim_geom + (size_t)n_prims * RTMI_PRIM_STRIDE);
        A.mat_kind.assign(mat_kind, mat_kind + n_mats); A.mat_tex.assign(mat_tex, mat_tex + n_mats); A.mat_param.assign(mat_param, mat_param + n_mats);
        A.tex_kind.assign(tex_kind, tex_kind + n_tex); A.tex_param.assign(tex_param, tex_param + (size_t)n_tex * RTMI_TEX_STRIDE);
        A.tex_child.assign(tex_child, tex_child + 2 * (size_t)n_tex);
        A.cam.assign(cam, cam + 24); A.cam_kind = cam_kind;
        if (prim_flip) A.prim_flip.assign(prim_flip, prim_flip + n_prims);
        if (prim_xform) A.prim_xform.assign(prim_xform, prim_xform + 2 * (size_t)n_prims);
        if (n_xforms > 0) { A.xform_kind.assign(xform_kind, xform_kind + n_xforms); A.xform_param.assign(xform_param, xform_param + 3 * (size_t)n_xforms); }
    }
    *out_scene = s;
    return RTMI_OK;
}

namespace {
int reupload_descriptor(rtmi_scene *s) {
    fill_media_fast(s);
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    HIP_TRY(hipMemcpy((void *)s->d_dev, &s->dev, sizeof(DevScene), hipMemcpyHostToDevice));
    return RTMI_OK;
}
} // namespace

RTMI_EXPORT int rtmi_scene_set_perlin(rtmi_scene *s, const double *vectors, const int32_t *perm) {
    if (!scene_ok(s)) return fail(RTMI_E_STATE, "invalid scene handle");
    if (!vectors || !perm) return fail(RTMI_E_ARG, "NULL array");
    for (int a = 0; a < 3; ++a) { // each must be a permutation of 0..255 (perlin.clj:10-17)
        bool seen[256] = {false};
        for (int k = 0; k < 256; ++k) {
            const int v = perm[a * 256 + k];
            if (v < 0 || v > 255 || seen[v]) return fail(RTMI_E_ARG, "perm-%c is not a permutation of 0..255", "xyz"[a]);
            seen[v] = true;
        }
    }
    HIP_TRY(hipSetDevice(s->ctx->device));
    std::vector<double> v(vectors, vectors + 768);
    std::vector<int> p(perm, perm + 768);
    int rc = upload(s, v, &s->dev.perlin_vec);
    if (!rc) rc = upload(s, p, &s->dev.perlin_perm);
    if (rc) return rc;
    s->have_perlin = true;
    s->args.perlin_vec = v; s->args.perm.assign(perm, perm + 768);
    return reupload_descriptor(s);
}

RTMI_EXPORT int rtmi_scene_set_media_calls(rtmi_scene *s, int32_t n_calls, const int32_t *calls) {
    if (!scene_ok(s)) return fail(RTMI_E_STATE, "invalid scene handle");
    if (n_calls < 0 || n_calls > 32 || (n_calls > 0 && !calls)) return fail(RTMI_E_ARG, "n_calls must be 0..32");
    for (int k = 0; k < n_calls; ++k)
        if (calls[k] < 0 || calls[k] >= s->n_prims || s->host_kind[(size_t)calls[k]] != RTMI_PRIM_MEDIUM) return fail(RTMI_E_ARG, "calls[%d] = %d is not a medium primitive", k, calls[k]);
    if (s->dev.media_seq == 1)
        for (int k = 1; k < n_calls; ++k) if (calls[k] <= calls[k - 1]) return fail(RTMI_E_ARG, "RTMI_MEDIA_HITLIST: the media must be called once each, in ascending primitive (= list) order");
    if (s->dev.media_seq == 2) { s->dev.media_seq = 0; s->args.media_mode = 0; } // a plain call sequence replaces a narrowed one
    s->dev.n_media = n_calls;
    for (int k = 0; k < n_calls; ++k) { s->dev.media_idx[k] = calls[k]; s->dev.media_lo[k] = calls[k]; }
    s->args.media_calls.assign(calls, calls + n_calls); s->args.has_media_calls = true;
    HIP_TRY(hipSetDevice(s->ctx->device));
    return reupload_descriptor(s);
}

// RTMI_MEDIA_NARROWED: the media call sequence with, per call, the first primitive of the Hitlist items that narrow its t-max (narrow_from[k] = calls[k]: none)
RTMI_EXPORT int rtmi_scene_set_media_calls_narrowed(rtmi_scene *s, int32_t n_calls, const int32_t *calls, const int32_t *narrow_from) {
    if (!scene_ok(s)) return fail(RTMI_E_STATE, "invalid scene handle");
    if (n_calls < 0 || n_calls > 32 || (n_calls > 0 && (!calls || !narrow_from))) return fail(RTMI_E_ARG, "n_calls must be 0..32");
    for (int k = 0; k < n_calls; ++k) {
        if (calls[k] < 0 || calls[k] >= s->n_prims || s->host_kind[(size_t)calls[k]] != RTMI_PRIM_MEDIUM) return fail(RTMI_E_ARG, "calls[%d] = %d is not a medium primitive", k, calls[k]);
        if (narrow_from[k] < 0 || narrow_from[k] > calls[k]) return fail(RTMI_E_ARG, "narrow_from[%d] = %d must lie in [0, calls[%d] = %d]", k, narrow_from[k], k, calls[k]);
        if (k > 0 && narrow_from[k] < calls[k] && narrow_from[k] == narrow_from[k - 1] && calls[k] <= calls[k - 1])
            return fail(RTMI_E_ARG, "calls %d and %d share a narrowing Hitlist and must come in list order", k - 1, k);
    }
    s->dev.n_media = n_calls;
    bool any = false;
    for (int k = 0; k < n_calls; ++k) { s->dev.media_idx[k] = calls[k]; s->dev.media_lo[k] = narrow_from[k]; any = any || narrow_from[k] < calls[k]; }
    s->dev.media_seq = any ? 2 : 0;
    s->args.media_calls.assign(calls, calls + n_calls); s->args.media_lo.assign(narrow_from, narrow_from + n_calls); s->args.has_media_calls = true;
    s->args.media_mode = any ? 2 : 0;
    HIP_TRY(hipSetDevice(s->ctx->device));
    return reupload_descriptor(s);
}

RTMI_EXPORT int rtmi_scene_set_media_mode(rtmi_scene *s, int32_t mode) {
    if (!scene_ok(s)) return fail(RTMI_E_STATE, "invalid scene handle");
    if (mode != RTMI_MEDIA_DESCENT && mode != RTMI_MEDIA_HITLIST) return fail(RTMI_E_ARG, "mode must be RTMI_MEDIA_DESCENT or RTMI_MEDIA_HITLIST");
    if (mode == RTMI_MEDIA_HITLIST)
        for (int k = 1; k < s->dev.n_media; ++k)
            if (s->dev.media_idx[k] <= s->dev.media_idx[k - 1]) return fail(RTMI_E_ARG, "RTMI_MEDIA_HITLIST: the media must be called once each, in ascending primitive (= list) order");
    s->dev.media_seq = mode == RTMI_MEDIA_HITLIST ? 1 : 0;
    s->args.media_mode = mode;
    HIP_TRY(hipSetDevice(s->ctx->device));
    return reupload_descriptor(s);
}

RTMI_EXPORT int rtmi_scene_set_images(rtmi_scene *s, int32_t n_images, const int32_t *wh, const uint8_t *rgb) {
    if (!scene_ok(s)) return fail(RTMI_E_STATE, "invalid scene handle");
    if (n_images < 0 || (n_images > 0 && (!wh || !rgb))) return fail(RTMI_E_ARG, "bad image arguments");
    std::vector<int> whv;
    std::vector<long long> off;
    long long total = 0;
    for (int i = 0; i < n_images; ++i) {
        if (wh[2 * i] <= 0 || wh[2 * i + 1] <= 0 || wh[2 * i] > 65536 || wh[2 * i + 1] > 65536) return fail(RTMI_E_ARG, "image %d: bad size %dx%d", i, wh[2 * i], wh[2 * i + 1]);
        whv.push_back(wh[2 * i]); whv.push_back(wh[2 * i + 1]);
        off.push_back(total);
        total += (long long)wh[2 * i] * wh[2 * i + 1] * 3;
    }
    HIP_TRY(hipSetDevice(s->ctx->device));
    std::vector<unsigned char> px(rgb, rgb + total);
    int rc = upload(s, whv, &s->dev.image_wh);
    if (!rc) rc = upload(s, off, &s->dev.image_off);
    if (!rc) rc = upload(s, px, &s->dev.image_rgb);
    if (rc) return rc;
    s->dev.n_images = n_images;
    s->args.image_wh = whv; s->args.image_rgb = px;
    return reupload_descriptor(s);
}

RTMI_EXPORT int rtmi_scene_device_bytes(rtmi_scene *s, int64_t *out_bytes) {
    if (!scene_ok(s)) return fail(RTMI_E_STATE, "invalid scene handle");
    if (!out_bytes) return fail(RTMI_E_ARG, "out_bytes is NULL");
    *out_bytes = (int64_t)s->device_bytes;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_scene_destroy(rtmi_scene *s) {
    if (!s || s->magic != 0x52545343u) return fail(RTMI_E_STATE, "invalid scene handle");
    if (ctx_ok(s->ctx)) { (void)hipSetDevice(s->ctx->device); (void)hipStreamSynchronize(s->ctx->stream); }
    for (void *p : s->allocs) (void)hipFree(p);
    s->magic = 0;
    delete s;
    return RTMI_OK;
}

// ---- the hot path -------------------------------------------------------------------------------------
RTMI_EXPORT int32_t rtmi_local_tiles(int32_t nx, int32_t ny, int32_t first, int32_t stride) {
    if (nx <= 0 || ny <= 0 || first < 0 || stride <= 0) return 0;
    const int ntiles = tiles_x_of(nx) * tiles_y_of(ny);
    return first < ntiles ? (ntiles - first - 1) / stride + 1 : 0;
}

RTMI_EXPORT int rtmi_render_tiles_device(rtmi_scene *s, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed, int32_t precision,
                                         int32_t tile_first, int32_t tile_stride, void *d_tiles_linear, void *d_out_counters, void *stream) {
    int rc = check_render_args(s, nx, ny, ns, depth, precision);
    if (rc) return rc;
    if (tile_first < 0 || tile_stride <= 0) return fail(RTMI_E_ARG, "tile_first must be >= 0 and tile_stride > 0");
    if (!d_tiles_linear) return fail(RTMI_E_ARG, "d_tiles_linear is NULL");
    HIP_TRY(hipSetDevice(s->ctx->device));
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : s->ctx->stream;
    if (precision == RTMI_F64) return render_tiles_impl<double>(s, nx, ny, ns, depth, seed, tile_first, tile_stride, nullptr, d_tiles_linear, d_out_counters, st);
    return render_tiles_impl<float>(s, nx, ny, ns, depth, seed, tile_first, tile_stride, nullptr, d_tiles_linear, d_out_counters, st);
}

RTMI_EXPORT int rtmi_assemble_device(rtmi_ctx *c, int32_t nx, int32_t ny, int32_t world, int32_t tiles_per_rank, const void *d_gathered,
                                     void *d_out_linear, void *d_out_rgb8, void *stream) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (nx <= 0 || ny <= 0 || world <= 0 || tiles_per_rank <= 0 || !d_gathered) return fail(RTMI_E_ARG, "bad assemble arguments");
    const int ntiles = tiles_x_of(nx) * tiles_y_of(ny);
    if ((long long)world * tiles_per_rank < ntiles) return fail(RTMI_E_ARG, "world*tiles_per_rank (%d*%d) does not cover %d tiles", world, tiles_per_rank, ntiles);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
    const long long npx = (long long)nx * ny;
    // the 8-bit quantiser always runs in double on the double mean: both precisions share it
    hipLaunchKernelGGL((assemble_kernel<double>), dim3((unsigned)((npx + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       reinterpret_cast<const double *>(d_gathered), world, (size_t)tiles_per_rank * 192, tiles_x_of(nx), nx, ny,
                       reinterpret_cast<double *>(d_out_linear), reinterpret_cast<unsigned char *>(d_out_rgb8));
    HIP_TRY(hipGetLastError());
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_render_device(rtmi_scene *s, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed, int32_t precision,
                                   void *d_out_linear, void *d_out_rgb8, void *d_out_counters, void *stream) {
    int rc = check_render_args(s, nx, ny, ns, depth, precision);
    if (rc) return rc;
    rtmi_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int ntiles = tiles_x_of(nx) * tiles_y_of(ny);
    rc = c->tiles.ensure((size_t)ntiles * 64 * 3 * sizeof(double));
    if (rc) return rc;
    rc = rtmi_render_tiles_device(s, nx, ny, ns, depth, seed, precision, 0, 1, c->tiles.p, d_out_counters, stream);
    if (rc) return rc;
    if (!d_out_linear && !d_out_rgb8) return RTMI_OK;
    return rtmi_assemble_device(c, nx, ny, 1, ntiles, c->tiles.p, d_out_linear, d_out_rgb8, stream);
}

RTMI_EXPORT int rtmi_render(rtmi_scene *s, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed, int32_t precision,
                            int32_t x0, int32_t y0, int32_t x1, int32_t y1, double *out_linear, uint8_t *out_rgb8, uint64_t *out_counters) {
    int rc = check_render_args(s, nx, ny, ns, depth, precision);
    if (rc) return rc;
    if (x0 < 0 || y0 < 0 || x1 > nx || y1 > ny || x1 <= x0 || y1 <= y0) return fail(RTMI_E_ARG, "region [%d,%d)x[%d,%d) outside %dx%d", x0, x1, y0, y1, nx, ny);
    rtmi_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    // only the 8x8 tiles that intersect the region are rendered, and only the region's pixels in them are traced: both
    // counters describe exactly the region (a 16x8 spot check of a 1920x1080 frame costs 2 tiles, not 32400)
    const int w = x1 - x0, h = y1 - y0;
    const int tx0 = x0 / RTMI_TILE, ty0 = y0 / RTMI_TILE, tx1 = (x1 + RTMI_TILE - 1) / RTMI_TILE, ty1 = (y1 + RTMI_TILE - 1) / RTMI_TILE;
    const int wtx = tx1 - tx0, nwin = wtx * (ty1 - ty0);
    const size_t npx = (size_t)w * h;
    rc = c->tiles.ensure((size_t)nwin * 64 * 3 * sizeof(double));
    if (rc) return rc;
    rc = c->scratch_lin.ensure(npx * 3 * sizeof(double) + npx * 3 + 2 * sizeof(u64) + 64);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(c->scratch_lin.p);
    double *d_lin = reinterpret_cast<double *>(base);
    u64 *d_cnt = reinterpret_cast<u64 *>(base + npx * 3 * sizeof(double));
    unsigned char *d_q = reinterpret_cast<unsigned char *>(base + npx * 3 * sizeof(double) + 2 * sizeof(u64));
    const int rg[4] = {x0, y0, x1, y1};
    hipStream_t st = c->stream;
    if (precision == RTMI_F64) rc = render_tiles_impl<double>(s, nx, ny, ns, depth, seed, 0, 1, rg, c->tiles.p, d_cnt, st);
    else rc = render_tiles_impl<float>(s, nx, ny, ns, depth, seed, 0, 1, rg, c->tiles.p, d_cnt, st);
    if (rc) return rc;
    hipLaunchKernelGGL((assemble_region_kernel<double>), dim3((unsigned)((npx + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       reinterpret_cast<const double *>(c->tiles.p), tx0, ty0, wtx, x0, y0, w, h, d_lin, d_q);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    if (out_linear) HIP_TRY(hipMemcpy(out_linear, d_lin, npx * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (out_rgb8) HIP_TRY(hipMemcpy(out_rgb8, d_q, npx * 3, hipMemcpyDeviceToHost));
    if (out_counters) HIP_TRY(hipMemcpy(out_counters, d_cnt, 2 * sizeof(u64), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

// ---- one host process, several GPUs (the reference's host is ONE JVM: core.clj:100-108) ---------------------------------
RTMI_EXPORT int rtmi_scene_clone(rtmi_scene *src, rtmi_ctx *ctx, rtmi_scene **out_scene) {
    if (!scene_ok(src)) return fail(RTMI_E_STATE, "invalid scene handle");
    if (!ctx_ok(ctx)) return fail(RTMI_E_STATE, "invalid context handle");
    if (!out_scene) return fail(RTMI_E_ARG, "out_scene is NULL");
    DeviceGuard guard;
    const rtmi_scene::Args &A = src->args;
    rtmi_scene *s = nullptr;
    int rc = rtmi_scene_create_ex(ctx, src->n_prims, A.prim_kind.data(), A.prim_geom.data(), A.prim_mat.data(), src->n_mats, A.mat_kind.data(), A.mat_tex.data(),
                                  A.mat_param.data(), src->n_tex, A.tex_kind.data(), A.tex_param.data(), A.tex_child.data(), A.cam_kind, A.cam.data(),
                                  A.prim_flip.empty() ? nullptr : A.prim_flip.data(), A.prim_xform.empty() ? nullptr : A.prim_xform.data(),
                                  (int32_t)A.xform_kind.size(), A.xform_kind.empty() ? nullptr : A.xform_kind.data(),
                                  A.xform_param.empty() ? nullptr : A.xform_param.data(), &s);
    if (rc) return rc;
    if (!rc && !A.perlin_vec.empty()) rc = rtmi_scene_set_perlin(s, A.perlin_vec.data(), A.perm.data());
    if (!rc && !A.image_wh.empty()) rc = rtmi_scene_set_images(s, (int32_t)(A.image_wh.size() / 2), A.image_wh.data(), A.image_rgb.data());
    if (!rc && A.has_media_calls && A.media_mode == 2) rc = rtmi_scene_set_media_calls_narrowed(s, (int32_t)A.media_calls.size(), A.media_calls.data(), A.media_lo.data());
    else {
        if (!rc && A.has_media_calls) rc = rtmi_scene_set_media_calls(s, (int32_t)A.media_calls.size(), A.media_calls.data());
        if (!rc && A.media_mode) rc = rtmi_scene_set_media_mode(s, A.media_mode);
    }
    if (rc) { const std::string keep = g_err; rtmi_scene_destroy(s); g_err = keep; return rc; }
    *out_scene = s;
    return RTMI_OK;
}

namespace {
// RCCL, opened on first use: a single-GPU host never loads it.  (In a process that already maps an RCCL -- e.g. PyTorch's
// bundled one -- dlopen by soname returns that copy.)
struct Rccl {
    bool tried = false;
    void *h = nullptr;
    std::string err;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
std::mutex g_multi_mu;
Rccl g_rccl;
bool g_rccl_failed = false; // RCCL could not be opened / initialised / a gather failed once: later gathers use copies
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms; // one communicator set per device list (ncclCommInitAll), kept for the process

// Opens the first of `names` that dlopen accepts and binds the six entry points the gather needs.  On failure R.h stays null and
// R.err says why (dlerror() is read ONCE per failure: a second call returns NULL).
bool rccl_open(Rccl &R, const std::vector<std::string> &names) {
    std::string first_err;
    for (const std::string &name : names) {
        R.h = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (R.h) break;
        const char *e = dlerror();
        if (first_err.empty()) first_err = std::string("dlopen(") + name + "): " + (e ? e : "not found");
    }
    if (!R.h) { R.err = first_err.empty() ? std::string("no library name given") : first_err; return false; }
    bool ok = true;
    auto sym = [&](const char *n) { void *p = dlsym(R.h, n); if (!p && ok) { ok = false; R.err = std::string("the RCCL library lacks ") + n; } return p; };
    R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(sym("ncclCommInitAll"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
    R.Gather = reinterpret_cast<decltype(R.Gather)>(sym("ncclGather"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(R.h); R.h = nullptr; }
    return ok;
}

std::vector<std::string> rccl_names() {
    if (const char *e = std::getenv("RTMI_RCCL_LIB")) return {std::string(e)}; // another build of RCCL, or a bogus name to rehearse the fallback
    return {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
}

bool rccl_load() {
    Rccl &R = g_rccl;
    if (R.tried) return R.h != nullptr;
    R.tried = true;
    return rccl_open(R, rccl_names());
}

int ensure_event(hipEvent_t *e) {
    if (!*e) HIP_TRY(hipEventCreate(e));
    return RTMI_OK;
}

// a communicator set that produced an error is not used again: destroy it and let later gathers copy
void rccl_give_up(const std::vector<int> &devs) {
    auto it = g_comms.find(devs);
    if (it != g_comms.end()) {
        for (ncclComm_t cm : it->second) if (cm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(cm);
        g_comms.erase(it);
    }
    g_rccl_failed = true;
}
} // namespace

// Can the RCCL library `soname` (NULL: the names the gather itself tries, or $RTMI_RCCL_LIB) be opened, and does it export the entry
// points the in-library gather binds?  No device is touched.
RTMI_EXPORT int rtmi_rccl_probe(const char *soname) {
    Rccl R;
    const bool ok = rccl_open(R, soname ? std::vector<std::string>{std::string(soname)} : rccl_names());
    if (!ok) return fail(RTMI_E_DEVICE, "%s", R.err.c_str());
    dlclose(R.h);
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_render_multi_device(int32_t n, rtmi_scene *const *scenes, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed,
                                         int32_t precision, void *d_out_linear, void *d_out_rgb8, void *d_out_counters) {
    if (n <= 0 || n > 64 || !scenes) return fail(RTMI_E_ARG, "n must be 1..64 and scenes non-NULL");
    for (int r = 0; r < n; ++r) {
        int rc = check_render_args(scenes[r], nx, ny, ns, depth, precision);
        if (rc) return rc;
        for (int q = 0; q < r; ++q)
            if (scenes[q]->ctx == scenes[r]->ctx) return fail(RTMI_E_ARG, "replicas %d and %d share a context (a context is not re-entrant: one per replica)", q, r);
        if (scenes[r]->n_prims != scenes[0]->n_prims) return fail(RTMI_E_ARG, "replica %d is not a clone of replica 0", r);
    }
    std::lock_guard<std::mutex> lock(g_multi_mu);
    DeviceGuard guard;
    const int ntiles = tiles_x_of(nx) * tiles_y_of(ny);
    const int per = (ntiles + n - 1) / n;                  // every replica's record is padded to this many tiles
    const size_t rec = (size_t)per * 192 + 2;              // 8-byte words per record: tiles [per][64][3] doubles + the two metrics counters
    rtmi_ctx *c0 = scenes[0]->ctx;
    std::vector<int> devs((size_t)n);
    bool distinct = true;
    for (int r = 0; r < n; ++r) {
        devs[(size_t)r] = scenes[r]->ctx->device;
        for (int q = 0; q < r; ++q) distinct = distinct && devs[(size_t)q] != devs[(size_t)r];
    }
    // Which gather: RCCL whenever the replicas sit on distinct devices (n > 1); copies when they share a device (RCCL refuses duplicate
    // devices in one communicator).  RTMI_MULTI_GATHER = "copy": never RCCL; "rccl": RCCL or an error -- no silent substitution -- and also
    // for n = 1 (a one-rank communicator: the whole path -- dlopen, ncclCommInitAll, grouped in-place ncclGather -- on a one-GPU host).
    const char *force = std::getenv("RTMI_MULTI_GATHER");
    const bool want_copy = force && !std::strcmp(force, "copy"), want_rccl = force && !std::strcmp(force, "rccl");
    if (want_rccl && !distinct) return fail(RTMI_E_ARG, "RTMI_MULTI_GATHER=rccl needs replicas on distinct devices (RCCL refuses one device twice in a communicator)");
    bool use_rccl = distinct && !want_copy && (want_rccl || (n > 1 && !g_rccl_failed));
    std::vector<ncclComm_t> *comms = nullptr;
    if (use_rccl) { // the library owns its communicators: one set per device list, created on first use, kept for the process
        std::string why;
        if (!rccl_load()) why = g_rccl.err;
        else {
            auto it = g_comms.find(devs);
            if (it == g_comms.end()) {
                std::vector<ncclComm_t> cs((size_t)n);
                const ncclResult_t e = g_rccl.CommInitAll(cs.data(), n, devs.data());
                if (e != ncclSuccess) why = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e);
                else it = g_comms.emplace(devs, std::move(cs)).first;
            }
            if (why.empty()) comms = &it->second;
        }
        if (!why.empty()) { // no usable RCCL: the gather falls back to peer copies (same result, ordered with events), once and for all
            if (want_rccl) return fail(RTMI_E_DEVICE, "multi-device gather: %s", why.c_str());
            fprintf(stderr, "[rtmi] multi-device gather falls back to hipMemcpyPeerAsync: %s\n", why.c_str());
            g_rccl_failed = true;
            use_rccl = false;
        }
    }
    // From here on work is enqueued on the replicas' streams.  An error after the first launch must not return with kernels still in
    // flight on streams the caller believes idle (it may destroy the contexts next): bail() waits for every stream touched so far.
    int launched = 0;
    auto bail = [&](int code) {
        const std::string keep = g_err;
        for (int r = 0; r < launched; ++r) { (void)hipSetDevice(scenes[r]->ctx->device); (void)hipStreamSynchronize(scenes[r]->ctx->stream); }
        (void)hipSetDevice(c0->device); (void)hipStreamSynchronize(c0->stream);
        g_err = keep;
        return code;
    };
#define HIP_BAIL(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(RTMI_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    // 1. every replica renders its tiles (r, r+n, ...) on its own device and stream, straight into its record
    HIP_TRY(hipSetDevice(c0->device));
    int rc = c0->multi.ensure((size_t)n * rec * 8);
    if (rc) return rc;
    char *gathered = reinterpret_cast<char *>(c0->multi.p);
    std::vector<char *> recs((size_t)n);
    for (int r = 0; r < n; ++r) {
        rtmi_ctx *cr = scenes[r]->ctx;
        HIP_BAIL(hipSetDevice(cr->device));
        if (r == 0) recs[0] = gathered; // in place: replica 0's record is the first of the gathered buffer
        else { rc = cr->multi.ensure(rec * 8); if (rc) return bail(rc); recs[(size_t)r] = reinterpret_cast<char *>(cr->multi.p); }
        char *buf = recs[(size_t)r];
        if (cr->consume_pending) { // the previous frame's copy out of this record (on another replica 0's stream) must have read it
            HIP_BAIL(hipStreamWaitEvent(cr->stream, cr->ev_consumed, 0));
            cr->consume_pending = false;
        }
        launched = r + 1;
        if (precision == RTMI_F64) rc = render_tiles_impl<double>(scenes[r], nx, ny, ns, depth, seed, r, n, nullptr, buf, buf + (size_t)per * 192 * 8, cr->stream);
        else rc = render_tiles_impl<float>(scenes[r], nx, ny, ns, depth, seed, r, n, nullptr, buf, buf + (size_t)per * 192 * 8, cr->stream);
        if (rc) return bail(rc);
        if (!use_rccl && r > 0) { rc = ensure_event(&cr->ev_done); if (rc) return bail(rc); HIP_BAIL(hipEventRecord(cr->ev_done, cr->stream)); }
    }
    // 2. ONE gather to replica 0's device
    HIP_BAIL(hipSetDevice(c0->device));
    rc = ensure_event(&c0->ev_g0); if (!rc) rc = ensure_event(&c0->ev_g1);
    if (rc) return bail(rc);
    HIP_BAIL(hipEventRecord(c0->ev_g0, c0->stream));
    if (use_rccl) {
        ncclResult_t e = g_rccl.GroupStart();
        for (int r = 0; r < n && e == ncclSuccess; ++r) {
            (void)hipSetDevice(scenes[r]->ctx->device); // (a communicator knows its device; older RCCLs still want it current for calls inside a group)
            e = g_rccl.Gather(recs[(size_t)r], r == 0 ? gathered : nullptr, rec, ncclUint64, 0, (*comms)[(size_t)r], scenes[r]->ctx->stream);
        }
        const ncclResult_t e2 = g_rccl.GroupEnd();
        (void)hipSetDevice(c0->device);
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) { // this communicator set is not trusted again: later calls gather by copies
            const int code = fail(RTMI_E_DEVICE, "ncclGather: %s", g_rccl.GetErrorString(e));
            rccl_give_up(devs);
            return bail(code);
        }
        c0->last_gather_path = RTMI_GATHER_RCCL;
    } else {
        bool peer = false;
        for (int r = 1; r < n; ++r) { // replicas sharing a device (rehearsal on a one-GPU host), RTMI_MULTI_GATHER=copy, or no usable RCCL: copies on replica 0's stream
            rtmi_ctx *cr = scenes[r]->ctx;
            HIP_BAIL(hipStreamWaitEvent(c0->stream, cr->ev_done, 0));
            if (cr->device == c0->device) HIP_BAIL(hipMemcpyAsync(gathered + (size_t)r * rec * 8, recs[(size_t)r], rec * 8, hipMemcpyDeviceToDevice, c0->stream));
            else { peer = true; HIP_BAIL(hipMemcpyPeerAsync(gathered + (size_t)r * rec * 8, c0->device, recs[(size_t)r], cr->device, rec * 8, c0->stream)); }
            // replica r's NEXT render into its record waits for this copy (its stream is not otherwise ordered with replica 0's)
            if (cr->ev_consumed && cr->ev_consumed_device != c0->device) { (void)hipSetDevice(cr->ev_consumed_device); (void)hipEventDestroy(cr->ev_consumed); cr->ev_consumed = nullptr; HIP_BAIL(hipSetDevice(c0->device)); }
            if (!cr->ev_consumed) { HIP_BAIL(hipEventCreateWithFlags(&cr->ev_consumed, hipEventDisableTiming)); cr->ev_consumed_device = c0->device; }
            HIP_BAIL(hipEventRecord(cr->ev_consumed, c0->stream));
            cr->consume_pending = true;
        }
        c0->last_gather_path = n == 1 ? RTMI_GATHER_NONE : (peer ? RTMI_GATHER_PEER_COPY : RTMI_GATHER_SAME_DEVICE);
    }
    HIP_BAIL(hipEventRecord(c0->ev_g1, c0->stream));
    c0->have_gather = true;
    // 3. replica 0 un-tiles, quantises and sums the counters
    if (d_out_linear || d_out_rgb8) {
        const long long npx = (long long)nx * ny;
        hipLaunchKernelGGL((assemble_kernel<double>), dim3((unsigned)((npx + kBlock - 1) / kBlock)), dim3(kBlock), 0, c0->stream,
                           reinterpret_cast<const double *>(gathered), n, rec, tiles_x_of(nx), nx, ny,
                           reinterpret_cast<double *>(d_out_linear), reinterpret_cast<unsigned char *>(d_out_rgb8));
        HIP_BAIL(hipGetLastError());
    }
    if (d_out_counters) {
        hipLaunchKernelGGL(sum_counters_kernel, dim3(1), dim3(64), 0, c0->stream, reinterpret_cast<const u64 *>(gathered), n, rec, (size_t)per * 192,
                           reinterpret_cast<u64 *>(d_out_counters));
        HIP_BAIL(hipGetLastError());
    }
#undef HIP_BAIL
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_render_multi(int32_t n, rtmi_scene *const *scenes, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed, int32_t precision,
                                  double *out_linear, uint8_t *out_rgb8, uint64_t *out_counters) {
    if (n <= 0 || !scenes || !scene_ok(scenes[0])) return fail(RTMI_E_ARG, "bad replica list");
    DeviceGuard guard;
    rtmi_ctx *c0 = scenes[0]->ctx;
    HIP_TRY(hipSetDevice(c0->device));
    const size_t npx = (size_t)nx * (size_t)ny;
    int rc = c0->scratch_lin.ensure(npx * 3 * sizeof(double) + npx * 3 + 2 * sizeof(u64) + 64);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(c0->scratch_lin.p);
    double *d_lin = reinterpret_cast<double *>(base);
    u64 *d_cnt = reinterpret_cast<u64 *>(base + npx * 3 * sizeof(double));
    unsigned char *d_q = reinterpret_cast<unsigned char *>(base + npx * 3 * sizeof(double) + 2 * sizeof(u64));
    rc = rtmi_render_multi_device(n, scenes, nx, ny, ns, depth, seed, precision, d_lin, d_q, d_cnt);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c0->device));
    HIP_TRY(hipStreamSynchronize(c0->stream)); // ordered after every replica's render through the gather
    if (out_linear) HIP_TRY(hipMemcpy(out_linear, d_lin, npx * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (out_rgb8) HIP_TRY(hipMemcpy(out_rgb8, d_q, npx * 3, hipMemcpyDeviceToHost));
    if (out_counters) HIP_TRY(hipMemcpy(out_counters, d_cnt, 2 * sizeof(u64), hipMemcpyDeviceToHost));
    for (int r = 1; r < n; ++r) { HIP_TRY(hipSetDevice(scenes[r]->ctx->device)); HIP_TRY(hipStreamSynchronize(scenes[r]->ctx->stream)); scenes[r]->ctx->consume_pending = false; }
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_last_gather_path(rtmi_ctx *c, int32_t *path) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (!c->have_gather) return fail(RTMI_E_STATE, "no multi-device render on this context yet");
    if (path) *path = c->last_gather_path;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_stream_idle(rtmi_ctx *c, int32_t *idle) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    HIP_TRY(hipSetDevice(c->device));
    const hipError_t e = hipStreamQuery(c->stream);
    if (e != hipSuccess && e != hipErrorNotReady) return fail(RTMI_E_DEVICE, "hipStreamQuery: %s", hipGetErrorString(e));
    if (idle) *idle = e == hipSuccess ? 1 : 0;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_last_passes(rtmi_ctx *c, int32_t *passes) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (passes) *passes = c->last_passes;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_last_accel(rtmi_ctx *c, int32_t *accel) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (c->last_accel < 0) return fail(RTMI_E_STATE, "no render on this context yet");
    if (accel) *accel = c->last_accel;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_last_gather_ms(rtmi_ctx *c, double *ms) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (!c->have_gather) return fail(RTMI_E_STATE, "no multi-device render on this context yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev_g1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, c->ev_g0, c->ev_g1));
    if (ms) *ms = t;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_last_traversal_counters(rtmi_ctx *c, uint64_t *out_aabb_tests, uint64_t *out_prim_tests) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (!c->count_traversal) return fail(RTMI_E_STATE, "set option count_traversal = 1 before the render");
    if (!c->counters.p) return fail(RTMI_E_STATE, "no render on this context yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream ? c->last_stream : c->stream));
    u64 v[2] = {0, 0};
    HIP_TRY(hipMemcpy(v, reinterpret_cast<u64 *>(c->counters.p) + 3, sizeof v, hipMemcpyDeviceToHost));
    if (out_aabb_tests) *out_aabb_tests = v[0];
    if (out_prim_tests) *out_prim_tests = v[1];
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_last_trace_ms(rtmi_ctx *c, double *ms, int32_t *launches) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (!(c->flags & RTMI_FLAG_TIMING)) return fail(RTMI_E_STATE, "context was not created with RTMI_FLAG_TIMING");
    double total = 0.0, reduce = 0.0;
    for (int k = 0; k < c->events_used; ++k) {
        HIP_TRY(hipEventSynchronize(c->events[(size_t)k].second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, c->events[(size_t)k].first, c->events[(size_t)k].second));
        total += t;
        if ((size_t)k < c->events_r.size()) {
            HIP_TRY(hipEventSynchronize(c->events_r[(size_t)k]));
            HIP_TRY(hipEventElapsedTime(&t, c->events[(size_t)k].second, c->events_r[(size_t)k]));
            reduce += t;
        }
    }
    if (ms) *ms = total;
    if (launches) *launches = c->events_used;
    c->last_reduce_ms = reduce; c->last_reduce_launches = c->events_used;
    c->events_used = 0; // the next render starts a new measurement window
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_last_reduce_ms(rtmi_ctx *c, double *ms, int32_t *launches) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (!(c->flags & RTMI_FLAG_TIMING)) return fail(RTMI_E_STATE, "context was not created with RTMI_FLAG_TIMING");
    if (ms) *ms = c->last_reduce_ms;
    if (launches) *launches = c->last_reduce_launches;
    return RTMI_OK;
}

// ---- probes ---------------------------------------------------------------------------------------------
namespace {
struct Tmp { // scoped device temporaries for the (synchronous) probe entry points
    std::vector<void *> ptrs;
    ~Tmp() { for (void *p : ptrs) (void)hipFree(p); }
    void *alloc(size_t bytes) { void *p = nullptr; if (hipMalloc(&p, std::max<size_t>(bytes, 8)) != hipSuccess) return nullptr; ptrs.push_back(p); return p; }
    void *up(const void *src, size_t bytes) { void *p = alloc(bytes); if (p && src && bytes) if (hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr; return p; }
};
#define PROBE_PROLOGUE(scene_)                                                             \
    if (!scene_ok(scene_)) return fail(RTMI_E_STATE, "invalid scene handle");             \
    if (precision != RTMI_F64 && precision != RTMI_F32) return fail(RTMI_E_ARG, "bad precision"); \
    if (n < 0) return fail(RTMI_E_ARG, "n < 0");                                           \
    if (n == 0) return RTMI_OK;                                                            \
    rtmi_ctx *c = (scene_)->ctx;                                                           \
    HIP_TRY(hipSetDevice(c->device));                                                      \
    Tmp tmp;                                                                               \
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
#define PROBE_EPILOGUE()                      \
    HIP_TRY(hipGetLastError());               \
    HIP_TRY(hipStreamSynchronize(c->stream));
} // namespace

RTMI_EXPORT int rtmi_probe_hit(rtmi_scene *s, int32_t precision, int32_t n, const double *rays, double t_min, double t_max, double *out) {
    PROBE_PROLOGUE(s)
    if (!rays || !out) return fail(RTMI_E_ARG, "NULL array");
    if (precision == RTMI_F32 && s->dev.has_ext) return fail(RTMI_E_UNSUPPORTED, "rectangles / triangles / instances are FP64 only");
    double *d_rays = (double *)tmp.up(rays, (size_t)n * 7 * sizeof(double));
    double *d_out = (double *)tmp.alloc((size_t)n * 11 * sizeof(double));
    if (!d_rays || !d_out) return fail(RTMI_E_NOMEM, "probe buffers");
    int ppt, npt; size_t lds;
    if (precision == RTMI_F64) {
        lds_plan(c, s->dev.n_static, sizeof(double), &ppt, &npt, &lds);
        if (s->dev.has_ext) {
            const int seq = s->dev.media_seq;
            if (c->accel == RTMI_ACCEL_BVH) hipLaunchKernelGGL((seq == 2 ? probe_hit_kernel<double, SCAN_BVH, true, 2> : seq ? probe_hit_kernel<double, SCAN_BVH, true, 1> : probe_hit_kernel<double, SCAN_BVH, true>), dim3(grid), dim3(kBlock), (size_t)RTMI_BVH_STACK * kBlock * sizeof(int) + 16, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out);
            else hipLaunchKernelGGL((seq == 2 ? probe_hit_kernel<double, SCAN_SGPR_CULL, true, 2> : seq ? probe_hit_kernel<double, SCAN_SGPR_CULL, true, 1> : probe_hit_kernel<double, SCAN_SGPR_CULL, true>), dim3(grid), dim3(kBlock), 64, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out);
        } else
        switch (c->accel == RTMI_ACCEL_BVH ? SCAN_BVH : c->scan_variant) {
        case SCAN_BVH: hipLaunchKernelGGL((probe_hit_kernel<double, SCAN_BVH>), dim3(grid), dim3(kBlock), std::max(lds, (size_t)RTMI_BVH_STACK * kBlock * sizeof(int) + 16), c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out); break;
        case SCAN_SGPR_CULL: hipLaunchKernelGGL((probe_hit_kernel<double, SCAN_SGPR_CULL>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out); break;
        case SCAN_SGPR: hipLaunchKernelGGL((probe_hit_kernel<double, SCAN_SGPR>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out); break;
        case SCAN_LDS_PIPE: hipLaunchKernelGGL((probe_hit_kernel<double, SCAN_LDS_PIPE>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out); break;
        default: hipLaunchKernelGGL((probe_hit_kernel<double, SCAN_LDS_LITERAL>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out);
        }
    } else {
        lds_plan(c, s->dev.n_static, sizeof(float), &ppt, &npt, &lds);
        switch (c->accel == RTMI_ACCEL_BVH ? SCAN_BVH : c->scan_variant) {
        case SCAN_BVH: hipLaunchKernelGGL((probe_hit_kernel<float, SCAN_BVH>), dim3(grid), dim3(kBlock), std::max(lds, (size_t)RTMI_BVH_STACK * kBlock * sizeof(int) + 16), c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out); break;
        case SCAN_SGPR_CULL: hipLaunchKernelGGL((probe_hit_kernel<float, SCAN_SGPR_CULL>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out); break;
        case SCAN_SGPR: hipLaunchKernelGGL((probe_hit_kernel<float, SCAN_SGPR>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out); break;
        case SCAN_LDS_PIPE: hipLaunchKernelGGL((probe_hit_kernel<float, SCAN_LDS_PIPE>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out); break;
        default: hipLaunchKernelGGL((probe_hit_kernel<float, SCAN_LDS_LITERAL>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, t_min, t_max, d_out);
        }
    }
    PROBE_EPILOGUE()
    HIP_TRY(hipMemcpy(out, d_out, (size_t)n * 11 * sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_probe_paths(rtmi_scene *s, int32_t precision, int32_t n, const double *rays, const uint64_t *keys, uint64_t ctr0, int32_t depth,
                                 double *out_rgb, uint64_t *out_nseg, double *log, int32_t max_seg, int32_t *out_nlog) {
    PROBE_PROLOGUE(s)
    if (!rays || !keys || !out_rgb) return fail(RTMI_E_ARG, "NULL array");
    if (log && max_seg <= 0) return fail(RTMI_E_ARG, "log given but max_seg <= 0");
    if (precision == RTMI_F32 && s->dev.has_ext) return fail(RTMI_E_UNSUPPORTED, "rectangles / triangles / instances are FP64 only");
    double *d_rays = (double *)tmp.up(rays, (size_t)n * 7 * sizeof(double));
    u64 *d_keys = (u64 *)tmp.up(keys, (size_t)n * sizeof(u64));
    double *d_rgb = (double *)tmp.alloc((size_t)n * 3 * sizeof(double));
    u64 *d_nseg = (u64 *)tmp.alloc((size_t)n * sizeof(u64));
    int *d_nlog = (int *)tmp.alloc((size_t)n * sizeof(int));
    const size_t log_bytes = log ? (size_t)n * max_seg * RTMI_SEG_REC * sizeof(double) : 0;
    double *d_log = log ? (double *)tmp.alloc(log_bytes) : nullptr;
    if (!d_rays || !d_keys || !d_rgb || !d_nseg || !d_nlog || (log && !d_log)) return fail(RTMI_E_NOMEM, "probe buffers");
    if (d_log) HIP_TRY(hipMemset(d_log, 0, log_bytes));
    int ppt, npt; size_t lds;
    if (precision == RTMI_F64) {
        lds_plan(c, s->dev.n_static, sizeof(double), &ppt, &npt, &lds);
        if (s->dev.has_ext) {
            const int seq = s->dev.media_seq;
            if (c->accel == RTMI_ACCEL_BVH) hipLaunchKernelGGL((seq == 2 ? probe_paths_kernel<double, SCAN_BVH, true, 2> : seq ? probe_paths_kernel<double, SCAN_BVH, true, 1> : probe_paths_kernel<double, SCAN_BVH, true>), dim3(grid), dim3(kBlock), (size_t)RTMI_BVH_STACK * kBlock * sizeof(int) + 16, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog);
            else hipLaunchKernelGGL((seq == 2 ? probe_paths_kernel<double, SCAN_SGPR_CULL, true, 2> : seq ? probe_paths_kernel<double, SCAN_SGPR_CULL, true, 1> : probe_paths_kernel<double, SCAN_SGPR_CULL, true>), dim3(grid), dim3(kBlock), 64, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog);
        } else
        switch (c->accel == RTMI_ACCEL_BVH ? SCAN_BVH : c->scan_variant) {
        case SCAN_BVH: hipLaunchKernelGGL((probe_paths_kernel<double, SCAN_BVH>), dim3(grid), dim3(kBlock), std::max(lds, (size_t)RTMI_BVH_STACK * kBlock * sizeof(int) + 16), c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog); break;
        case SCAN_SGPR_CULL: hipLaunchKernelGGL((probe_paths_kernel<double, SCAN_SGPR_CULL>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog); break;
        case SCAN_SGPR: hipLaunchKernelGGL((probe_paths_kernel<double, SCAN_SGPR>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog); break;
        case SCAN_LDS_PIPE: hipLaunchKernelGGL((probe_paths_kernel<double, SCAN_LDS_PIPE>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog); break;
        default: hipLaunchKernelGGL((probe_paths_kernel<double, SCAN_LDS_LITERAL>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog);
        }
    } else {
        lds_plan(c, s->dev.n_static, sizeof(float), &ppt, &npt, &lds);
        switch (c->accel == RTMI_ACCEL_BVH ? SCAN_BVH : c->scan_variant) {
        case SCAN_BVH: hipLaunchKernelGGL((probe_paths_kernel<float, SCAN_BVH>), dim3(grid), dim3(kBlock), std::max(lds, (size_t)RTMI_BVH_STACK * kBlock * sizeof(int) + 16), c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog); break;
        case SCAN_SGPR_CULL: hipLaunchKernelGGL((probe_paths_kernel<float, SCAN_SGPR_CULL>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog); break;
        case SCAN_SGPR: hipLaunchKernelGGL((probe_paths_kernel<float, SCAN_SGPR>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog); break;
        case SCAN_LDS_PIPE: hipLaunchKernelGGL((probe_paths_kernel<float, SCAN_LDS_PIPE>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog); break;
        default: hipLaunchKernelGGL((probe_paths_kernel<float, SCAN_LDS_LITERAL>), dim3(grid), dim3(kBlock), lds, c->stream, s->d_dev, ppt, npt, n, d_rays, d_keys, (u64)ctr0, depth, d_rgb, d_nseg, d_log, max_seg, d_nlog);
        }
    }
    PROBE_EPILOGUE()
    HIP_TRY(hipMemcpy(out_rgb, d_rgb, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (out_nseg) HIP_TRY(hipMemcpy(out_nseg, d_nseg, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost));
    if (out_nlog) HIP_TRY(hipMemcpy(out_nlog, d_nlog, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    if (log) HIP_TRY(hipMemcpy(log, d_log, log_bytes, hipMemcpyDeviceToHost));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_probe_camera(rtmi_scene *s, int32_t precision, int32_t n, const double *uv, const uint64_t *keys, double *out) {
    PROBE_PROLOGUE(s)
    if (!uv || !keys || !out) return fail(RTMI_E_ARG, "NULL array");
    double *d_uv = (double *)tmp.up(uv, (size_t)n * 2 * sizeof(double));
    u64 *d_keys = (u64 *)tmp.up(keys, (size_t)n * sizeof(u64));
    double *d_out = (double *)tmp.alloc((size_t)n * 8 * sizeof(double));
    if (!d_uv || !d_keys || !d_out) return fail(RTMI_E_NOMEM, "probe buffers");
    if (precision == RTMI_F64) hipLaunchKernelGGL((probe_camera_kernel<double>), dim3(grid), dim3(kBlock), 0, c->stream, s->d_dev, n, d_uv, d_keys, d_out);
    else hipLaunchKernelGGL((probe_camera_kernel<float>), dim3(grid), dim3(kBlock), 0, c->stream, s->d_dev, n, d_uv, d_keys, d_out);
    PROBE_EPILOGUE()
    HIP_TRY(hipMemcpy(out, d_out, (size_t)n * 8 * sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_probe_texture(rtmi_scene *s, int32_t precision, int32_t tex, int32_t n, const double *uvp, double *out) {
    PROBE_PROLOGUE(s)
    if (!uvp || !out) return fail(RTMI_E_ARG, "NULL array");
    if (tex < 0 || tex >= s->n_tex) return fail(RTMI_E_ARG, "texture index %d out of range", tex);
    double *d_in = (double *)tmp.up(uvp, (size_t)n * 5 * sizeof(double));
    double *d_out = (double *)tmp.alloc((size_t)n * 3 * sizeof(double));
    if (!d_in || !d_out) return fail(RTMI_E_NOMEM, "probe buffers");
    if (s->dev.has_ext) {
        if (precision != RTMI_F64) return fail(RTMI_E_UNSUPPORTED, "procedural / image textures are FP64 only");
        hipLaunchKernelGGL((probe_texture_kernel<double, true>), dim3(grid), dim3(kBlock), 0, c->stream, s->d_dev, tex, n, d_in, d_out);
    } else if (precision == RTMI_F64) hipLaunchKernelGGL((probe_texture_kernel<double>), dim3(grid), dim3(kBlock), 0, c->stream, s->d_dev, tex, n, d_in, d_out);
    else hipLaunchKernelGGL((probe_texture_kernel<float>), dim3(grid), dim3(kBlock), 0, c->stream, s->d_dev, tex, n, d_in, d_out);
    PROBE_EPILOGUE()
    HIP_TRY(hipMemcpy(out, d_out, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_probe_scatter(rtmi_scene *s, int32_t precision, int32_t mat, int32_t n, const double *rays, const double *hits,
                                   const uint64_t *keys, double *out) {
    PROBE_PROLOGUE(s)
    if (!rays || !hits || !keys || !out) return fail(RTMI_E_ARG, "NULL array");
    if (mat < 0 || mat >= s->n_mats) return fail(RTMI_E_ARG, "material index %d out of range", mat);
    double *d_rays = (double *)tmp.up(rays, (size_t)n * 7 * sizeof(double));
    double *d_hits = (double *)tmp.up(hits, (size_t)n * 8 * sizeof(double));
    u64 *d_keys = (u64 *)tmp.up(keys, (size_t)n * sizeof(u64));
    double *d_out = (double *)tmp.alloc((size_t)n * 9 * sizeof(double));
    if (!d_rays || !d_hits || !d_keys || !d_out) return fail(RTMI_E_NOMEM, "probe buffers");
    if (s->dev.has_ext) {
        if (precision != RTMI_F64) return fail(RTMI_E_UNSUPPORTED, "procedural / image textures are FP64 only");
        hipLaunchKernelGGL((probe_scatter_kernel<double, true>), dim3(grid), dim3(kBlock), 0, c->stream, s->d_dev, mat, n, d_rays, d_hits, d_keys, d_out);
    } else if (precision == RTMI_F64) hipLaunchKernelGGL((probe_scatter_kernel<double>), dim3(grid), dim3(kBlock), 0, c->stream, s->d_dev, mat, n, d_rays, d_hits, d_keys, d_out);
    else hipLaunchKernelGGL((probe_scatter_kernel<float>), dim3(grid), dim3(kBlock), 0, c->stream, s->d_dev, mat, n, d_rays, d_hits, d_keys, d_out);
    PROBE_EPILOGUE()
    HIP_TRY(hipMemcpy(out, d_out, (size_t)n * 9 * sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_probe_rng(rtmi_ctx *c, int32_t precision, uint64_t key, uint64_t d0, int32_t n, uint64_t *out_bits, double *out_real) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (precision != RTMI_F64 && precision != RTMI_F32) return fail(RTMI_E_ARG, "bad precision");
    if (n <= 0 || !out_bits || !out_real) return fail(RTMI_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    Tmp tmp;
    u64 *d_bits = (u64 *)tmp.alloc((size_t)n * sizeof(u64));
    double *d_real = (double *)tmp.alloc((size_t)n * sizeof(double));
    if (!d_bits || !d_real) return fail(RTMI_E_NOMEM, "probe buffers");
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    if (precision == RTMI_F64) hipLaunchKernelGGL((probe_rng_kernel<double>), dim3(grid), dim3(kBlock), 0, c->stream, (u64)key, (u64)d0, n, d_bits, d_real);
    else hipLaunchKernelGGL((probe_rng_kernel<float>), dim3(grid), dim3(kBlock), 0, c->stream, (u64)key, (u64)d0, n, d_bits, d_real);
    PROBE_EPILOGUE()
    HIP_TRY(hipMemcpy(out_bits, d_bits, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_real, d_real, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

// the path's own FP64 helpers (rtmi_device.h): square root with the wave-uniform fast path, the table-driven atan2 / asin and the uv
// formula of get-sphere-uv, the per-ray reciprocal of the sphere roots, division by a constant
__global__ void probe_math_kernel(int n, const double *abc, double tmin, double tmax, int n_slots, double *out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double a = abc[3 * k], b = abc[3 * k + 1], c = abc[3 * k + 2];
    double o[RTMI_PROBE_MATH_SLOTS];
    o[0] = rt_sqrt(a);
    const TrigTable K = trig_table();
    o[1] = rt_atan2(a, b, K);
    o[2] = rt_asin(a, K);
    Real<double>::sphere_uv(a, b, c, &o[3], &o[4]);
    const Quot<double> q = make_quot<double>(b, tmin, tmax);
    o[5] = q(a);
    o[6] = div_const(a, K[29], K[27]);
    o[7] = (double)q.fast;
    o[8] = (double)float_above(c); // the traversal's float bound of the closest hit so far: a float >= c, within two ulps
    o[9] = rt_log_unit(a);         // ConstantMedium's free-flight log of a draw in [0, 1)
    const bool fast = rcp_in_range(b) && tmin >= 0x1p-300 && tmax <= 0x1p200; // a rectangle's t = a / b by the refined reciprocal of a signed divisor (ext_box_faces, scan_small_ext)
    o[10] = div_by(a, refined_rcp(b), fast);
    o[11] = (double)fast;
    for (int j = 0; j < n_slots; ++j) out[(size_t)k * n_slots + j] = o[j];
}

RTMI_EXPORT int rtmi_probe_arith(rtmi_ctx *c, int32_t n, const double *abc, double *out) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (n <= 0 || !abc || !out) return fail(RTMI_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    Tmp tmp;
    double *d_in = (double *)tmp.up(abc, (size_t)n * 3 * sizeof(double));
    double *d_out = (double *)tmp.alloc((size_t)n * 3 * sizeof(double));
    if (!d_in || !d_out) return fail(RTMI_E_NOMEM, "probe buffers");
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(probe_arith_kernel, dim3(grid), dim3(kBlock), 0, c->stream, n, d_in, d_out);
    PROBE_EPILOGUE()
    HIP_TRY(hipMemcpy(out, d_out, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

// n_slots values per triple (1 .. RTMI_PROBE_MATH_SLOTS): the caller states how many its buffer holds per triple
RTMI_EXPORT int rtmi_probe_math2(rtmi_ctx *c, int32_t n, const double *abc, double tmin, double tmax, int32_t n_slots, double *out) {
    if (!ctx_ok(c)) return fail(RTMI_E_STATE, "invalid context handle");
    if (n <= 0 || !abc || !out) return fail(RTMI_E_ARG, "bad arguments");
    if (n_slots < 1 || n_slots > RTMI_PROBE_MATH_SLOTS) return fail(RTMI_E_ARG, "n_slots must be 1..%d", RTMI_PROBE_MATH_SLOTS);
    HIP_TRY(hipSetDevice(c->device));
    Tmp tmp;
    double *d_in = (double *)tmp.up(abc, (size_t)n * 3 * sizeof(double));
    double *d_out = (double *)tmp.alloc((size_t)n * (size_t)n_slots * sizeof(double));
    if (!d_in || !d_out) return fail(RTMI_E_NOMEM, "probe buffers");
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(probe_math_kernel, dim3(grid), dim3(kBlock), 0, c->stream, n, d_in, tmin, tmax, (int)n_slots, d_out);
    PROBE_EPILOGUE()
    HIP_TRY(hipMemcpy(out, d_out, (size_t)n * (size_t)n_slots * sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}
// the entry as first published: EIGHT values per triple (version 203 wrote nine into the same signature; hosts built against either header get eight again)
RTMI_EXPORT int rtmi_probe_math(rtmi_ctx *c, int32_t n, const double *abc, double tmin, double tmax, double *out) {
    return rtmi_probe_math2(c, n, abc, tmin, tmax, 8, out);
}
