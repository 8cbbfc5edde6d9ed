/*
 * rtmi.h -- C-ABI of librtmi.so: raytrace-clj's per-pixel Monte-Carlo sampling path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  The reference (gonewest818/raytrace-clj) has no FFI of its own; the
 * seam this library replaces is the body of the render loop
 *     src/raytrace_clj/core.clj:100-108   (cp/upmap over tiled-coords -> pixel -> set-pixel)
 * i.e. everything `pixel` (core.clj:43-57) and `color` (core.clj:17-41) call through the Hitable /
 * Shader / Texture / Camera protocols (hitable.clj:7-10, shader.clj:22-24, texture.clj:8-9,
 * camera.clj:5-6).  A host (Clojure via JNA, Python via ctypes, C++) flattens a
 * {:camera c :world w} scene (scene.clj:321,331) into the arrays below and calls rtmi_render.
 * See INTEGRATION.md for the JNA stub.
 *
 * Conventions (normative):
 *   - plain C symbols, no exceptions cross the boundary;
 *   - every call returns int: 0 = RTMI_OK, < 0 = error class; rtmi_last_error() gives the text
 *     (thread-local, library-owned, valid until the next failing call on that thread);
 *   - rtmi_ctx / rtmi_scene are opaque, created and destroyed by the library only;
 *   - every host array argument is caller-allocated, caller-owned and only read/written for the
 *     duration of the call (the library keeps no host pointers);
 *   - host-side scalars and arrays are double / int32 whatever precision the kernels compute in;
 *   - a context is not re-entrant; distinct contexts may be driven from distinct threads;
 *   - framebuffers are row-major, RGB interleaved, ROW 0 = TOP, i.e. after the reference's flip
 *     y_out = ny-1-j (core.clj:105).
 *
 * Flat scene layout (what a flattener over the reference's records produces):
 *   primitives, in Hitlist order (hitable.clj:15-26):
 *     prim_kind[i]  RTMI_PRIM_SPHERE (hitable.clj:180) | RTMI_PRIM_UVSPHERE (:141) | RTMI_PRIM_MOVING (:224)
 *     prim_geom[i*9 + 0..8] = center0.xyz, radius, center1.xyz, t0, t1   (static: center1 = center0, t0 = 0, t1 = 1)
 *     prim_mat[i]   material index
 *   materials (shader.clj):
 *     mat_kind[m]   RTMI_MAT_LAMBERTIAN (:29) | RTMI_MAT_METAL (:46) | RTMI_MAT_DIELECTRIC (:76) | RTMI_MAT_DIFFUSE_LIGHT (:114)
 *     mat_tex[m]    texture index of albedo / emission (-1 for dielectric)
 *     mat_param[m]  fuzz (metal) | ri (dielectric) | 0
 *   textures (texture.clj):
 *     tex_kind[t]   RTMI_TEX_CONSTANT (:14) | RTMI_TEX_UVGRADIENT (:26) | RTMI_TEX_CHECKER (:44)
 *     tex_param[t*12 + ..] = constant: color.rgb | gradient: co, cu, cv, cuv (4 x rgb) | checker: scale
 *     tex_child[t*2 + 0..1] = checker tex0, tex1 (else -1)
 *   camera (camera.clj):
 *     cam_kind      RTMI_CAM_PINHOLE (:8) | RTMI_CAM_THINLENS (:35)
 *     cam[24]       origin, lleft, horiz, vert, u, v, w (7 x xyz), aperture, t0, t1
 */
#ifndef RTMI_H
#define RTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_OK 0
#define RTMI_E_ARG (-1)         /* bad argument / malformed scene */
#define RTMI_E_DEVICE (-2)      /* no usable gfx950 device, or a HIP call failed */
#define RTMI_E_UNSUPPORTED (-3) /* record type / option not available on the GPU path */
#define RTMI_E_NOMEM (-4)
#define RTMI_E_STATE (-5)       /* handle used after destroy, wrong context, ... */

enum { RTMI_PRIM_SPHERE = 0, RTMI_PRIM_UVSPHERE = 1, RTMI_PRIM_MOVING = 2,
       /* section 8(f3), via rtmi_scene_create_ex: */
       RTMI_PRIM_RECT_XY = 3, RTMI_PRIM_RECT_XZ = 4, RTMI_PRIM_RECT_YZ = 5, RTMI_PRIM_TRIANGLE = 6,
       /* ConstantMedium (hitable.clj:516): prim_geom = density, first boundary primitive, boundary primitive count; prim_mat =
        * its phase function (RTMI_MAT_ISOTROPIC).  The primitives of a medium's boundary come AFTER all world primitives and
        * carry RTMI_PRIM_BOUNDARY in their kind: they exist only for the medium (a record that is both in the world and a
        * boundary, scene.clj:434,466-469, is listed twice). */
       RTMI_PRIM_MEDIUM = 7, RTMI_PRIM_BOUNDARY = 16 };
enum { RTMI_XFORM_TRANSLATE = 0, RTMI_XFORM_ROTATE_Y = 1 };
enum { RTMI_MAT_LAMBERTIAN = 0, RTMI_MAT_METAL = 1, RTMI_MAT_DIELECTRIC = 2, RTMI_MAT_DIFFUSE_LIGHT = 3,
       RTMI_MAT_ISOTROPIC = 4 /* shader.clj:129, only as a ConstantMedium's phase function */ };
enum { RTMI_TEX_CONSTANT = 0, RTMI_TEX_UVGRADIENT = 1, RTMI_TEX_CHECKER = 2,
       /* section 8(f4), texture.clj:60-138: PerlinNoise (tex_param = scale), PerlinTurbulence / Marble (scale, depth),
        * FlipTextureU / FlipTextureV (tex_child[0] = wrapped texture), ImageMap (tex_param[0] = image index) */
       RTMI_TEX_PERLIN_NOISE = 3, RTMI_TEX_PERLIN_TURB = 4, RTMI_TEX_MARBLE = 5, RTMI_TEX_FLIP_U = 6, RTMI_TEX_FLIP_V = 7,
       RTMI_TEX_IMAGE = 8 };
enum { RTMI_CAM_PINHOLE = 0, RTMI_CAM_THINLENS = 1 };
enum { RTMI_F64 = 0, RTMI_F32 = 1 };             /* arithmetic the kernels compute in */
enum { RTMI_ACCEL_FLAT = 0, RTMI_ACCEL_BVH = 1 }; /* Hitlist scan (hitable.clj:15-26) | bvh-node descent (hitable.clj:97-123) */

#define RTMI_PRIM_STRIDE 9
#define RTMI_TEX_STRIDE 12
#define RTMI_TILE 8          /* framebuffer tiles are RTMI_TILE x RTMI_TILE pixels */
#define RTMI_TILE_PIXELS 64
#define RTMI_SEG_REC 12      /* doubles per logged path segment: prim, t, p.xyz, n.xyz, next dir.xyz, scattered? */

#define RTMI_FLAG_TIMING 1u  /* record HIP events around the trace kernel and its reduction (rtmi_last_trace_ms, rtmi_last_reduce_ms) */

typedef struct rtmi_ctx rtmi_ctx;
typedef struct rtmi_scene rtmi_scene;

/* ---- library / context ---------------------------------------------------------------- */
const char *rtmi_last_error(void);
const char *rtmi_backend_name(void); /* "hip-gfx950" */
int rtmi_version(void);

/* Binds a context to HIP device `device` (one process per GPU: pass LOCAL_RANK; one process for the node: one context per
 * device, see rtmi_render_multi). */
int rtmi_init(int device, uint32_t flags, rtmi_ctx **out_ctx);
int rtmi_shutdown(rtmi_ctx *ctx);
/* knobs: "accel" (RTMI_ACCEL_*; default RTMI_ACCEL_BVH -- bit-identical to the flat scan), "count_traversal" (0/1: the next
 * renders run the counting instantiation of the BVH kernel, see rtmi_last_traversal_counters), "suspend_lanes" (0..64, default 8: the
 * BVH traversal of a wave stops descending / hands the wave back when fewer lanes than this are still descending / in the tree, and the
 * parked lanes resume in the next trip; 0 = the plain loop; the image does not depend on it), "flat_below" (default 24: a scene of rectangles /
 * triangles / instances / media / f4 textures with fewer primitives than this renders through the flat scan even under RTMI_ACCEL_BVH -- same image, a
 * tree over so few primitives only costs; 0 = never; renders with "count_traversal" always walk the tree), "workspace_bytes" (sample-buffer budget, default 64 GiB, allocated as needed: a frame is rendered in as many sample passes as
 * it takes), "blocks_per_cu" (cap on resident trace workgroups per CU; the launch never exceeds what stays resident),
 * "scan_variant" (flat scan: 0 LDS literal, 1 LDS pipelined, 2 scalar cache, 3 scalar cache + FP32 cull = default),
 * "lds_tile_bytes" (LDS variants), "timing" (0/1 = RTMI_FLAG_TIMING); test hooks: "test_fail_next_render" (the context's next render
 * fails before launching anything), "test_fail_allocs" (its next n sample-buffer allocations fail).
 * A context owns its workspace and work queue and is not re-entrant; for several frames in flight on one GPU use one context
 * (and one stream) per frame slot. */
int rtmi_set_option(rtmi_ctx *ctx, const char *name, int64_t value);
int rtmi_device_info(rtmi_ctx *ctx, int32_t *compute_units, int32_t *lds_bytes_per_cu, int64_t *hbm_bytes, char *arch, int32_t arch_len);

/* ---- scene ------------------------------------------------------------------------------- */
/* Replaces building the world the reference's Hitlist.hit? walks (hitable.clj:15-26) and the
 * camera record get-ray reads (camera.clj:8-16,35-48): uploads the flat arrays to HBM (SoA). */
int rtmi_scene_create(rtmi_ctx *ctx,
                      int32_t n_prims, const int32_t *prim_kind, const double *prim_geom, const int32_t *prim_mat,
                      int32_t n_mats, const int32_t *mat_kind, const int32_t *mat_tex, const double *mat_param,
                      int32_t n_tex, const int32_t *tex_kind, const double *tex_param, const int32_t *tex_child,
                      int32_t cam_kind, const double *cam, rtmi_scene **out_scene);
/* The same plus the instancing records of hitable.clj:269-511, 548-581 (Cornell-box class scenes):
 *   prim_kind[i] may also be RTMI_PRIM_RECT_XY/XZ/YZ (hitable.clj:269/301/333; prim_geom = a0, b0, a1, b1, k: the two in-plane
 *   extents in the order of the record's fields, then the plane's coordinate) or RTMI_PRIM_TRIANGLE (hitable.clj:548;
 *   prim_geom = v0.xyz, v1.xyz, v2.xyz);
 *   prim_flip[i]      parity of the FlipNormals wrappers (hitable.clj:375) around primitive i;
 *   prim_xform[i*2..] first index and count of primitive i's Translate / RotateY wrappers (hitable.clj:391, 410) in the
 *                     xform table, OUTERMOST FIRST;  xform_kind[k] = RTMI_XFORM_*;
 *   xform_param[k*3..] Translate: offset.xyz | RotateY: sin-theta, cos-theta, 0 (the record's fields, hitable.clj:410).
 * Box (hitable.clj:491) flattens to its six rectangles.  ConstantMedium (hitable.clj:516-541) draws its random number inside
 * hit?: by default media are evaluated in primitive-index order with the un-narrowed (t-min, t-max) of the reference's bvh-node descent;
 * a world that IS a Hitlist holding media is declared with rtmi_scene_set_media_mode(RTMI_MEDIA_HITLIST); at most 16 media per scene.  Scenes that use
 * any of this are rendered by the FP64 kernels only (RTMI_F32 -> RTMI_E_UNSUPPORTED). */
int rtmi_scene_create_ex(rtmi_ctx *ctx,
                         int32_t n_prims, const int32_t *prim_kind, const double *prim_geom, const int32_t *prim_mat,
                         int32_t n_mats, const int32_t *mat_kind, const int32_t *mat_tex, const double *mat_param,
                         int32_t n_tex, const int32_t *tex_kind, const double *tex_param, const int32_t *tex_child,
                         int32_t cam_kind, const double *cam,
                         const int32_t *prim_flip, const int32_t *prim_xform,
                         int32_t n_xforms, const int32_t *xform_kind, const double *xform_param, rtmi_scene **out_scene);
/* The namespace-level tables of perlin.clj:6-17 as scene data (the reference fills them from the unseeded global RNG when
 * the namespace loads): vectors[256*3] = random-vectors (unit vectors), perm[3*256] = perm-x, perm-y, perm-z (each a
 * permutation of 0..255).  Required before rendering a scene that holds a Perlin texture. */
int rtmi_scene_set_perlin(rtmi_scene *scene, const double *vectors, const int32_t *perm);
/* The pixels ImageMap (texture.clj:126-133) samples: n images, wh[2*i] = width, height of image i, rgb = the images'
 * rows (top row first, RGB bytes) concatenated.  Replaces imagez load-image / get-pixel (texture.clj:76,138). */
int rtmi_scene_set_images(rtmi_scene *scene, int32_t n_images, const int32_t *wh, const uint8_t *rgb);
/* The order in which, and how often, the reference's descent calls hit? of the scene's ConstantMedium primitives per ray:
 * calls[k] = index of a RTMI_PRIM_MEDIUM primitive, n_calls <= 32.  Default: every medium once, ascending index.  (make-bvh
 * stores a lone item as bvh-node(L, L), hitable.clj:113-114, and bvh-node.hit? evaluates both children: a medium there is
 * asked twice, draws two random numbers and the nearer scattering point wins.) */
int rtmi_scene_set_media_calls(rtmi_scene *scene, int32_t n_calls, const int32_t *calls);
/* How a ConstantMedium's hit? is called (it draws its random number INSIDE hit?, hitable.clj:529, so the t-max it is handed matters):
 *   RTMI_MEDIA_DESCENT (default)  the world is a make-bvh tree: bvh-node.hit? hands both children the un-narrowed (t-min, t-max)
 *                                 (hitable.clj:99-105); the media are evaluated after the surfaces, in call order (rtmi_scene_set_media_calls);
 *   RTMI_MEDIA_HITLIST            the world is a Hitlist (nested Hitlists / Boxes / instance wrappers spliced in item order, no bvh-node
 *                                 anywhere): Hitlist.hit? hands every item the t-max narrowed by the items BEFORE it (hitable.clj:15-26), so
 *                                 a medium is evaluated at its place in the list with the closest hit so far as its t-max; the primitives
 *                                 must be in list order and every medium is called once, in ascending index order.
 * A world that mixes the two (a Hitlist holding media below a bvh-node) is not supported: the flatteners raise "unsupported on GPU path". */
enum { RTMI_MEDIA_DESCENT = 0, RTMI_MEDIA_HITLIST = 1 };
int rtmi_scene_set_media_mode(rtmi_scene *scene, int32_t mode);
/* The general form (round 4): Hitlists holding media BELOW bvh-nodes.  calls = the media call sequence as for rtmi_scene_set_media_calls; narrow_from[k] = the
 * first primitive of the Hitlist items that stand before call k's medium in its own (possibly nested) Hitlist -- Hitlist.hit? hands the medium the closest hit
 * among primitives [narrow_from[k], calls[k]) as its t-max (hitable.clj:15-26), whatever the bvh-nodes above that Hitlist do (they pass the interval on
 * un-narrowed, hitable.clj:99-105); narrow_from[k] = calls[k]: no item narrows it (a medium reached through bvh-nodes only).  The items of a Hitlist are
 * contiguous in the flattened order; calls that share a narrowing list come in list order.  Replaces both calls above for such worlds; still unsupported: a
 * bvh-node BETWEEN a narrowing Hitlist and its medium, a medium inside a medium's boundary. */
int rtmi_scene_set_media_calls_narrowed(rtmi_scene *scene, int32_t n_calls, const int32_t *calls, const int32_t *narrow_from);
/* HBM bytes the scene occupies (everything its creation uploaded: records, tree, tables) -- bench.py's `upload_bytes` */
int rtmi_scene_device_bytes(rtmi_scene *scene, int64_t *out_bytes);
int rtmi_scene_destroy(rtmi_scene *scene);

/* ---- the hot path --------------------------------------------------------------------------- */
/* Replaces core.clj:100-108 for the output region [x0,x1) x [y0,y1): for each pixel, ns jittered
 * samples of `color` (core.clj:17-41, depth as core.clj:20,45), mean, gamma 2, 8-bit (core.clj:52-57).
 * out_linear: (y1-y0)*(x1-x0)*3 doubles, the per-pixel mean BEFORE sqrt (for RMS parity), may be NULL;
 * out_rgb8: same shape uint8, trunc(min(255.99, 255.99*sqrt(mean))), NaN -> 0, may be NULL;
 * out_counters: {total-rays (core.clj:24), total-pixels (core.clj:47)} (metrics.clj:8-9) of the region's pixels, may be NULL.
 * Only the 8x8 tiles that intersect the region are rendered.
 * Every random draw is the next value of the counter stream keyed (seed, j*nx+i, sample). */
int rtmi_render(rtmi_scene *scene, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed, int32_t precision,
                int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                double *out_linear, uint8_t *out_rgb8, uint64_t *out_counters);

/* Same path with every buffer resident in HBM (device pointers), launched on `stream` (a hipStream_t); asynchronous.
 * stream = NULL means the CONTEXT'S OWN stream (created hipStreamNonBlocking), NOT the HIP default stream: work the caller
 * has queued on any other stream -- including the legacy default stream, whose handle is also 0 -- is not ordered with it.
 * A caller whose buffers are produced / consumed on another stream passes that stream's handle here (for the legacy
 * default stream: hipStreamLegacy) or brackets the call with events.  A context's workspace belongs to one stream at a time.
 * d_out_linear holds doubles for both precisions. */
int rtmi_render_device(rtmi_scene *scene, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed, int32_t precision,
                       void *d_out_linear, void *d_out_rgb8, void *d_out_counters, void *stream);

/* Tile-partitioned form for one-process-per-GPU rendering (the reference's tiled-coords,
 * core.clj:59-71, becomes 8x8 tiles dealt round-robin): renders global tiles
 * tile_first, tile_first+tile_stride, ... (row-major tile index over ceil(nx/8) x ceil(ny/8)) into
 * d_tiles_linear[k][64][3] doubles (per-pixel mean, tile-major; pixels outside the image are 0).
 * The number of local tiles is rtmi_local_tiles(nx, ny, tile_first, tile_stride). */
int rtmi_render_tiles_device(rtmi_scene *scene, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed, int32_t precision,
                             int32_t tile_first, int32_t tile_stride,
                             void *d_tiles_linear, void *d_out_counters, void *stream);
int32_t rtmi_local_tiles(int32_t nx, int32_t ny, int32_t tile_first, int32_t tile_stride);

/* After the gather: d_gathered[r][k][64][3] (r < world, k < tiles_per_rank, rank r's k-th tile is global
 * tile r + k*world) -> dense row-major frame (doubles, may be NULL) + 8-bit frame (may be NULL). */
int rtmi_assemble_device(rtmi_ctx *ctx, int32_t nx, int32_t ny, int32_t world, int32_t tiles_per_rank,
                         const void *d_gathered, void *d_out_linear, void *d_out_rgb8, void *stream);

/* ---- one host process, several GPUs ---------------------------------------------------------------
 * The reference's host is ONE JVM whose render loop fans out over a thread pool (cp/upmap, core.clj:100-108); the same
 * single process reaches the 8 GPUs of a node through these entries: one context per device (rtmi_init), the scene
 * replicated onto each (rtmi_scene_clone; it is tiny), the framebuffer's 8x8 tiles dealt round-robin (replica r renders
 * global tiles r, r+n, ...: the reference's tiled-coords chunks, core.clj:59-71), ONE gather over xGMI to replica 0's
 * device -- ncclGather on a communicator the library creates with ncclCommInitAll and owns (librccl is opened on first
 * use; a single-GPU host never loads it) -- and replica 0 un-tiles / quantises.  Pixels are independent and the stream key is
 * the global pixel index: the image is bit-identical to the single-device render.
 * Replicas that share a device (to rehearse the control flow on a one-GPU host) are gathered by device copies instead, and so
 * is a host on which librccl cannot be opened or initialised, or whose gather failed once (one line on stderr).
 * RTMI_MULTI_GATHER=rccl: RCCL or an error, never a substitution -- and a ONE-replica call then goes through the whole RCCL path too
 * (dlopen, ncclCommInitAll of one rank, grouped in-place ncclGather): that is how the path is executed on a one-GPU host (status:
 * the one-rank path runs in the GPU test suite; a communicator over several devices has not run yet -- no multi-GPU host so far).
 * RTMI_MULTI_GATHER=copy forces the copies.  rtmi_last_gather_path says which one ran.
 * On an error after the first replica's launch every replica stream touched so far is synchronised before the call returns. */
/* Replicates `scene` onto `ctx` (another device): the library copied the caller's arrays at creation. */
int rtmi_scene_clone(rtmi_scene *scene, rtmi_ctx *ctx, rtmi_scene **out_scene);
/* Replaces core.clj:100-108 on n devices.  scenes[r] = replica r (each on its own context).  Host buffers as rtmi_render
 * (whole frame); out_counters = {total-rays summed over the replicas, total-pixels}. */
int rtmi_render_multi(int32_t n, rtmi_scene *const *scenes, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed,
                      int32_t precision, double *out_linear, uint8_t *out_rgb8, uint64_t *out_counters);
/* The same, asynchronous, outputs resident on replica 0's device (ordered on replica 0's context stream; every replica
 * renders on its own context stream). */
int rtmi_render_multi_device(int32_t n, rtmi_scene *const *scenes, int32_t nx, int32_t ny, int32_t ns, int32_t depth, uint64_t seed,
                             int32_t precision, void *d_out_linear, void *d_out_rgb8, void *d_out_counters);
/* How the last rtmi_render_multi* on replica 0's context moved the replicas' records to replica 0's device. */
enum { RTMI_GATHER_NONE = 0,        /* one replica: nothing to move */
       RTMI_GATHER_SAME_DEVICE = 1, /* replicas share replica 0's device (rehearsal on a one-GPU host): device-to-device copies */
       RTMI_GATHER_PEER_COPY = 2,   /* distinct devices, hipMemcpyPeerAsync (RTMI_MULTI_GATHER=copy, or RCCL unusable on this host) */
       RTMI_GATHER_RCCL = 3 };      /* ONE grouped ncclGather on the library's communicator set */
int rtmi_last_gather_path(rtmi_ctx *ctx0, int32_t *path);
/* Can the RCCL library `soname` be opened and does it export the entry points the in-library gather binds (ncclCommInitAll,
 * ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclGather, ncclGetErrorString)?  soname = NULL: the names the gather itself tries
 * ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", or $RTMI_RCCL_LIB).  Touches no device.  RTMI_E_DEVICE + the loader's
 * message when not. */
int rtmi_rccl_probe(const char *soname);
/* *idle = 1 when nothing is pending on the context's own stream (hipStreamQuery). */
int rtmi_stream_idle(rtmi_ctx *ctx, int32_t *idle);
/* Sample passes (trace-kernel launches) the context's most recent render took: the per-sample colour buffer is sized by option
 * "workspace_bytes", by the HBM that is free when it has to grow (at most 80 % of it) and, if the allocation still fails, by halving. */
int rtmi_last_passes(rtmi_ctx *ctx, int32_t *passes);
/* RTMI_ACCEL_* the context's most recent render actually ran: option "accel" as set, except that a small mixed-kind scene (option
 * "flat_below") is scanned even when the tree was asked for.  RTMI_E_STATE before the first render. */
int rtmi_last_accel(rtmi_ctx *ctx, int32_t *accel);
/* Milliseconds between the end of replica 0's own render and the end of the gather of the last rtmi_render_multi* on
 * replica 0's context (HIP events on its stream): the transfer plus the wait for the slowest replica. */
int rtmi_last_gather_ms(rtmi_ctx *ctx0, double *ms);

/* The reference's third counter, metrics.clj:10 `aabb.intersection.total` (incremented once per AABB.hit?, hitable.clj:39),
 * for the device's own tree: *out_aabb_tests = AABB slab tests, *out_prim_tests = exact primitive tests (leaves + the
 * primitives kept out of the tree) of the context's most recent render.  Needs option "count_traversal" = 1 before that
 * render (a separate kernel instantiation: the default kernel does not pay for the counting); synchronises on the render's
 * stream.  The numbers depend on the tree the library built, not on the reference's random-axis tree. */
int rtmi_last_traversal_counters(rtmi_ctx *ctx, uint64_t *out_aabb_tests, uint64_t *out_prim_tests);

/* Total milliseconds of the trace-kernel launches of every render on this context since the previous call
 * (HIP events recorded on the launch stream around each launch; needs RTMI_FLAG_TIMING; synchronises on the
 * last event; at most 8192 launches are kept).  *launches = kernel launches covered.  Resets the window. */
int rtmi_last_trace_ms(rtmi_ctx *ctx, double *ms, int32_t *launches);
/* The in-order sample reductions (reduce_kernel, core.clj:52-53) of the measurement window the last rtmi_last_trace_ms call closed:
 * sum of their durations (from each trace kernel's end event to the event after its reduction) and their number. */
int rtmi_last_reduce_ms(rtmi_ctx *ctx, double *ms, int32_t *launches);

/* ---- probes: the same device functions the render kernel runs, one protocol call at a time ---- */
/* Hitable.hit? of the world for n rays {o.xyz, d.xyz, time}; out[n][11] = hit?, prim, t, p.xyz, normal.xyz, u, v */
int rtmi_probe_hit(rtmi_scene *scene, int32_t precision, int32_t n, const double *rays, double t_min, double t_max, double *out);
/* `color` (core.clj:17-41) of n explicit rays with their own stream keys; log (may be NULL):
 * [n][max_seg][RTMI_SEG_REC]; out_nlog[n] segments logged. */
int rtmi_probe_paths(rtmi_scene *scene, int32_t precision, int32_t n, const double *rays, const uint64_t *keys, uint64_t ctr0,
                     int32_t depth, double *out_rgb, uint64_t *out_nseg, double *log, int32_t max_seg, int32_t *out_nlog);
/* Camera.get-ray for n (u,v) pairs; out[n][8] = o.xyz, d.xyz, time, draws consumed */
int rtmi_probe_camera(rtmi_scene *scene, int32_t precision, int32_t n, const double *uv, const uint64_t *keys, double *out);
/* Texture.sample of texture `tex` for n {u, v, p.xyz}; out[n][3] */
int rtmi_probe_texture(rtmi_scene *scene, int32_t precision, int32_t tex, int32_t n, const double *uvp, double *out);
/* Shader.scatter of material `mat` for n rays and hit records {p.xyz, normal.xyz, u, v};
 * out[n][9] = scattered?, dir.xyz, attenuation.rgb, time, draws consumed */
int rtmi_probe_scatter(rtmi_scene *scene, int32_t precision, int32_t mat, int32_t n, const double *rays, const double *hits,
                       const uint64_t *keys, double *out);
/* The counter stream: out_bits[k] = 64 raw bits, out_real[k] = the uniform in [0,1) for draw index d0+k of `key` */
int rtmi_probe_rng(rtmi_ctx *ctx, int32_t precision, uint64_t key, uint64_t d0, int32_t n, uint64_t *out_bits, double *out_real);
/* sample key of (seed, pixel index j*nx+i, sample) -- pure host function, no device needed */
uint64_t rtmi_sample_key(uint64_t seed, uint64_t pixel, uint64_t sample);
/* correctly-rounded device arithmetic check: out[k] = {a/b, sqrt(|a|), a*b+c unfused} */
int rtmi_probe_arith(rtmi_ctx *ctx, int32_t n, const double *abc, double *out);
/* the path's own FP64 helpers, one thread per triple (a, b, c):  out[n_slots k + j], j < n_slots <= RTMI_PROBE_MATH_SLOTS =
 * { 0 sqrt(a) (fast path for finite arguments >= 2^-767, libm otherwise), 1 atan2(a, b), 2 asin(a), 3, 4 u and v of get-sphere-uv for the unit normal
 * (a, b, c) (hitable.clj:128-139), 5 a / b through the per-ray reciprocal of the sphere roots (t-min / t-max decide with b whether the lane takes
 * it), 6 a / (2 pi) by the constant-divisor form, 7 1.0 if the lane took the reciprocal path else 0.0, 8 the traversal's float bound of a closest hit at
 * t = c: a float >= c within two ulps (FLT_MAX beyond the float range), 9 log(a) as ConstantMedium's free-flight distance evaluates it for a draw a in
 * [0, 1), 10 a / b through the refined reciprocal of a SIGNED divisor (a rectangle's t = (k - o) / d), 11 1.0 if slot 10 took the reciprocal path }.
 * The caller's buffer holds n * n_slots doubles: the slot count is an argument so that a host built against an older header is never overrun.
 * rtmi_probe_math is the entry as first published: the first EIGHT slots (library version 203 wrote nine; from 204 on it is eight again). */
#define RTMI_PROBE_MATH_SLOTS 12
int rtmi_probe_math2(rtmi_ctx *ctx, int32_t n, const double *abc, double tmin, double tmax, int32_t n_slots, double *out);
int rtmi_probe_math(rtmi_ctx *ctx, int32_t n, const double *abc, double tmin, double tmax, double *out);

/* test hook, host arithmetic only (no device): the IEEE half (bits) at or beyond x in the given direction -- what the tree's half-plane node records are
 * rounded with (up != 0: the smallest half >= x, else the largest half <= x; +-inf beyond the half range) */
int rtmi_test_build_tree(int32_t n, const double *geom, const double *cam, int32_t threads, uint64_t *out_hash, int32_t *out_info, double *out_ms);
/* ^ test hook, host code only: the device's tree and entry grid over n spheres (geom[n][4] = cx cy cz r) exactly as rtmi_scene_create builds them -- on the
 * library's team of build threads, or with threads = 1 on the calling thread alone; out_hash = FNV-1a of the node array and the grid's root codes (the same
 * for any thread count), out_info[4] = node records, depth, grid cells per side, big primitives; out_ms = the build's wall time */
int rtmi_test_half_outward(double x, int32_t up); /* x: a float value (host scalars are doubles at this boundary) */

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
