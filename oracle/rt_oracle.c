/*
 * rt_oracle.c -- CPU restatement of raytrace-clj's per-pixel Monte-Carlo sampling path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (raytrace_clj_amd/, include/, bench's
 * GPU leg) may link, load or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * What it restates (all paths relative to /root/reference):
 *   src/raytrace_clj/core.clj:17-57       color, pixel
 *   src/raytrace_clj/util.clj:5-22,32-52  vec3, ray, point-at-parameter, rejection samplers
 *   src/raytrace_clj/camera.clj:8-66      PinholeCamera, ThinLensCamera (+ ctors)
 *   src/raytrace_clj/hitable.clj:15-26    Hitlist
 *   src/raytrace_clj/hitable.clj:36-48,87-123  AABB, make-surrounding-bbox, bvh-node (equivalence tests)
 *   src/raytrace_clj/hitable.clj:128-264  get-sphere-uv, UVSphere, Sphere, MovingSphere
 *   src/raytrace_clj/shader.clj:6-124     reflect, refract, schlick, Lambertian, Metal, Dielectric, DiffuseLight
 *   src/raytrace_clj/texture.clj:14-55    Constant, UVGradient, Checkerboard
 *
 * Third-party arithmetic that is NOT under /root/reference (un-vendored Maven deps,
 * project.clj:6-16): net.mikera/core.matrix 0.52.0 + net.mikera/vectorz-clj 0.44.0.  Their
 * published semantics are restated here as: element-wise add/sub/mul with scalar broadcast,
 * variadic add/mul as a LEFT fold, dot = ((x0*y0 + x1*y1) + x2*y2), magnitude = sqrt(dot),
 * normalise = v * (1.0/magnitude) when magnitude > 0, lerp(a,b,f) = a*(1-f) + b*f,
 * ereduce/emap in element order.  java.lang.Math sqrt and IEEE + - * / are correctly rounded, so
 * they are bit-reproducible here (this file is built with -ffp-contract=off); Math.sin/asin/
 * atan2/pow/tan are only specified to 1-2 ulp and are replaced by the host libm.
 *
 * PARITY STATUS: the reference cannot run in this environment (no JVM) and draws every random
 * number from an unseeded global Math.random(), so image-level parity against the genuine
 * Clojure path is UNPINNED.  This restatement is pinned by the known-answer data that the
 * reference's own tests hold (test/raytrace_clj/util_test.clj:44-49,
 * test/raytrace_clj/hitable_test.clj:23-141) plus analytic values derived from the cited
 * formulas -- see tests/test_oracle_kat.py and tests/golden/.
 *
 * Reproducible randomness is something this build introduces: every (rand) of the reference
 * becomes the next draw of a counter-based stream keyed on (seed, pixel j*nx+i, sample s), in
 * exactly the order the reference consumes draws (SURVEY.md section 8a "RNG order").
 *
 * Build:  make -C oracle      (gcc, -ffp-contract=off; REAL=double by default)
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef RTO_REAL
#define RTO_REAL double
#endif
typedef RTO_REAL real;

#if defined(RTO_FLOAT)
#define R_SQRT sqrtf
#define R_SIN sinf
#define R_ASIN asinf
#define R_ATAN2 atan2f
#define R_POW powf
#else
#define R_SQRT sqrt
#define R_SIN sin
#define R_ASIN asin
#define R_ATAN2 atan2
#define R_POW pow
#endif

#define RTO_API __attribute__((visibility("default")))

/* ---- flat scene description (same arrays the product's C-ABI takes; include/rtmi.h) ---- */
enum { PRIM_SPHERE = 0, PRIM_UVSPHERE = 1, PRIM_MOVING = 2 };
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3, MAT_ISOTROPIC = 4 };
enum { TEX_CONSTANT = 0, TEX_UVGRADIENT = 1, TEX_CHECKER = 2,
       TEX_PERLIN_NOISE = 3, TEX_PERLIN_TURB = 4, TEX_MARBLE = 5, TEX_FLIP_U = 6, TEX_FLIP_V = 7, TEX_IMAGE = 8 };
enum { CAM_PINHOLE = 0, CAM_THINLENS = 1 };
#define PRIM_STRIDE 9 /* c0xyz, radius, c1xyz, t0, t1 */
#define TEX_STRIDE 12 /* constant: rgb; gradient: co cu cv cuv; checker: scale */

typedef struct {
    int32_t n_prims;
    const int32_t *prim_kind;
    const double *prim_geom; /* n_prims * PRIM_STRIDE */
    const int32_t *prim_mat;
    int32_t n_mats;
    const int32_t *mat_kind;
    const int32_t *mat_tex;
    const double *mat_param; /* fuzz | ri */
    int32_t n_tex;
    const int32_t *tex_kind;
    const double *tex_param; /* n_tex * TEX_STRIDE */
    const int32_t *tex_child; /* n_tex * 2 */
    int32_t cam_kind;
    const double *cam; /* 24: origin lleft horiz vert u v w aperture t0 t1 */
    /* optional NESTED world (the reference's records as they are built, wrappers and bvh-nodes included); when
     * n_nodes > 0 it replaces the flat primitive list above.  node_a[n][3] = child|left|first, right|count, material;
     * node_d[n][12] = geometry; node_prim[n] = index the product's flattener gives this leaf (for logs), else -1. */
    int32_t n_nodes;
    const int32_t *node_kind;
    const int32_t *node_a;
    const double *node_d;
    const int32_t *node_prim;
    const int32_t *node_children; /* Hitlist items */
    int32_t root;
    /* texture.clj:60-138 + perlin.clj: the namespace-level tables (perlin.clj:6-17; SEEDED here) and ImageMap pixels */
    const double *perlin_vec;   /* [256][3] random-vectors */
    const int32_t *perlin_perm; /* [3][256] perm-x, perm-y, perm-z */
    int32_t n_images;
    const int32_t *image_wh;    /* [n][2] */
    const int64_t *image_off;   /* [n] byte offset into image_rgb */
    const uint8_t *image_rgb;   /* rows top-down, RGB */
} rto_scene;
enum { N_SPHERE = 0, N_UVSPHERE = 1, N_MOVING = 2, N_RECT_XY = 3, N_RECT_XZ = 4, N_RECT_YZ = 5, N_TRIANGLE = 6,
       N_FLIP = 7, N_TRANSLATE = 8, N_ROTATE_Y = 9, N_HITLIST = 10, N_BOX = 11, N_BVH = 12, N_MEDIUM = 13 };

typedef struct { real x, y, z; } v3;
typedef struct { v3 o, d; real time; } ray_t;
typedef struct { real t; v3 p; real u, v; v3 n; int32_t prim; int32_t mat; } hit_t;

/* ---- util.clj:5-11 vec3 and the core.matrix element-wise ops it is used with ---- */
static inline v3 V(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, real s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline real vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline real vmag(v3 a) { return R_SQRT(vdot(a, a)); }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vectorz AVector.normalise: d = magnitude(); if (d > 0) multiply(1.0/d) */
static inline v3 vnormalise(v3 a) {
    real d = vmag(a);
    if (d > (real)0) return vscale(a, (real)1.0 / d);
    return a;
}
/* core.matrix lerp: a*(1-f) + b*f   (hitable.clj:219-222 center-at-time) */
static inline v3 vlerp(v3 a, v3 b, real f) { return vadd(vscale(a, (real)1.0 - f), vscale(b, f)); }

/* util.clj:18-22 point-at-parameter: direction*t + origin */
static inline v3 point_at(const ray_t *r, real t) { return vadd(vscale(r->d, t), r->o); }

/* ---- counter-based random stream (introduced by this build; replaces clojure.core/rand) ---- */
#define GOLD 0x9E3779B97F4A7C15ULL
static inline uint64_t mix64(uint64_t z) { /* the stream's mixer ("degski64": shifts of 32 cost a 32-bit ALU nothing), DESIGN.md section 2 */
    z ^= z >> 32; z *= 0xD6E8FEB86659FD93ULL;
    z ^= z >> 32; z *= 0xD6E8FEB86659FD93ULL;
    z ^= z >> 32; return z;
}
static inline uint64_t splitmix64_fin(uint64_t z) { /* splitmix64 finaliser: the axis choice of this file's own make-bvh (below), as in rounds 1-2 */
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31; return z;
}
typedef struct { uint64_t key; uint64_t ctr; } rng_t;
RTO_API uint64_t rto_sample_key(uint64_t seed, uint64_t pix, uint64_t s) {
    return mix64(mix64(seed ^ (GOLD * (pix + 1))) + 0xD1B54A32D192ED03ULL * (s + 1));
}
RTO_API uint64_t rto_draw_bits(uint64_t key, uint64_t d) { return mix64(key + GOLD * (d + 1)); }
static inline real rng_next(rng_t *g) {
    uint64_t z = rto_draw_bits(g->key, g->ctr++);
#if defined(RTO_FLOAT)
    return (real)(z >> 40) * (real)(1.0 / 16777216.0);
#else
    return (real)(z >> 11) * (real)(1.0 / 9007199254740992.0);
#endif
}
RTO_API double rto_draw(uint64_t key, uint64_t d) {
    rng_t g = {key, d};
    return (double)rng_next(&g);
}

/* util.clj:32-41 rand-in-unit-disk: draw x then y; retry while p.p >= 1.0 */
static v3 rand_in_unit_disk(rng_t *g) {
    for (;;) {
        real x = (real)2.0 * rng_next(g) - (real)1.0;
        real y = (real)2.0 * rng_next(g) - (real)1.0;
        v3 p = V(x, y, 0);
        if (!(vdot(p, p) >= (real)1.0)) return p;
    }
}
/* util.clj:43-52 rand-in-unit-sphere: draw x, y, z; retry while p.p >= 1.0 */
static v3 rand_in_unit_sphere(rng_t *g) {
    for (;;) {
        real x = (real)2.0 * rng_next(g) - (real)1.0;
        real y = (real)2.0 * rng_next(g) - (real)1.0;
        real z = (real)2.0 * rng_next(g) - (real)1.0;
        v3 p = V(x, y, z);
        if (!(vdot(p, p) >= (real)1.0)) return p;
    }
}

/* ---- scene accessors ---- */
static inline v3 ld3(const double *p) { return V((real)p[0], (real)p[1], (real)p[2]); }

/* ---- perlin.clj:19-64 ---- */
static real perlin_noise(const rto_scene *sc, v3 p) {
    /* noise (45-50): ijk = (int (Math/floor p)), uvw = p - ijk */
    const real fi = (real)floor((double)p.x), fj = (real)floor((double)p.y), fk = (real)floor((double)p.z);
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    const real u = p.x - fi, v = p.y - fj, w = p.z - fk;
    /* perlin-interp (31-43): hermite weights u*u*(3 - 2u) */
    const real uu = (u * u) * ((real)3.0 - (real)2.0 * u), vv = (v * v) * ((real)3.0 - (real)2.0 * v), ww = (w * w) * ((real)3.0 - (real)2.0 * w);
    real acc = 0;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                /* perlin-coefficients (19-29) */
                const int idx = sc->perlin_perm[(i + di) & 255] ^ sc->perlin_perm[256 + ((j + dj) & 255)] ^ sc->perlin_perm[512 + ((k + dk) & 255)];
                const v3 c = ld3(sc->perlin_vec + (size_t)idx * 3);
                const v3 wv = V(u - (real)di, v - (real)dj, w - (real)dk);
                const real A = (real)di * uu + ((real)1.0 - (real)di) * ((real)1.0 - uu);
                const real B = (real)dj * vv + ((real)1.0 - (real)dj) * ((real)1.0 - vv);
                const real C = (real)dk * ww + ((real)1.0 - (real)dk) * ((real)1.0 - ww);
                const real term = ((A * B) * C) * vdot(wv, c);
                acc = (di | dj | dk) ? acc + term : term; /* (reduce + ...) */
            }
    return acc;
}
static real perlin_turbulence(const rto_scene *sc, v3 p, int depth) { /* perlin.clj:52-64 */
    real acc = 0, w = (real)1.0;
    v3 pt = p;
    for (int i = 0; i < depth; ++i) {
        acc = acc + w * perlin_noise(sc, pt);
        pt = vscale(pt, (real)2.0);
        w = w / (real)2.0;
    }
    return acc < 0 ? -acc : acc;
}

/* ---- texture.clj:14-138 ---- */
static v3 tex_sample(const rto_scene *sc, int32_t t, real u, real v, v3 p) {
    for (int guard = 0; guard <= sc->n_tex; ++guard) {
        const double *tp = sc->tex_param + (size_t)t * TEX_STRIDE;
        switch (sc->tex_kind[t]) {
        case TEX_CONSTANT: /* texture.clj:14-16 */
            return ld3(tp);
        case TEX_UVGRADIENT: { /* texture.clj:26-34 */
            v3 co = ld3(tp), cu = ld3(tp + 3), cv = ld3(tp + 6), cuv = ld3(tp + 9);
            v3 a = vadd(vscale(cu, (real)1.0 - u), vscale(co, u));
            v3 b = vadd(vscale(cuv, (real)1.0 - u), vscale(cv, u));
            return vadd(vscale(b, (real)1.0 - v), vscale(a, v));
        }
        case TEX_CHECKER: { /* texture.clj:44-50: (ereduce * (emap sin (mul scale p))) */
            real scale = (real)tp[0];
            real sines = (R_SIN(scale * p.x) * R_SIN(scale * p.y)) * R_SIN(scale * p.z);
            t = (sines < (real)0) ? sc->tex_child[2 * t] : sc->tex_child[2 * t + 1];
            break;
        }
        case TEX_PERLIN_NOISE: { /* texture.clj:60-64: (1,1,1) * 0.5 * (inc (noise (mul scale p))) */
            real c = (real)0.5 * (perlin_noise(sc, vscale(p, (real)tp[0])) + (real)1.0);
            return V(c, c, c);
        }
        case TEX_PERLIN_TURB: { /* texture.clj:74-78 */
            real c = (real)0.5 * (perlin_turbulence(sc, vscale(p, (real)tp[0]), (int)tp[1]) + (real)1.0);
            return V(c, c, c);
        }
        case TEX_MARBLE: { /* texture.clj:88-93: 0.5 * (inc (sin (+ (* scale pz) (* 10.0 (turbulence p depth))))) */
            real c = (real)0.5 * (R_SIN((real)tp[0] * p.z + (real)10.0 * perlin_turbulence(sc, p, (int)tp[1])) + (real)1.0);
            return V(c, c, c);
        }
        case TEX_FLIP_U: u = (real)1.0 - u; t = sc->tex_child[2 * t]; break; /* texture.clj:103-106 */
        case TEX_FLIP_V: v = (real)1.0 - v; t = sc->tex_child[2 * t]; break; /* texture.clj:113-116 */
        case TEX_IMAGE: { /* texture.clj:126-133: i = (int (* u width)), j = (int (* v height)); rgb / 255.0.
                             (u = 1.0 indexes one past the row in the reference and throws; clamped here) */
            const int im = (int)tp[0];
            const int w = sc->image_wh[2 * im], hgt = sc->image_wh[2 * im + 1];
            int i = (int)(u * (real)w), j = (int)(v * (real)hgt);
            i = i < 0 ? 0 : (i >= w ? w - 1 : i); j = j < 0 ? 0 : (j >= hgt ? hgt - 1 : j);
            const uint8_t *px = sc->image_rgb + sc->image_off[im] + ((size_t)j * w + i) * 3;
            return V((real)px[0] / (real)255.0, (real)px[1] / (real)255.0, (real)px[2] / (real)255.0);
        }
        default:
            return V(0, 0, 0);
        }
    }
    return V(0, 0, 0);
}

/* ---- hitable.clj:128-139 get-sphere-uv ---- */
static void sphere_uv(v3 n, real *u, real *v) {
    const real PI = (real)3.141592653589793;
    real phi = R_ATAN2(n.z, n.x);
    real theta = R_ASIN(n.y);
    *u = (real)1.0 - (phi + PI) / ((real)2.0 * PI);
    *v = (theta + PI / (real)2.0) / PI;
}

/* ---- hitable.clj:141-168 (UVSphere), 180-207 (Sphere), 219-252 (MovingSphere) ---- */
static int sphere_hit(const rto_scene *sc, int32_t i, const ray_t *r, real tmin, real tmax, hit_t *h) {
    const double *g = sc->prim_geom + (size_t)i * PRIM_STRIDE;
    int kind = sc->prim_kind[i];
    v3 center = ld3(g);
    real radius = (real)g[3];
    if (kind == PRIM_MOVING) { /* hitable.clj:219-222,227 */
        real t0 = (real)g[7], t1 = (real)g[8];
        center = vlerp(center, ld3(g + 4), (r->time - t0) / (t1 - t0));
    }
    v3 oc = vsub(r->o, center);
    real a = vdot(r->d, r->d);
    real b = (real)2.0 * vdot(oc, r->d);
    real c = vdot(oc, oc) - radius * radius;
    real disc = b * b - ((real)4.0 * a) * c;
    if (disc >= (real)0) {
        real sq = R_SQRT(disc);
        for (int root = 0; root < 2; ++root) {
            real t = root == 0 ? (-b - sq) / ((real)2.0 * a) : (-b + sq) / ((real)2.0 * a);
            if (t > tmin && t < tmax) {
                h->t = t;
                h->p = point_at(r, t);
                h->n = vnormalise(vsub(h->p, center));
                h->u = 0; h->v = 0;
                if (kind == PRIM_UVSPHERE) sphere_uv(h->n, &h->u, &h->v);
                h->prim = i;
                h->mat = sc->prim_mat[i];
                return 1;
            }
        }
    }
    return 0;
}

/* ---- nested world: every record of hitable.clj evaluated as the reference nests them ---- */
RTO_API int rto_aabb_hit(const double *vmin, const double *vmax, const double *o, const double *d, double tmin, double tmax);
static int sphere_like_hit(v3 center0, real radius, int kind, const double *g, const ray_t *r, real tmin, real tmax, hit_t *h) {
    v3 center = center0;
    if (kind == N_MOVING) { /* hitable.clj:219-222,227 */
        real t0 = (real)g[7], t1 = (real)g[8];
        center = vlerp(center0, ld3(g + 4), (r->time - t0) / (t1 - t0));
    }
    v3 oc = vsub(r->o, center);
    real a = vdot(r->d, r->d);
    real b = (real)2.0 * vdot(oc, r->d);
    real c = vdot(oc, oc) - radius * radius;
    real disc = b * b - ((real)4.0 * a) * c;
    if (disc >= (real)0) {
        real sq = R_SQRT(disc);
        for (int root = 0; root < 2; ++root) {
            real t = root == 0 ? (-b - sq) / ((real)2.0 * a) : (-b + sq) / ((real)2.0 * a);
            if (t > tmin && t < tmax) {
                h->t = t; h->p = point_at(r, t); h->n = vnormalise(vsub(h->p, center));
                h->u = 0; h->v = 0;
                if (kind == N_UVSPHERE) sphere_uv(h->n, &h->u, &h->v);
                return 1;
            }
        }
    }
    return 0;
}

static int node_hit(const rto_scene *sc, int n, const ray_t *r, real tmin, real tmax, hit_t *h, rng_t *rng) {
    const int kind = sc->node_kind[n];
    const int32_t *a = sc->node_a + (size_t)n * 3;
    const double *g = sc->node_d + (size_t)n * 12;
    switch (kind) {
    case N_SPHERE: case N_UVSPHERE: case N_MOVING:
        if (!sphere_like_hit(ld3(g), (real)g[3], kind, g, r, tmin, tmax, h)) return 0;
        h->prim = sc->node_prim[n]; h->mat = a[2];
        return 1;
    case N_RECT_XY: case N_RECT_XZ: case N_RECT_YZ: { /* hitable.clj:269-363: inclusive t range, inclusive extents */
        const real a0 = (real)g[0], b0 = (real)g[1], a1 = (real)g[2], b1 = (real)g[3], k = (real)g[4];
        const real o[3] = {r->o.x, r->o.y, r->o.z}, d[3] = {r->d.x, r->d.y, r->d.z};
        const int ax = kind == N_RECT_XY ? 2 : (kind == N_RECT_XZ ? 1 : 0);          /* the plane's axis */
        const int ua = kind == N_RECT_YZ ? 1 : 0, va = kind == N_RECT_XY ? 1 : 2;     /* the two in-plane axes */
        const real t = (k - o[ax]) / d[ax];
        if (!(t >= tmin && t <= tmax)) return 0;
        const real x = o[ua] + t * d[ua], y = o[va] + t * d[va];
        if (!(x >= a0 && x <= a1 && y >= b0 && y <= b1)) return 0;
        h->t = t; h->p = point_at(r, t);
        h->u = (x - a0) / (a1 - a0); h->v = (y - b0) / (b1 - b0);
        h->n = V(ax == 0, ax == 1, ax == 2);
        h->prim = sc->node_prim[n]; h->mat = a[2];
        return 1;
    }
    case N_TRIANGLE: { /* hitable.clj:548-571 Moeller-Trumbore, one sided */
        const v3 v0 = ld3(g), v1 = ld3(g + 3), v2 = ld3(g + 6);
        const v3 v0v1 = vsub(v1, v0), v0v2 = vsub(v2, v0);
        const v3 pvec = vcross(r->d, v0v2);
        const real det = vdot(v0v1, pvec);
        if (!(det > (real)0.00000001)) return 0;
        const real inv_det = (real)1.0 / det;
        const v3 tvec = vsub(r->o, v0);
        const real u = vdot(tvec, pvec) * inv_det;
        if (!(u > (real)0 && u <= (real)1)) return 0;
        const v3 qvec = vcross(tvec, v0v1);
        const real v = vdot(r->d, qvec) * inv_det;
        if (!(v > (real)0 && u + v <= (real)1)) return 0;
        const real t = vdot(v0v2, qvec) * inv_det;
        if (!(t >= tmin && t <= tmax)) return 0;
        h->t = t; h->p = point_at(r, t); h->u = u; h->v = v; h->n = vcross(v0v1, v0v2);
        h->prim = sc->node_prim[n]; h->mat = a[2];
        return 1;
    }
    case N_FLIP: /* hitable.clj:375-381 */
        if (!node_hit(sc, a[0], r, tmin, tmax, h, rng)) return 0;
        h->n = vneg(h->n);
        return 1;
    case N_TRANSLATE: { /* hitable.clj:391-396 */
        ray_t tr = *r;
        tr.o = vsub(r->o, ld3(g));
        if (!node_hit(sc, a[0], &tr, tmin, tmax, h, rng)) return 0;
        h->p = vadd(h->p, ld3(g));
        return 1;
    }
    case N_ROTATE_Y: { /* hitable.clj:410-450 */
        const real sn = (real)g[0], cs = (real)g[1];
        ray_t rr;
        rr.o = V(cs * r->o.x - sn * r->o.z, r->o.y, sn * r->o.x + cs * r->o.z);
        rr.d = V(cs * r->d.x - sn * r->d.z, r->d.y, sn * r->d.x + cs * r->d.z);
        rr.time = r->time;
        if (!node_hit(sc, a[0], &rr, tmin, tmax, h, rng)) return 0;
        const v3 p = h->p, nn = h->n;
        h->p = V(cs * p.x + sn * p.z, p.y, (-(sn * p.x)) + cs * p.z);
        h->n = V(cs * nn.x + sn * nn.z, nn.y, (-(sn * nn.x)) + cs * nn.z);
        return 1;
    }
    case N_BOX: /* hitable.clj:491-494: hit? of the six sides' Hitlist */
        return node_hit(sc, a[0], r, tmin, tmax, h, rng);
    case N_MEDIUM: { /* hitable.clj:516-541 ConstantMedium: the boundary is hit twice on the whole line, the segment inside is
                        clipped to [t-min, t-max], and ONE random number decides where (whether) the ray scatters inside */
        const real FMAX = (real)3.4028234663852886e38;
        hit_t h1, h2;
        if (!node_hit(sc, a[0], r, -FMAX, FMAX, &h1, rng)) return 0;
        if (!node_hit(sc, a[0], r, h1.t + (real)0.0001, FMAX, &h2, rng)) return 0;
        real t1 = h1.t, t2 = h2.t;
        if (t1 < tmin) t1 = tmin;
        if (t2 > tmax) t2 = tmax;
        if (!(t1 < t2)) return 0;
        if (t1 < (real)0) t1 = 0;
        const real mag = vmag(r->d);
        const real dist_in = (t2 - t1) * mag;
        const real hit_distance = -((real)log((double)rng_next(rng)) / (real)g[0]);
        if (!(hit_distance < dist_in)) return 0;
        h->t = t1 + hit_distance / mag;
        h->p = point_at(r, h->t);
        h->u = 0; h->v = 0; h->n = V(1, 0, 0);
        h->prim = sc->node_prim[n]; h->mat = a[2];
        return 1;
    }
    case N_HITLIST: { /* hitable.clj:15-26 */
        int found = 0; real closest = tmax; hit_t tmp;
        for (int k = 0; k < a[1]; ++k)
            if (node_hit(sc, sc->node_children[a[0] + k], r, tmin, closest, &tmp, rng)) { found = 1; closest = tmp.t; *h = tmp; }
        return found;
    }
    case N_BVH: { /* hitable.clj:97-105: slab test, both children with the un-narrowed interval, ties -> right */
        const double o[3] = {r->o.x, r->o.y, r->o.z}, d[3] = {r->d.x, r->d.y, r->d.z};
        if (!rto_aabb_hit(g, g + 3, o, d, tmin, tmax)) return 0;
        hit_t hl, hr;
        const int fl = node_hit(sc, a[0], r, tmin, tmax, &hl, rng), fr = node_hit(sc, a[1], r, tmin, tmax, &hr, rng);
        if (fl && fr) { *h = (hl.t < hr.t) ? hl : hr; return 1; }
        if (fl) { *h = hl; return 1; }
        if (fr) { *h = hr; return 1; }
        return 0;
    }
    default:
        return 0;
    }
}

/* ---- hitable.clj:15-26 Hitlist: linear scan, each item tested with t-max = best so far ---- */
static int hitlist_hit(const rto_scene *sc, const ray_t *r, real tmin, real tmax, hit_t *h, rng_t *rng) {
    if (sc->n_nodes > 0) return node_hit(sc, sc->root, r, tmin, tmax, h, rng);
    int found = 0;
    real closest = tmax;
    hit_t tmp;
    for (int32_t i = 0; i < sc->n_prims; ++i) {
        if (sphere_hit(sc, i, r, tmin, closest, &tmp)) {
            found = 1; closest = tmp.t; *h = tmp;
        }
    }
    return found;
}

/* ---- shader.clj:6-20 ---- */
static v3 reflect(v3 v, v3 n) { return vsub(v, vscale(n, (real)2.0 * vdot(v, n))); }
static int refract(v3 v, v3 n, real ni_over_nt, v3 *out) {
    v3 uv = vnormalise(v);
    real dt = vdot(uv, n);
    real disc = (real)1.0 - (ni_over_nt * ni_over_nt) * ((real)1.0 - dt * dt);
    if (disc > (real)0) {
        *out = vsub(vscale(vsub(uv, vscale(n, dt)), ni_over_nt), vscale(n, R_SQRT(disc)));
        return 1;
    }
    return 0;
}
/* shader.clj:69-74 */
static real schlick(real cosine, real ri) {
    real r0 = ((real)1.0 - ri) / ((real)1.0 + ri);
    r0 = r0 * r0;
    return r0 + ((real)1.0 - r0) * R_POW((real)1.0 - cosine, (real)5.0);
}
RTO_API double rto_schlick(double cosine, double ri) { return (double)schlick((real)cosine, (real)ri); }

/* ---- Shader protocol: scatter (shader.clj:29-36, 46-57, 76-102, 114-117) ---- */
static int scatter(const rto_scene *sc, const ray_t *rin, const hit_t *h, rng_t *g, ray_t *out, v3 *att) {
    int32_t m = h->mat;
    switch (sc->mat_kind[m]) {
    case MAT_LAMBERTIAN: {
        v3 target = vadd(vadd(h->p, h->n), rand_in_unit_sphere(g));
        out->o = h->p; out->d = vsub(target, h->p); out->time = rin->time;
        *att = tex_sample(sc, sc->mat_tex[m], h->u, h->v, h->p);
        return 1;
    }
    case MAT_METAL: {
        real fuzz = (real)sc->mat_param[m];
        v3 reflected = reflect(vnormalise(rin->d), h->n);
        v3 dir = vadd(reflected, vscale(rand_in_unit_sphere(g), fuzz));
        if (vdot(dir, h->n) > (real)0) {
            out->o = h->p; out->d = dir; out->time = rin->time;
            *att = tex_sample(sc, sc->mat_tex[m], h->u, h->v, h->p);
            return 1;
        }
        return 0;
    }
    case MAT_DIELECTRIC: {
        real ri = (real)sc->mat_param[m];
        v3 d = rin->d;
        real dn = vdot(d, h->n);
        v3 outward; real ni_over_nt, cosine;
        if (dn > (real)0) {
            outward = vneg(h->n); ni_over_nt = ri; cosine = ri * (dn / vmag(d));
        } else {
            outward = h->n; ni_over_nt = (real)1.0 / ri; cosine = -(dn / vmag(d));
        }
        v3 refr;
        out->o = h->p; out->time = rin->time;
        *att = V(1, 1, 1);
        if (refract(d, outward, ni_over_nt, &refr)) {
            if (rng_next(g) < schlick(cosine, ri)) out->d = reflect(d, h->n);
            else out->d = refr;
        } else {
            out->d = reflect(d, h->n);
        }
        return 1;
    }
    case MAT_ISOTROPIC: { /* shader.clj:129-138: (ray p (rand-in-unit-sphere) t) -- the new ray's TIME is the hit's t */
        out->o = h->p; out->d = rand_in_unit_sphere(g); out->time = h->t;
        *att = tex_sample(sc, sc->mat_tex[m], h->u, h->v, h->p);
        return 1;
    }
    default: /* DiffuseLight: scatter -> nil */
        return 0;
    }
}
/* Shader protocol: emitted (shader.clj:35-36, 58-59, 103-104, 118-119) */
static v3 emitted(const rto_scene *sc, const hit_t *h) {
    if (sc->mat_kind[h->mat] == MAT_DIFFUSE_LIGHT) return tex_sample(sc, sc->mat_tex[h->mat], h->u, h->v, h->p);
    return V(0, 0, 0);
}

/* ---- camera.clj:8-16, 35-48 get-ray ---- */
static ray_t get_ray(const rto_scene *sc, real s, real t, rng_t *g) {
    const double *c = sc->cam;
    v3 origin = ld3(c), lleft = ld3(c + 3), horiz = ld3(c + 6), vert = ld3(c + 9);
    ray_t r;
    if (sc->cam_kind == CAM_PINHOLE) {
        r.o = origin;
        r.d = vadd(vadd(vadd(lleft, vscale(horiz, s)), vscale(vert, t)), vneg(origin));
        r.time = 0;
        return r;
    }
    v3 u = ld3(c + 12), v = ld3(c + 15);
    real aperture = (real)c[21], t0 = (real)c[22], t1 = (real)c[23];
    real lens_radius = aperture / (real)2.0;
    v3 rd = vscale(rand_in_unit_disk(g), lens_radius);
    v3 offset = vadd(vscale(u, rd.x), vscale(v, rd.y));
    r.o = vadd(origin, offset);
    r.d = vadd(vadd(vadd(vadd(lleft, vscale(horiz, s)), vscale(vert, t)), vneg(origin)), vneg(offset));
    r.time = t0 + (t1 - t0) * rng_next(g);
    return r;
}

/* ---- core.clj:17-41 color ---- */
typedef struct { /* optional per-segment log for bit-level parity tests */
    double *rec; /* max_seg * SEG_REC doubles */
    int max_seg;
    int n;
} seglog_t;
#define SEG_REC 12 /* prim, t, p(3), n(3), next-dir(3), scattered? */

static v3 color(const rto_scene *sc, ray_t r, int depth, rng_t *g, uint64_t *nrays, seglog_t *lg) {
    const real T_MIN = (real)0.001;
    const real T_MAX = (real)3.4028234663852886e38; /* Float/MAX_VALUE */
    v3 atten = V(1, 1, 1), accum = V(0, 0, 0);
    for (;;) {
        ++*nrays; /* metrics count-rays, core.clj:24 */
        hit_t h;
        if (!hitlist_hit(sc, &r, T_MIN, T_MAX, &h, g)) return accum; /* core.clj:40-41 */
        ray_t sr; v3 att;
        int scat = depth > 0 ? scatter(sc, &r, &h, g, &sr, &att) : 0;
        v3 e = emitted(sc, &h);
        if (lg && lg->n < lg->max_seg) {
            double *q = lg->rec + (size_t)lg->n * SEG_REC;
            q[0] = h.prim; q[1] = h.t; q[2] = h.p.x; q[3] = h.p.y; q[4] = h.p.z;
            q[5] = h.n.x; q[6] = h.n.y; q[7] = h.n.z;
            q[8] = scat ? sr.d.x : 0; q[9] = scat ? sr.d.y : 0; q[10] = scat ? sr.d.z : 0;
            q[11] = scat;
            lg->n++;
        }
        accum = vadd(accum, vmul(atten, e)); /* uses the OLD atten, core.clj:32-34 / 37-39 */
        if (!scat) return accum;
        atten = vmul(atten, att);
        r = sr;
        --depth;
    }
}

/* ---- core.clj:43-57 pixel (i, j in reference coordinates: j = 0 is the bottom row) ---- */
static void pixel(const rto_scene *sc, int i, int j, int nx, int ny, int ns, int depth, uint64_t seed,
                  real mean[3], uint8_t rgb8[3], uint64_t *nrays) {
    v3 sum = V(0, 0, 0);
    for (int s = 0; s < ns; ++s) {
        rng_t g = {rto_sample_key(seed, (uint64_t)j * (uint64_t)nx + (uint64_t)i, (uint64_t)s), 0};
        real u = ((real)(float)i + rng_next(&g)) / (real)nx;
        real v = ((real)(float)j + rng_next(&g)) / (real)ny;
        ray_t r = get_ray(sc, u, v, &g);
        v3 c = color(sc, r, depth, &g, nrays, NULL);
        sum = s == 0 ? c : vadd(sum, c);
    }
    v3 m = vscale(sum, (real)1.0 / (real)ns);
    real comp[3] = {m.x, m.y, m.z};
    for (int k = 0; k < 3; ++k) {
        mean[k] = comp[k];
        real q = R_SQRT(comp[k]) * (real)255.99;
        /* (int (min 255.99 q)): clojure min propagates NaN, (int NaN) = 0 */
        real mq = (q != q) ? q : (q < (real)255.99 ? q : (real)255.99);
        rgb8[k] = (mq != mq) ? 0 : (uint8_t)(int)mq;
    }
}

/* ---- the render loop (core.clj:100-108): 32-pixel row-major chunks (tiled-coords, core.clj:59-71) handed to a
 * pool of threads in no particular order (cp/upmap), y-flip on store ---- */
typedef struct {
    const rto_scene *sc; int nx, ny, ns, depth; uint64_t seed;
    int x0, y0, x1, y1; double *lin; uint8_t *rgb8;
    long *next_chunk; uint64_t nrays;
} job_t;

static void *render_chunks(void *arg) {
    job_t *jb = (job_t *)arg;
    const int w = jb->x1 - jb->x0;
    const long npix = (long)w * (jb->y1 - jb->y0);
    for (;;) {
        long c = __atomic_fetch_add(jb->next_chunk, 1, __ATOMIC_RELAXED);
        long first = c * 32;
        if (first >= npix) break;
        long lastp = first + 32 < npix ? first + 32 : npix;
        for (long k = first; k < lastp; ++k) {
            int x = jb->x0 + (int)(k % w), y = jb->y0 + (int)(k / w);
            int j = jb->ny - 1 - y; /* core.clj:105: set-pixel image i (- (dec ny) j) */
            real mean[3]; uint8_t q[3];
            pixel(jb->sc, x, j, jb->nx, jb->ny, jb->ns, jb->depth, jb->seed, mean, q, &jb->nrays);
            size_t o = (size_t)k * 3;
            for (int ch = 0; ch < 3; ++ch) {
                if (jb->lin) jb->lin[o + ch] = (double)mean[ch];
                if (jb->rgb8) jb->rgb8[o + ch] = q[ch];
            }
        }
    }
    return NULL;
}

/* Region [x0,x1) x [y0,y1) in OUTPUT coordinates (row 0 = top).  counters = {total-rays, total-pixels}. */
RTO_API int rto_render(const rto_scene *sc, int nx, int ny, int ns, int depth, uint64_t seed,
                       int x0, int y0, int x1, int y1, double *out_linear, uint8_t *out_rgb8,
                       uint64_t *counters, int nthreads) {
    if (!sc || nx <= 0 || ny <= 0 || ns <= 0 || x0 < 0 || y0 < 0 || x1 > nx || y1 > ny || x1 < x0 || y1 < y0) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t jobs[256]; pthread_t th[256];
    long next_chunk = 0;
    for (int t = 0; t < nthreads; ++t) {
        job_t jb = {sc, nx, ny, ns, depth, seed, x0, y0, x1, y1, out_linear, out_rgb8, &next_chunk, 0};
        jobs[t] = jb;
    }
    if (nthreads == 1) render_chunks(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, render_chunks, &jobs[t]);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    if (counters) {
        uint64_t n = 0;
        for (int t = 0; t < nthreads; ++t) n += jobs[t].nrays;
        counters[0] = n;
        counters[1] = (uint64_t)(x1 - x0) * (uint64_t)(y1 - y0);
    }
    return 0;
}

/* ---- probes used by the parity tests ---- */

/* Hitable.hit? of the whole world (Hitlist) for n rays {o(3), d(3), time}; out rec = {hit, prim, t, p(3), n(3), u, v} */
RTO_API int rto_probe_hit(const rto_scene *sc, int n, const double *rays, double tmin, double tmax, double *out) {
    for (int k = 0; k < n; ++k) {
        const double *q = rays + (size_t)k * 7;
        ray_t r = {ld3(q), ld3(q + 3), (real)q[6]};
        hit_t h; double *o = out + (size_t)k * 11;
        memset(o, 0, 11 * sizeof(double));
        rng_t g = {0, 0}; /* a ConstantMedium draws inside hit?: the probe uses stream key 0 */
        if (hitlist_hit(sc, &r, (real)tmin, (real)tmax, &h, &g)) {
            o[0] = 1; o[1] = h.prim; o[2] = h.t; o[3] = h.p.x; o[4] = h.p.y; o[5] = h.p.z;
            o[6] = h.n.x; o[7] = h.n.y; o[8] = h.n.z; o[9] = h.u; o[10] = h.v;
        }
    }
    return 0;
}

/* color of n explicit rays, each with its own stream key (draw counter starts at ctr0);
 * out_rgb n*3, out_nseg n, optional log n*max_seg*SEG_REC (+ out_nlog n). */
RTO_API int rto_probe_paths(const rto_scene *sc, int n, const double *rays, const uint64_t *keys, uint64_t ctr0,
                            int depth, double *out_rgb, uint64_t *out_nseg, double *log, int max_seg, int32_t *out_nlog) {
    for (int k = 0; k < n; ++k) {
        const double *q = rays + (size_t)k * 7;
        ray_t r = {ld3(q), ld3(q + 3), (real)q[6]};
        rng_t g = {keys[k], ctr0};
        uint64_t ns = 0;
        seglog_t lg = {log ? log + (size_t)k * max_seg * SEG_REC : NULL, max_seg, 0};
        v3 c = color(sc, r, depth, &g, &ns, log ? &lg : NULL);
        out_rgb[3 * k] = c.x; out_rgb[3 * k + 1] = c.y; out_rgb[3 * k + 2] = c.z;
        if (out_nseg) out_nseg[k] = ns;
        if (out_nlog) out_nlog[k] = lg.n;
    }
    return 0;
}

/* Camera.get-ray for n (u,v) pairs with per-ray keys; out {o(3), d(3), time, draws consumed} */
RTO_API int rto_probe_camera(const rto_scene *sc, int n, const double *uv, const uint64_t *keys, double *out) {
    for (int k = 0; k < n; ++k) {
        rng_t g = {keys[k], 0};
        ray_t r = get_ray(sc, (real)uv[2 * k], (real)uv[2 * k + 1], &g);
        double *o = out + (size_t)k * 8;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
        o[6] = r.time; o[7] = (double)g.ctr;
    }
    return 0;
}

/* Texture.sample for n (u, v, p) tuples */
RTO_API int rto_probe_texture(const rto_scene *sc, int tex, int n, const double *uvp, double *out) {
    for (int k = 0; k < n; ++k) {
        const double *q = uvp + (size_t)k * 5;
        v3 c = tex_sample(sc, tex, (real)q[0], (real)q[1], ld3(q + 2));
        out[3 * k] = c.x; out[3 * k + 1] = c.y; out[3 * k + 2] = c.z;
    }
    return 0;
}

/* Shader.scatter on an explicit hit record {p(3), n(3), u, v} with material m for ray {o,d,time};
 * out {scattered?, dir(3), att(3), time, draws consumed} */
RTO_API int rto_probe_scatter(const rto_scene *sc, int m, int n, const double *rays, const double *hits,
                              const uint64_t *keys, double *out) {
    for (int k = 0; k < n; ++k) {
        const double *q = rays + (size_t)k * 7, *hq = hits + (size_t)k * 8;
        ray_t r = {ld3(q), ld3(q + 3), (real)q[6]}, sr;
        hit_t h; h.t = 0; h.p = ld3(hq); h.n = ld3(hq + 3); h.u = (real)hq[6]; h.v = (real)hq[7]; h.prim = -1; h.mat = m;
        rng_t g = {keys[k], 0};
        v3 att = V(0, 0, 0);
        memset(&sr, 0, sizeof sr);
        int s = scatter(sc, &r, &h, &g, &sr, &att);
        double *o = out + (size_t)k * 9;
        o[0] = s; o[1] = s ? sr.d.x : 0; o[2] = s ? sr.d.y : 0; o[3] = s ? sr.d.z : 0;
        o[4] = s ? att.x : 0; o[5] = s ? att.y : 0; o[6] = s ? att.z : 0; o[7] = s ? sr.time : 0; o[8] = (double)g.ctr;
    }
    return 0;
}

/* small pure functions exposed for the known-answer tests */
RTO_API void rto_point_at_parameter(const double *o, const double *d, double t, double *out) {
    ray_t r = {ld3(o), ld3(d), 0};
    v3 p = point_at(&r, (real)t);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
RTO_API void rto_reflect(const double *v, const double *n, double *out) {
    v3 r = reflect(ld3(v), ld3(n));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
RTO_API int rto_refract(const double *v, const double *n, double ni_over_nt, double *out) {
    v3 r;
    if (!refract(ld3(v), ld3(n), (real)ni_over_nt, &r)) return 0;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return 1;
}
RTO_API void rto_center_at_time(const double *c0, double t0, const double *c1, double t1, double t, double *out) {
    v3 c = vlerp(ld3(c0), ld3(c1), (real)((t - t0) / (t1 - t0)));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
RTO_API void rto_sphere_uv(const double *n, double *uv) {
    real u, v;
    sphere_uv(ld3(n), &u, &v);
    uv[0] = u; uv[1] = v;
}
RTO_API void rto_quantise(const double *mean, uint8_t *out) { /* core.clj:54-56 */
    for (int k = 0; k < 3; ++k) {
        real q = R_SQRT((real)mean[k]) * (real)255.99;
        real mq = (q != q) ? q : (q < (real)255.99 ? q : (real)255.99);
        out[k] = (mq != mq) ? 0 : (uint8_t)(int)mq;
    }
}

/* ---- camera constructors: camera.clj:18-33 (pinhole), 50-66 (thin lens) -> cam[24] ---- */
RTO_API void rto_make_camera(int kind, const double *lookfrom, const double *lookat, const double *vup, double vfov,
                             double aspect, double aperture, double focus_dist, double t0, double t1, double *cam) {
    real theta = (real)vfov * ((real)3.141592653589793 / (real)180.0);
    real hh = (real)tan((double)(theta / (real)2.0));
    real hw = (real)aspect * hh;
    v3 from = ld3(lookfrom);
    v3 w = vnormalise(vsub(from, ld3(lookat)));
    v3 u = vnormalise(vcross(ld3(vup), w));
    v3 v = vcross(w, u);
    v3 lleft, horiz, vert;
    if (kind == CAM_PINHOLE) {
        lleft = vsub(from, vadd(vadd(vscale(u, hw), vscale(v, hh)), w));
        horiz = vscale(u, (real)2.0 * hw);
        vert = vscale(v, (real)2.0 * hh);
        aperture = 0; t0 = 0; t1 = 0;
    } else {
        real fd = (real)focus_dist;
        lleft = vsub(from, vadd(vadd(vscale(u, fd * hw), vscale(v, fd * hh)), vscale(w, fd)));
        horiz = vscale(u, ((real)2.0 * fd) * hw);
        vert = vscale(v, ((real)2.0 * fd) * hh);
    }
    v3 all[7] = {from, lleft, horiz, vert, u, v, w};
    for (int k = 0; k < 7; ++k) { cam[3 * k] = all[k].x; cam[3 * k + 1] = all[k].y; cam[3 * k + 2] = all[k].z; }
    cam[21] = aperture; cam[22] = t0; cam[23] = t1;
}

/* ---- AABB + BVH (hitable.clj:36-48, 87-123): only to show closest-hit equivalence with the flat scan ---- */
RTO_API int rto_aabb_hit(const double *vmin, const double *vmax, const double *o, const double *d, double tmin, double tmax) {
    /* m = (vmin-org)/dir, n = (vmax-org)/dir; t0 = emap min, t1 = emap max (clojure.core/min|max propagate NaN);
     * mat/maximum, mat/minimum (vectorz elementMax/elementMin) scan with `d > max` / `d < min`, so a NaN element
     * (0/0: axis-parallel ray lying in a face plane) never wins.  The reference's own test data requires this:
     * hitable_test.clj:126-131 "grazing x/y/z" hit although one slab is NaN. */
    double lo = -INFINITY, hi = INFINITY;
    for (int k = 0; k < 3; ++k) {
        double m = (vmin[k] - o[k]) / d[k], n = (vmax[k] - o[k]) / d[k];
        double t0 = (m != m || n != n) ? NAN : (m < n ? m : n);
        double t1 = (m != m || n != n) ? NAN : (m > n ? m : n);
        if (t0 > lo) lo = t0;
        if (t1 < hi) hi = t1;
    }
    double a = lo > tmin ? lo : tmin;
    double b = hi < tmax ? hi : tmax;
    return b > a;
}
RTO_API void rto_prim_bbox(const rto_scene *sc, int i, double t0, double t1, double *vmin, double *vmax) {
    const double *g = sc->prim_geom + (size_t)i * PRIM_STRIDE;
    double r = g[3];
    if (sc->prim_kind[i] == PRIM_MOVING) { /* hitable.clj:253-259 */
        double f0 = (t0 - g[7]) / (g[8] - g[7]), f1 = (t1 - g[7]) / (g[8] - g[7]);
        for (int k = 0; k < 3; ++k) {
            double a = g[k] * (1.0 - f0) + g[4 + k] * f0, b = g[k] * (1.0 - f1) + g[4 + k] * f1;
            double lo0 = a - r, lo1 = b - r, hi0 = a + r, hi1 = b + r;
            vmin[k] = lo0 < lo1 ? lo0 : lo1;
            vmax[k] = hi0 > hi1 ? hi0 : hi1;
        }
    } else {
        for (int k = 0; k < 3; ++k) { vmin[k] = g[k] - r; vmax[k] = g[k] + r; }
    }
}

RTO_API void rto_surrounding_bbox(const double *min0, const double *max0, const double *min1, const double *max1,
                                  double *vmin, double *vmax) { /* hitable.clj:87-92 */
    for (int k = 0; k < 3; ++k) { vmin[k] = min0[k] < min1[k] ? min0[k] : min1[k]; vmax[k] = max0[k] > max1[k] ? max0[k] : max1[k]; }
}

typedef struct bvh_s { int left, right; /* >=0 node index, <0 = ~prim */ double vmin[3], vmax[3]; } bvh_t;
typedef struct { const rto_scene *sc; bvh_t *nodes; int n_nodes; double t0, t1; int axis; uint64_t rs; } bvhb_t;
static bvhb_t *g_sort_ctx;
static int cmp_axis(const void *a, const void *b) {
    double mn[3], mx[3], nn[3], nx_[3];
    int ia = *(const int *)a, ib = *(const int *)b;
    bvhb_t *c = g_sort_ctx;
    /* entries are encoded: >=0 node, <0 ~prim */
    if (ia < 0) rto_prim_bbox(c->sc, ~ia, c->t0, c->t1, mn, mx); else memcpy(mn, c->nodes[ia].vmin, sizeof mn);
    if (ib < 0) rto_prim_bbox(c->sc, ~ib, c->t0, c->t1, nn, nx_); else memcpy(nn, c->nodes[ib].vmin, sizeof nn);
    (void)mx; (void)nx_;
    return (mn[c->axis] > nn[c->axis]) - (mn[c->axis] < nn[c->axis]);
}
static void ent_bbox(bvhb_t *c, int e, double *mn, double *mx) {
    if (e < 0) rto_prim_bbox(c->sc, ~e, c->t0, c->t1, mn, mx);
    else { memcpy(mn, c->nodes[e].vmin, 3 * sizeof(double)); memcpy(mx, c->nodes[e].vmax, 3 * sizeof(double)); }
}
static int bvh_build(bvhb_t *c, int *list, int n) { /* hitable.clj:108-123 make-bvh (axis from our own stream) */
    c->rs += GOLD;
    c->axis = (int)(splitmix64_fin(c->rs) % 3);
    g_sort_ctx = c;
    qsort(list, (size_t)n, sizeof(int), cmp_axis);
    int L, R;
    if (n == 1) { L = list[0]; R = list[0]; }
    else if (n == 2) { L = list[0]; R = list[1]; }
    else {
        int h = (n + 1) / 2; /* (split-at (/ n 2) ...): for odd n, (/ n 2) is a Ratio and take/drop count down past it -> ceil */
        int *l2 = (int *)malloc(sizeof(int) * (size_t)n);
        memcpy(l2, list, sizeof(int) * (size_t)n);
        L = bvh_build(c, l2, h);
        R = bvh_build(c, l2 + h, n - h);
        free(l2);
    }
    int id = c->n_nodes++;
    bvh_t *nd = &c->nodes[id];
    nd->left = L; nd->right = R;
    double a0[3], a1[3], b0[3], b1[3];
    ent_bbox(c, L, a0, a1); ent_bbox(c, R, b0, b1);
    for (int k = 0; k < 3; ++k) { nd->vmin[k] = a0[k] < b0[k] ? a0[k] : b0[k]; nd->vmax[k] = a1[k] > b1[k] ? a1[k] : b1[k]; }
    return id;
}
static int bvh_hit(const rto_scene *sc, const bvh_t *nodes, int e, const ray_t *r, real tmin, real tmax, hit_t *h) {
    if (e < 0) return sphere_hit(sc, ~e, r, tmin, tmax, h);
    const bvh_t *nd = &nodes[e];
    double o[3] = {r->o.x, r->o.y, r->o.z}, d[3] = {r->d.x, r->d.y, r->d.z};
    if (!rto_aabb_hit(nd->vmin, nd->vmax, o, d, tmin, tmax)) return 0;
    hit_t hl, hr;
    int a = bvh_hit(sc, nodes, nd->left, r, tmin, tmax, &hl);
    int b = bvh_hit(sc, nodes, nd->right, r, tmin, tmax, &hr);
    if (a && b) { *h = (hl.t < hr.t) ? hl : hr; return 1; } /* hitable.clj:103: ties -> right */
    if (a) { *h = hl; return 1; }
    if (b) { *h = hr; return 1; }
    return 0;
}
/* Build a reference-style BVH over the prims and intersect n rays with it; same out record as rto_probe_hit. */
RTO_API int rto_probe_hit_bvh(const rto_scene *sc, uint64_t build_seed, int n, const double *rays, double tmin, double tmax, double *out) {
    int np = sc->n_prims;
    if (np <= 0) return -1;
    bvhb_t c; c.sc = sc; c.nodes = (bvh_t *)malloc(sizeof(bvh_t) * (size_t)(2 * np + 2)); c.n_nodes = 0; c.t0 = 0.0; c.t1 = 1.0; c.rs = build_seed; c.axis = 0;
    int *list = (int *)malloc(sizeof(int) * (size_t)np);
    for (int i = 0; i < np; ++i) list[i] = ~i;
    int root = bvh_build(&c, list, np);
    for (int k = 0; k < n; ++k) {
        const double *q = rays + (size_t)k * 7;
        ray_t r = {ld3(q), ld3(q + 3), (real)q[6]};
        hit_t h; double *o = out + (size_t)k * 11;
        memset(o, 0, 11 * sizeof(double));
        if (bvh_hit(sc, c.nodes, root, &r, (real)tmin, (real)tmax, &h)) {
            o[0] = 1; o[1] = h.prim; o[2] = h.t; o[3] = h.p.x; o[4] = h.p.y; o[5] = h.p.z;
            o[6] = h.n.x; o[7] = h.n.y; o[8] = h.n.z; o[9] = h.u; o[10] = h.v;
        }
    }
    free(list); free(c.nodes);
    return 0;
}

RTO_API int rto_real_bytes(void) { return (int)sizeof(real); }
