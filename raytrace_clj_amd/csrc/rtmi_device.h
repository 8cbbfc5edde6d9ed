// rtmi_device.h -- device functions of the MI355X sampling path (gfx950 only, wave64).
//
// Every function states the reference lines it replaces (paths relative to the reference repo).
// Arithmetic contract: this file is compiled with -ffp-contract=off; every + - * / sqrt below is one
// correctly-rounded IEEE operation in the order the reference's core.matrix/vectorz calls perform
// them (variadic add/mul = left fold, dot = (x0*y0 + x1*y1) + x2*y2, normalise = v * (1/|v|)), so
// ray geometry is bit-reproducible against a CPU evaluation of the same formulas.  sin/asin/atan2/
// pow only feed terminal colours and sign/probability tests, never the ray geometry.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtmi.h"

namespace rtmi {

typedef unsigned long long u64;

// ---- flat scene as the kernels see it (all pointers are HBM) -----------------------------------
struct DevScene {
    int n_static;            // Sphere + UVSphere records, original relative order
    int n_moving;            // MovingSphere records
    const double *stat_geom; // [n_static][4]  cx cy cz radius
    const double *stat4_d;   // [round_up(n_static,8)+8][4]  cx cy cz r*r (double; hitable.clj:188), padded with copies of the last
    const float *stat4_f;    // the same, evaluated in float for RTMI_F32
    // scan variant SCAN_SGPR_CULL walks ALL primitives in the caller's Hitlist order (index = original index):
    int n_all;               // number of primitives
    const float *cull20;     // [round_up(n_all,8)/4 + 2][20] per group of 4: cx[4] cy[4] cz[4] r2[4] w[4] (see CullRay);
                             // a MovingSphere is represented by a sphere bounding its sweep over [cull_t_lo, cull_t_hi]
    const double *exact12;   // [round_up(n_all,8)+8][12] c0.xyz r*r c1.xyz t0 t1 moving? 0 0 (padded with copies of the last)
    double cull_t_lo, cull_t_hi; // ray times the swept bounds are valid for (the camera's [t0, t1])
    // RTMI_ACCEL_BVH (the reference's bvh-node descent, hitable.clj:97-123, rebuilt for the device):
    const float *bvh_nodes;  // [n_nodes][16]: l.lo.xy l.hi.xy r.lo.xy r.hi.xy l.lo.z l.hi.z r.lo.z r.hi.z, left, right (int bits), 0, 0
    int bvh_root;            // child code: >= 0 byte offset of the node record, < 0 = ~(original primitive index | moving << 30), 0x7fffffff = empty
    int n_big;               // primitives too large to bound usefully (sky dome, ground): always tested exactly
    int big_idx[16];
    int bvh_node16;          // node records: 1 = Node16 (32 bytes, half planes), 0 = 16 floats
    float bvh_obound;        // rays starting outside [-obound, obound]^3 move every box plane out by 2^-20 |o| themselves
    float bvh_cbound;        // largest |coordinate| of the boxes in the tree (the big primitives are outside it)
    int n_moving_all;        // all MovingSphere world primitives (tested exhaustively for rays outside the shutter interval)
    const int *moving_all;
    // Entry grid (sphere-only scenes whose primitives form a layer over the x-z plane, e.g. the cover scene): the layer is cut into
    // grid_n x grid_n cells, every cell owns a BVH over the primitives whose boxes overlap it (a primitive on a border is in several), the
    // few primitives much taller than the rest form one more small tree.  A ray whose clipped segment stays within grid_kmax cells starts
    // at those cells' roots instead of descending from the root of the whole tree (half of a traversal's node visits only locate the ray).
    int grid_n;              // 0: no grid
    int grid_kmax;           // a start may touch at most this many cells (experiments; 4 = up to 2 x 2)
    int grid_walk;           // 1: a segment touching more cells is walked in pieces (RTMI_GRID_CHUNK); 0: it takes the whole tree, from its root (RTMI_GRID_WALK=0, or grid_kmax < 4)
    int grid_tall;           // root code of the tall primitives' tree (RTMI_BVH_EMPTY: none)
    float grid_lo_x, grid_lo_z, grid_inv_x, grid_inv_z; // cell index = floor((p - lo) * inv)
    float grid_box[6];       // lo.xyz hi.xyz of the layer primitives' boxes (rounded outward like every node box): a ray's t range inside it
    float grid_tall_box[6];  // the same for the tall primitives: their tree is entered only by rays that meet this box
    float grid_eps;          // the cell rectangle of a segment is grown by this much (float position error, far below a cell)
    unsigned grid_box_h[3], grid_tall_box_h[3]; // the two boxes again, per axis a {lo, hi} pair of IEEE halves rounded outward (like a Node16's planes): what the device tests
    const int *grid_cells;   // [4][grid_n][grid_n] root codes: family wi + 2 wj = the tree over the (1 + wi) x (1 + wj) cells whose low corner is (row = z cell, column = x cell)
    // section 8(f3): rectangles / triangles and FlipNormals / Translate / RotateY instances (hitable.clj:269-511, 548-581)
    int has_ext;             // any primitive kind > 2, any instance wrapper, any flip
    int small_scan;          // the flat scan of this mixed-kind world is scan_small_ext (n_all <= RTMI_SMALL_SCAN_MAX; RTMI_SMALL_SCAN=0 at scene creation: the culled scan)
    const double *leaf_rec;  // [n_all][14]: everything a tree leaf's exact test reads, in ONE 112-byte record (LeafRec: one round trip instead of info -> chain -> geometry)
    const int *ext_info;     // [n_all][4]: kind, FlipNormals parity, first xform, xform count (outermost first)
    const double *ext_xf;    // [n_xforms][4]: 0 | offset.xyz  (Translate)   or   1 | sin, cos, 0  (RotateY)
    // Neighbourhood trees of the media (RTMI_MEDIA_DESCENT): a segment that a medium hit bounds INSIDE the box of that medium's boundary -- a path scattering in
    // make-final's subsurface sphere: a third of that scene's segments, each a few units long -- can only meet the surfaces that reach into that box: its
    // traversal starts at a tree over those (scan_bvh_ext) instead of locating itself from the root of the whole tree.
    int n_mloc;              // number of such regions (media whose boundary box holds less than half of the tree's primitives)
    int mloc_root[4];        // root code of the region's tree (RTMI_BVH_EMPTY: no surface reaches into it)
    float mloc_box[4][6];    // lo.xyz hi.xyz: every surface point inside this box belongs to a primitive of the region's tree (the device shrinks it by the float error of a ray's end points)
    int media_lo[32];        // RTMI_MEDIA_NARROWED (media_seq = 2): per call of media_idx, the first primitive of the Hitlist items that narrow its t-max -- the call sees the
                             // closest hit among primitives [media_lo[k], media_idx[k]) (hitable.clj:15-26 inside a Hitlist that sits below bvh-nodes); = media_idx[k]: un-narrowed
    int media_seq;           // RTMI_MEDIA_HITLIST: the world is a Hitlist (hitable.clj:15-26), a medium's hit? sees the t-max narrowed by the items before it
    int n_media;             // hit? invocations of ConstantMedium primitives (hitable.clj:516) per ray, in the reference's call order
    int media_idx[32];       // (a medium may appear twice: rtmi_scene_set_media_calls); exact12 = density, first boundary prim, count
    // media calls 0 .. RTMI_MEDIA_FAST_MAX - 1 whose medium and boundary are ONE plain sphere without wrappers (make-final's two media): {1, density, c.xyz, r*r, -, -} in the
    // descriptor itself -- one scalar load where ext_medium_chord chases exact12[medium] -> ext_info[boundary] -> exact12[boundary]; {0, ...}: the general path
    double media_fast[8][8];
    // section 8(f4): perlin.clj:6-17 tables (seeded scene data) and ImageMap pixels (texture.clj:126-133)
    const double *perlin_vec; // [256][3]
    const int *perlin_perm;   // [3][256]
    int n_images;
    const int *image_wh;      // [n][2]
    const long long *image_off;
    const unsigned char *image_rgb;
    const int *stat_orig;    // [n_static] index in the caller's Hitlist
    const double *mov_geom;  // [n_moving][9]  c0.xyz radius c1.xyz t0 t1
    const int *mov_orig;
    const int *prim_kind;    // by original index
    const int *prim_mat;
    const int *prim_km;      // [n][2] = (prim_kind incl. RTMI_PRIM_NEEDS_UV, prim_mat): one 8-byte fetch for the winner
    const double *mat_rec;   // [n_mats][12]: MatRec -- what scatter needs about a material, in one record (no chain of dependent fetches)
    const double *mat_grad;  // [n_mats][12]: the four corner colours co cu cv cuv of a material whose texture is a UVGradient (tex_kind RTMI_TEX_GRADIENT_REC)
    const int *mat_kind;
    const int *mat_tex;
    const double *mat_param;
    int n_tex;
    const int *tex_kind;
    const double *tex_param; // [n_tex][12]
    const int *tex_child;    // [n_tex][2]
    int cam_kind;
    int cam_fixed_origin;    // every camera ray starts at cam[0..2] bit for bit (pinhole, or thin lens with aperture 0: origin + (+-0)): the refill need not move origins across lanes
    double cam[24];
};
// The descriptor lives in HBM and is read through the constant address space: every field access is a scalar load
// (s_load) at its use site instead of ~80 kernarg SGPRs held live across the sphere scan.
typedef const __attribute__((address_space(4))) DevScene &SceneRef;
typedef const __attribute__((address_space(4))) DevScene *ScenePtr;

// ---- counter-based stream: replaces clojure.core/rand (core.clj:49-50, util.clj:35-36,46-48,
//      camera.clj:39,48, shader.clj:93); identical bits on host and device ------------------------
#define RTMI_GOLD 0x9E3779B97F4A7C15ULL
// mix64: two rounds of xor-shift-32 / multiply and a final xor-shift-32 (the mixer published as "degski64").  Round 3 replaced the splitmix64
// finaliser (shifts 30 / 27 / 31, two constants) by it: on a 32-bit ALU a 64-bit shift by 32 is no instruction at all -- x ^= x >> 32 is ONE
// v_xor of the register pair's halves, where a shift by 30 is v_lshrrev_b64 + two v_xor -- 21 of a draw's 65 mixer cycles, and a sample draws
// ~20 times.  Avalanche and stream statistics measure the same as the finaliser's (scripts/rng_quality.py, and a CPU test of the suite).
__host__ __device__ inline u64 mix64(u64 z) {
    z ^= z >> 32; z *= 0xD6E8FEB86659FD93ULL;
    z ^= z >> 32; z *= 0xD6E8FEB86659FD93ULL;
    z ^= z >> 32;
    return z;
}
__host__ __device__ inline u64 sample_key(u64 seed, u64 pix, u64 s) {
    return mix64(mix64(seed ^ (RTMI_GOLD * (pix + 1))) + 0xD1B54A32D192ED03ULL * (s + 1));
}
__host__ __device__ inline u64 draw_bits(u64 key, u64 d) { return mix64(key + RTMI_GOLD * (d + 1)); }

#ifdef RTMI_STAMPS // diagnostic build only (make stamps): where a wave's time and its lane-slots go, phase by phase
// Every stamp closes the interval since the wave's previous stamp and books it on phase k twice: as wave ticks (s_memtime = shader
// cycles) and as lane-ticks = ticks x the lanes active AT the stamp.  A stamp therefore sits at the END of the code it names, inside
// the divergent region if that code runs under a lane mask (a region no lane enters executes no stamp: its few scalar cycles fall to
// the next stamp).  Accumulators live in LDS, one row per wave, updated by the wave's first active lane; flushed once per launch.
enum { PH_LOOP = 0, PH_REFILL_GEN, PH_REFILL_DEAL, PH_BVH_SETUP, PH_BIG, PH_DESCENT, PH_LEAF, PH_BVH_POST, PH_HITREC, PH_DNORM, PH_SAMPLER, PH_DIRS, PH_TEXTURE, PH_STORE, PH_UV, PH_CAL, PH_GRID, PH_MEDIA, PH_N }; // PH_CAL: back-to-back stamps = the cost of a stamp, subtracted per stamp on the host
#define PH_SLOTS 20
__shared__ unsigned long long g_ph[4][3 * PH_SLOTS + 2]; // [wave]: ticks[PH_SLOTS], lane-ticks[PH_SLOTS], stamps[PH_SLOTS], last stamp, unused
__device__ inline void ph_stamp(int k, unsigned lane_trips = 0, unsigned trips = 0) { // trips > 0: book lane_trips / trips lanes per tick instead of the stamp's own mask (loops that count their lanes per trip)
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    const unsigned long long ex = __ballot(1);
    const int w = threadIdx.x >> 6;
    if ((int)(threadIdx.x & 63) == __ffsll((long long)ex) - 1) {
        const unsigned long long dt = t - g_ph[w][3 * PH_SLOTS];
        if (k >= 0) {
            g_ph[w][2 * PH_SLOTS + k] += 1;
            g_ph[w][k] += dt;
            g_ph[w][PH_SLOTS + k] += trips ? dt * (unsigned long long)lane_trips / (unsigned long long)trips : dt * (unsigned long long)__popcll(ex);
        }
        g_ph[w][3 * PH_SLOTS] = t;
    }
    __builtin_amdgcn_sched_barrier(0);
}
#define RTMI_PH(k) ph_stamp(k);
#define RTMI_PH_LANES(k, lt, tr) ph_stamp(k, lt, tr);
#elif defined(RTMI_MARKERS) // static analysis build (scripts/isa_phases.py): an assembler comment at every phase boundary, no code
#define RTMI_PH(k) asm volatile("; PHASE_END " #k);
#define RTMI_PH_LANES(k, lt, tr) asm volatile("; PHASE_END " #k);
#else
#define RTMI_PH(k)
#define RTMI_PH_LANES(k, lt, tr)
#endif
// FP64 square root.  The device libm's correctly rounded sqrt is v_rsq_f64 + nine fma/mul (Goldschmidt, two residual corrections)
// wrapped in 14 more instructions: a 2^256 pre-scale for arguments below 2^-767 (whose residuals would underflow), the matching
// post-scale and a v_cmp_class fix-up for 0 / inf.  When every active lane's argument is a finite number >= 2^-767 -- every
// lane of a render; one integer compare on the high word decides -- the wrapper does nothing, and the core sequence alone returns
// the same bits.  The choice is PER LANE (a lane's result is a function of its own argument only, whatever its wave-mates hold):
// the libm call sits in a branch the wave skips unless one of its lanes needs it.
__device__ inline double rt_sqrt(double x) {
    const unsigned hi = (unsigned)__double2hiint(x);
    if (__builtin_expect(hi - 0x10000000u >= 0x7ff00000u - 0x10000000u, 0)) return ::sqrt(x); // hi word of 2^-767 = 0x10000000
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = ::fma(-h, g, 0.5);
    g = ::fma(g, r, g);
    h = ::fma(h, r, h);
    g = ::fma(::fma(-g, g, x), h, g);
    g = ::fma(::fma(-g, g, x), h, g);
    return g;
}

// FP64 atan2 / asin of the sphere uv (get-sphere-uv, hitable.clj:128-139: phi = atan2(z, x), theta = asin(y)).  The generic device libm spends two
// v_mov per polynomial coefficient (a VALU FP64 instruction takes one literal-free scalar operand at most) -- ~75 of the ~280
// instructions of the uv block.  Here the coefficients sit in a constant-memory table the wave reads with a few scalar loads and
// the fma's take them as SGPR pairs.  atan2 divides ONCE: with a = min(|x|,|y|), b = max(|x|,|y|) the argument reduction about
// 0.75 is (a - 0.75 b) / (b + 0.75 a) (two fma's), not (q - 0.75) / (1 + 0.75 q) of a quotient q = a / b.  Polynomials: the
// fdlibm e_atan / e_asin minimax fits (11-term odd atan on |x| <= 7/16, 6/4 rational asin on t <= 1/4).  Both agree with a
// host libm within 2 ulp (<= 4.5e-16 absolute) over the unit sphere -- they only feed texture coordinates.
__device__ __constant__ double kTrig[32] = {
    // [0..10] atan: aT0..aT10
    3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
    9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
    4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02,
    // [11,12] atan(0.75) hi, lo   [13,14] pi/2 hi, lo   [15,16] pi hi, lo
    0x1.4978fa3269ee1p-1, 0x1.2419a87f2a458p-56, 1.57079632679489655800e+00, 6.12323399573676603587e-17, 3.1415926535897931160e+00,
    1.2246467991473531772e-16,
    // [17..22] asin: pS0..pS5   [23..26] qS1..qS4
    1.66666666666666657415e-01, -3.25565818622400915405e-01, 2.01212532134862925881e-01, -4.00555345006794114027e-02,
    7.91534994289814532176e-04, 3.47933107596021167570e-05, -2.40339491173441421878e+00, 2.02094576023350569471e+00,
    -6.88283971605453293030e-01, 7.70381505559019352791e-02,
    // [27] RN(1 / (2 pi))   [28] RN(1 / pi)   [29] 2 pi   [30] pi
    0x1.45f306dc9c883p-3, 0x1.45f306dc9c883p-2, 6.283185307179586, 3.141592653589793, 0.0};
// The table pointer passes through an empty asm so that the scalar loads stay next to their use: hoisted out of the path loop
// (the whole kernel) the 54 SGPRs would be spilled to VGPR lanes and read back with one v_readlane per dword -- the v_mov's again.
typedef const __attribute__((address_space(4))) double *TrigTable; // constant address space: scalar loads
__device__ inline TrigTable trig_table() {
    TrigTable K = (TrigTable)kTrig;
    asm volatile("" : "+s"(K));
    return K;
}
// a * b + c with c in an SGPR pair.  The compiler selects the two-address v_fmac (addend tied to the destination) and copies an
// SGPR addend to VGPRs with two v_mov first; the three-address form takes it in place.
__device__ inline double fma_s(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ inline double rt_atan2(double y, double x, TrigTable K) {
    double c_hi = K[11], c_lo = K[12], h_hi = K[13], h_lo = K[14], p_hi = K[15], p_lo = K[16];
    asm volatile("" : "+s"(c_hi), "+s"(c_lo), "+s"(h_hi), "+s"(h_lo), "+s"(p_hi), "+s"(p_lo)); // loaded here, not inside a branch
    const double ax = ::fabs(x), ay = ::fabs(y);
    const bool swap = ay > ax;
    const double a = swap ? ax : ay, b = swap ? ay : ax;
    const bool far = a > 0.4375 * b; // reduce about 0.75: |xr| <= 0.235 either way
    const bool zero = b == 0.0;      // atan2(+-0, +-0)
    double num = far ? ::fma(-0.75, b, a) : a, den = far ? ::fma(0.75, a, b) : b;
    den = zero ? 1.0 : den;
    const double xr = num / den, z = xr * xr, w = z * z;
    const double s1 = z * fma_s(w, fma_s(w, fma_s(w, fma_s(w, ::fma(w, K[10], K[8]), K[6]), K[4]), K[2]), K[0]);
    const double s2 = w * fma_s(w, fma_s(w, fma_s(w, ::fma(w, K[9], K[7]), K[5]), K[3]), K[1]);
    const double hi = far ? c_hi : 0.0, lo = far ? c_lo : 0.0;
    double r = hi - (::fma(xr, s1 + s2, -lo) - xr);
    const double rs = h_hi - (r - h_lo);
    r = swap ? rs : r;
    const bool neg = (x < 0.0) | (zero & (bool)__builtin_signbit(x));
    const double rn = p_hi - (r - p_lo);
    r = neg ? rn : r;
    return ::copysign(r, y);
}
__device__ inline double rt_asin(double x, TrigTable K) {
    const double ax = ::fabs(x);
    const bool small = ax < 0.5;
    const double t = small ? x * x : ::fma(ax, -0.5, 0.5); // (1 - |x|) / 2, exact
    const double p = t * fma_s(t, fma_s(t, fma_s(t, fma_s(t, ::fma(t, K[22], K[21]), K[20]), K[19]), K[18]), K[17]);
    const double q = ::fma(t, fma_s(t, fma_s(t, ::fma(t, K[26], K[25]), K[24]), K[23]), 1.0);
    const double r = p / q, s = rt_sqrt(t); // |x| > 1: t < 0, NaN like libm
    const double big = K[13] - ::fma(2.0, ::fma(s, r, s), -K[14]);
    return ::copysign(small ? ::fma(ax, r, ax) : big, x);
}
// log of a uniform draw u in [0, 1) (ConstantMedium's free-flight distance, hitable.clj:529: (Math/log (rand))).  The generic device log is ~100 instructions
// of double-double arithmetic for any argument; a draw is 0 or a normal number in [2^-53, 1), so the fdlibm e_log kernel applies without its special cases
// (x = 2^k (1 + f), sqrt(1/2) < 1 + f < sqrt(2), s = f / (2 + f), log(1 + f) = f - (hfsq - s (hfsq + R(s^2)))): < 1 ulp, like a JVM's or a host libm's own log --
// the one operation of a medium hit that is not bit-comparable across math libraries anyway.  log(0) = -inf (no hit: the distance is +inf).
__device__ inline double rt_log_unit(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    int hx = __double2hiint(x);
    const unsigned lx = (unsigned)__double2loint(x);
    if (!(x > 0.0) || hx >= 0x7ff00000) return x == 0.0 ? -__builtin_inf() : ::log(x); // 0, and what a draw never is (negative, inf, NaN)
    if (hx < 0x00100000) return ::log(x);                                                // (denormal: never a draw)
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;
    const double m = __hiloint2double(hx | (i ^ 0x3ff00000), (int)lx); // 1 + f in [sqrt(1/2), sqrt(2))
    k += i >> 20;
    const double f = m - 1.0, dk = (double)k;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * ::fma(w, ::fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * ::fma(w, ::fma(w, ::fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    return ::fma(dk, ln2_hi, -((hfsq - ::fma(s, hfsq + R, dk * ln2_lo)) - f));
}

// x / c for a constant c with rc = RN(1 / c): q = RN(x rc) is within an ulp, r = x - c q is exact in one fma, and RN(q + r rc) is the
// correctly rounded quotient (Markstein) -- the same value as the IEEE division the reference performs, in 3 instructions, not ~14
__device__ inline double div_const(double x, double c, double rc) {
    const double q = x * rc;
    return ::fma(::fma(-c, q, x), rc, q);
}

// t = n / a for the roots of one ray: a = d.d is the same for every sphere the ray meets (hitable.clj:186,192,200 divide by it per
// root).  The IEEE division the compiler emits is: scale operands (v_div_scale x2), r = rcp(a) refined by two Newton steps
// (4 fma), q = n r, e = n - a q, t = q + e r (v_div_fmas), special cases (v_div_fixup) -- 11 instructions + hazard nops, and the
// scaling / fix-up only act on zero, infinite, NaN, denormal or far-apart operands.  Quot keeps the refined reciprocal of a per ray
// and a root costs the last three operations, bit for bit the same quotient whenever the division would not have scaled, i.e.
// for 2^-969 <= |n| < 2^568 when 2^-200 <= a <= 2^200.  Outside that range the quotients may differ, but then |t| < 2^-768 or
// |t| >= 2^368 (or t is NaN where the division gives +-inf or 0): with 2^-300 <= t-min and best-so-far <= 2^200 either value fails
// the same tests (t > t-min, t < best: hitable.clj:195,203), so nothing the scan keeps can differ.  `fast` is decided PER LANE from
// the lane's own a (and the t-min / t-max of the call): a ray's roots never depend on which rays share its wave, so a render is
// reproducible run to run whatever the work queue deals to a wave.  (Correct rounding of the fast form: r is the reciprocal after two
// Newton steps, |r - 1/a| <= 2^-104.9 r -- the same r the IEEE sequence holds at that point -- so q = RN(n r) is within 1 ulp of n/a,
// e = n - a q is exact in one fma, and RN(q + e r) is the correctly rounded quotient by Markstein's theorem; it is the IEEE
// sequence's own last three operations, probed against the quotient on hard-to-round operands by test_per_ray_reciprocal_division.)
template <typename R> struct Quot {
    R a;
    __device__ inline R operator()(R n) const { return n / a; }
};
template <> struct Quot<double> {
    double a, r;
    bool fast;
    __device__ inline double operator()(double n) const {
        if (__builtin_expect(!fast, 0)) return n / a;
        const double q = n * r;
        return ::fma(::fma(-a, q, n), r, q);
    }
};
template <typename R> __device__ inline Quot<R> make_quot(R a, R, R) { return Quot<R>{a}; }
template <> __device__ inline Quot<double> make_quot<double>(double a, double tmin, double tmax) {
    Quot<double> q;
    q.a = a;
    const unsigned hi = (unsigned)__double2hiint(a);
    q.fast = (hi - (823u << 20) < (400u << 20)) && tmin >= 0x1p-300 && tmax <= 0x1p200; // biased exponent in [823, 1223)
    double r = __builtin_amdgcn_rcp(a);
    r = ::fma(r, ::fma(-a, r, 1.0), r);
    q.r = ::fma(r, ::fma(-a, r, 1.0), r);
    return q;
}
template <typename R> __device__ inline R quot(R n, R a) { return n / a; }
template <typename R> __device__ inline R quot(R n, const Quot<R> &a) { return a(n); }
template <typename R> __device__ inline R coef(R a) { return a; }
template <typename R> __device__ inline R coef(const Quot<R> &a) { return a.a; }

#ifdef RTMI_HIST
__device__ unsigned long long g_hist[200]; // diagnostic: [0..64] inner trips by active lanes, [64..128] leaf phases, [128..192] outer iterations
#endif
template <typename R> struct Real;
template <> struct Real<double> {
    // k = z >> 11 (53 bits) as a double: hi * 2^32 + lo in one fma (exact: k needs 53 bits), then k * 2^-53 (exact)
    __device__ static inline double k53(u64 z) { const u64 k = z >> 11; return ::fma((double)(unsigned)(k >> 32), 4294967296.0, (double)(unsigned)k); }
#ifndef RTMI_RNG_BITS
#define RTMI_RNG_BITS 1
#endif
#if RTMI_RNG_BITS
    // The same two values without an integer -> double conversion (two v_cvt_f64_u32 + add + fma per draw): with k = z >> 11 = b 2^52 + f
    // (b = its top bit, f = its low 52 bits) the double whose bits are (exponent of 2^e | f) is 2^e (1 + f 2^-52) exactly, and
    //   uniform   = k 2^-53     = (1/2 + f 2^-53) - (b ? 0 : 1/2)      symmetric = k 2^-52 - 1 = (1 + f 2^-52) - (b ? 1 : 2)
    // are single subtractions of doubles on the same 2^-53 (2^-52) grid with a result below 1 in magnitude: exact, like the forms below.
    __device__ static inline double from_bits(u64 z, unsigned expo, unsigned sub_b1, unsigned sub_b0) {
        const unsigned hi = (unsigned)(z >> 32), lo = (unsigned)z;
        const double m = __hiloint2double((int)(expo | ((hi >> 11) & 0x000fffffu)), (int)__builtin_amdgcn_alignbit(hi, lo, 11));
        return m - __hiloint2double((int)hi < 0 ? (int)sub_b1 : (int)sub_b0, 0);
    }
    __device__ static inline double uniform(u64 z) { return from_bits(z, 0x3fe00000u, 0u, 0x3fe00000u); }
    __device__ static inline double symmetric(u64 z) { return from_bits(z, 0x3ff00000u, 0x3ff00000u, 0x40000000u); }
#else
    __device__ static inline double uniform(u64 z) { return k53(z) * (1.0 / 9007199254740992.0); }
    // 2 * uniform - 1 (util.clj:35-36, 46-48) in one fma: k * 2^-52 - 1 = (k - 2^52) * 2^-52 is a 53-bit integer times a power of two,
    // i.e. exact, like the two exact operations (* 2.0 u) and (- ... 1.0) it replaces
    __device__ static inline double symmetric(u64 z) { return ::fma(k53(z), 1.0 / 4503599627370496.0, -1.0); }
#endif
    __device__ static inline double tmax() { return 3.4028234663852886e38; } // Float/MAX_VALUE, core.clj:25
    __device__ static inline double pi() { return 3.141592653589793; }
    __device__ static inline double sqrt_(double x) { return rt_sqrt(x); }
    __device__ static inline double sqrt_lib(double x) { return ::sqrt(x); } // no wave-level branch: for loops the compiler unrolls
    __device__ static inline double sin_(double x) { return ::sin(x); }
    // get-sphere-uv (hitable.clj:128-139): u = 1 - (phi + pi) / (2 pi), v = (theta + pi/2) / pi; the two divisions by constants
    // are the correctly rounded quotients (div_const)
    __device__ static inline void sphere_uv(double nx, double ny, double nz, double *u, double *v) { *u = sphere_u(nx, nz); *v = sphere_v(ny); }
    __device__ static inline double sphere_u(double nx, double nz) { const TrigTable K = trig_table(); return 1.0 - div_const(rt_atan2(nz, nx, K) + K[30], K[29], K[27]); }
    __device__ static inline double sphere_v(double ny) { const TrigTable K = trig_table(); return div_const(rt_asin(ny, K) + K[13], K[30], K[28]); }
    __device__ static inline double pow_(double x, double y) { return ::pow(x, y); }
};
template <> struct Real<float> {
    __device__ static inline float uniform(u64 z) { return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f); }
    __device__ static inline float symmetric(u64 z) { return 2.0f * uniform(z) - 1.0f; } // both operations exact (24-bit k)
    __device__ static inline float tmax() { return 3.4028234663852886e38f; }
    __device__ static inline float pi() { return 3.141592653589793f; }
    __device__ static inline float sqrt_(float x) { return ::sqrtf(x); }
    __device__ static inline float sqrt_lib(float x) { return ::sqrtf(x); }
    __device__ static inline float sin_(float x) { return ::sinf(x); }
    __device__ static inline void sphere_uv(float nx, float ny, float nz, float *u, float *v) { *u = sphere_u(nx, nz); *v = sphere_v(ny); }
    __device__ static inline float sphere_u(float nx, float nz) { return 1.0f - (::atan2f(nz, nx) + pi()) / (2.0f * pi()); }
    __device__ static inline float sphere_v(float ny) { return (::asinf(ny) + pi() / 2.0f) / pi(); }
    __device__ static inline float pow_(float x, float y) { return ::powf(x, y); }
};

// x^5 for Schlick's approximation (shader.clj:69-74: (Math/pow (- 1.0 cosine) 5.0)).  Math/pow and a host libm pow are
// (almost always) correctly rounded; a generic device pow() is ~200 FP64 instructions and no closer.  Three double-double
// products keep the error below 2^-100 before the final rounding, i.e. the correctly rounded power except for near-ties.
__device__ inline double pow5(double x) {
    const double h2 = x * x, l2 = ::fma(x, x, -h2);
    const double h4 = h2 * h2, l4 = ::fma(h2, h2, -h4) + 2.0 * (h2 * l2);
    const double h5 = h4 * x, l5 = ::fma(h4, x, -h5) + l4 * x;
    return h5 + l5;
}
__device__ inline float pow5(float x) { const double d = (double)x, d2 = d * d; return (float)(d2 * d2 * d); } // 3 roundings at 2^-53, then one to float

// ---- per-path state: the loop/recur state of `color` (core.clj:23) plus the sample's stream -------
template <typename R> struct Path {
    R ox, oy, oz, dx, dy, dz, time; // ray map {:origin :direction :time}, util.clj:13-16
    R ar, ag, ab;                   // atten
    // accum is not carried: every Shader whose emitted is non-zero (DiffuseLight, shader.clj:114-119) returns nil from scatter,
    // so accum is still (0 0 0) when the path's last segment adds atten * emitted (core.clj:37-39) -- see scatter_emit's `emit`
    u64 key;
    u64 rs;       // running stream state = key + GOLD * ctr  (bits(key, d) = mix64(key + GOLD*(d+1)): one add per draw)
    unsigned ctr; // draws consumed (reported by the probes)
    int depth;
};
template <typename R> __device__ inline void seed_stream(Path<R> &P, u64 key, unsigned ctr0) { P.key = key; P.ctr = ctr0; P.rs = key + RTMI_GOLD * (u64)ctr0; }

template <typename R> __device__ inline R next_uniform(Path<R> &P) {
    P.rs += RTMI_GOLD;
    P.ctr++;
    return Real<R>::uniform(mix64(P.rs));
}

template <typename R> __device__ inline R next_symmetric(Path<R> &P) { // (dec (* 2.0 (rand))) of the rejection samplers
    P.rs += RTMI_GOLD;
    P.ctr++;
    return Real<R>::symmetric(mix64(P.rs));
}

template <typename R> __device__ inline R dot3(R ax, R ay, R az, R bx, R by, R bz) { return (ax * bx + ay * by) + az * bz; }

// util.clj:43-52 rand-in-unit-sphere (x, y, z drawn in that order; retry while p.p >= 1.0)
template <typename R> __device__ inline void rand_in_unit_sphere(Path<R> &P, R &x, R &y, R &z) {
    for (;;) {
        x = next_symmetric(P);
        y = next_symmetric(P);
        z = next_symmetric(P);
        if (!(dot3(x, y, z, x, y, z) >= R(1.0))) return;
    }
}
// The same for the lanes with `need`, evaluated COOPERATIVELY when the whole wave is here (every lane of the wave calls; the
// lanes without `need` help).  In the plain loop a wave retries until its unluckiest lane accepts (acceptance pi/6: ~5 trips for
// ~35 lanes) while the lanes that already hold a sample idle.  The stream is random access -- candidate c of a lane's stream is
// draws 3c+1 .. 3c+3 after its current position -- so after RTMI_SAMPLER_PLAIN ordinary trips the n lanes still without a sample
// ("victims") get K = 64/n candidates each evaluated at once, one per lane: lane L takes candidate floor(L/n) of victim L mod n.
// A victim takes its FIRST accepted candidate (util.clj:49-52: candidates are tried in order) and advances its stream past it,
// i.e. by exactly the draws the sequential loop would have consumed; with no accepted candidate it advances by 3K and goes on.
// kStrideMask[n] = sum over q < 64/n of 2^(q n)  (n = 1..64; scalar load, n is wave-uniform)
__device__ __constant__ const unsigned long long kStrideMask[65] = {0ull, 0xffffffffffffffffull, 0x5555555555555555ull, 0x1249249249249249ull, 0x1111111111111111ull, 0x84210842108421ull, 0x41041041041041ull, 0x102040810204081ull, 0x101010101010101ull, 0x40201008040201ull, 0x4010040100401ull, 0x100200400801ull, 0x1001001001001ull, 0x8004002001ull, 0x40010004001ull, 0x200040008001ull, 0x1000100010001ull, 0x400020001ull, 0x1000040001ull, 0x4000080001ull, 0x10000100001ull, 0x40000200001ull, 0x400001ull, 0x800001ull, 0x1000001ull, 0x2000001ull, 0x4000001ull, 0x8000001ull, 0x10000001ull, 0x20000001ull, 0x40000001ull, 0x80000001ull, 0x100000001ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull, 0x1ull};
#ifndef RTMI_COOP_SAMPLER
#define RTMI_COOP_SAMPLER 1
#endif
#ifndef RTMI_SAMPLER_PLAIN
#define RTMI_SAMPLER_PLAIN 2
#endif
template <typename R> __device__ inline void rand_in_unit_sphere_wave(Path<R> &P, bool need, R &x, R &y, R &z) {
#if RTMI_COOP_SAMPLER
    if (__ballot(1) != ~0ull) { // part of the wave is elsewhere (it cannot help): the plain loop
        if (need) rand_in_unit_sphere(P, x, y, z);
        return;
    }
    const int lane = threadIdx.x & 63;
    bool pending = need;
    for (int trip = 0; trip < RTMI_SAMPLER_PLAIN; ++trip) {
        if (__ballot(pending) == 0) return;
        if (pending) {
            x = next_symmetric(P); y = next_symmetric(P); z = next_symmetric(P);
            pending = dot3(x, y, z, x, y, z) >= R(1.0);
            RTMI_PH(PH_SAMPLER)
        }
    }
    u64 rem = __ballot(pending);
    while (rem) { // wave-uniform
        const int n = __popcll(rem), K = 64 / n;
        const int vrank = __popcll(rem & ((1ull << lane) - 1ull));
        // victims to the front (by rank), everyone else behind them: a permutation of the 64 lanes; lane i < n then knows victim i
        const int dest = pending ? vrank : n + (lane - vrank);
        const int tbl = __builtin_amdgcn_ds_permute(dest << 2, lane);
        const unsigned M = (unsigned)(65536.0f / (float)n) + 1u;          // (L * M) >> 16 = floor(L / n) for L < 64, n <= 64
        const int c = (int)(((unsigned)lane * M) >> 16), j = lane - c * n; // this lane: candidate c of victim j
        const int vic = __builtin_amdgcn_ds_bpermute(j << 2, tbl);
        const u64 rs_c = __shfl(P.rs, vic) + RTMI_GOLD * (u64)(3 * c);
        const R cx = Real<R>::symmetric(mix64(rs_c + RTMI_GOLD)), cy = Real<R>::symmetric(mix64(rs_c + 2ull * RTMI_GOLD)),
                cz = Real<R>::symmetric(mix64(rs_c + 3ull * RTMI_GOLD));
        const u64 acc = __ballot(c < K && !(dot3(cx, cy, cz, cx, cy, cz) >= R(1.0)));
        const u64 sm = kStrideMask[n]; // the lanes that hold one victim's candidates, relative to its rank: bits 0, n, 2n, ..., (K-1) n
        const u64 mine = pending ? ((acc >> vrank) & sm) : 0ull;
        const int first = mine ? __ffsll((long long)mine) - 1 : 0;
        const int src = mine ? vrank + first : lane;
        const R fx = __shfl(cx, src), fy = __shfl(cy, src), fz = __shfl(cz, src);
        if (pending) {
            const unsigned used = mine ? 3u * (unsigned)(__popcll(sm & ((1ull << first) - 1ull)) + 1) : 3u * (unsigned)K;
            P.rs += RTMI_GOLD * (u64)used;
            P.ctr += used;
            if (mine) { x = fx; y = fy; z = fz; pending = false; }
        }
        RTMI_PH(PH_SAMPLER) // a cooperative round: every lane evaluates a candidate
        rem = __ballot(pending);
    }
#else
    if (need) rand_in_unit_sphere(P, x, y, z);
#endif
}

// util.clj:32-41 rand-in-unit-disk
template <typename R> __device__ inline void rand_in_unit_disk(Path<R> &P, R &x, R &y) {
    for (;;) {
        x = next_symmetric(P);
        y = next_symmetric(P);
        if (!(dot3(x, y, R(0), x, y, R(0)) >= R(1.0))) return;
    }
}

// camera.clj:8-16 (PinholeCamera.get-ray) and camera.clj:35-48 (ThinLensCamera.get-ray)
template <typename R> __device__ inline void get_ray(SceneRef sc, R s, R t, Path<R> &P) {
    // The camera is loop-invariant, so LICM would hoist its 24 loads out of the path loop into long-lived VGPRs (which
    // then spill).  Laundering the address through an empty asm pins the s_loads here, at the (rare) use site.
    unsigned long long cam_addr = (unsigned long long)(&sc.cam[0]);
    asm volatile("" : "+s"(cam_addr));
    const __attribute__((address_space(4))) double *c = (const __attribute__((address_space(4))) double *)cam_addr;
    const R ox = (R)c[0], oy = (R)c[1], oz = (R)c[2];
    // (add lleft (mul s horiz) (mul t vert) (negate origin) ...): left fold
    R dx = (((R)c[3] + (R)c[6] * s) + (R)c[9] * t) + (-ox);
    R dy = (((R)c[4] + (R)c[7] * s) + (R)c[10] * t) + (-oy);
    R dz = (((R)c[5] + (R)c[8] * s) + (R)c[11] * t) + (-oz);
    if (sc.cam_kind == RTMI_CAM_PINHOLE) {
        P.ox = ox; P.oy = oy; P.oz = oz; P.dx = dx; P.dy = dy; P.dz = dz; P.time = R(0);
        return;
    }
    const R lens_radius = (R)c[21] / R(2.0);
    R rx, ry;
    rand_in_unit_disk(P, rx, ry); // consumed even when aperture = 0
    rx = lens_radius * rx; ry = lens_radius * ry;
    const R offx = (R)c[12] * rx + (R)c[15] * ry;
    const R offy = (R)c[13] * rx + (R)c[16] * ry;
    const R offz = (R)c[14] * rx + (R)c[17] * ry;
    P.ox = ox + offx; P.oy = oy + offy; P.oz = oz + offz;
    P.dx = dx + (-offx); P.dy = dy + (-offy); P.dz = dz + (-offz);
    const R t0 = (R)c[22], t1 = (R)c[23];
    P.time = t0 + (t1 - t0) * next_uniform(P);
}

// Sign of sin(x) without evaluating the sine (Checkerboard only needs (neg? (* sin sin sin)), texture.clj:47-48).
// k = rint(x/pi), r = x - k*pi evaluated with a two-term pi (fma: |error| <= 2 ulp(r) + |k| 4e-32); then
// sin(x) = (-1)^k sin(r), |r| <= pi/2 + eps, so sin(x) < 0  <=>  (k odd) xor (r < 0).  No double within |x| <= 1e5 is
// closer than ~1e-19 to a multiple of pi (the classic worst case of double range reduction is 2^-61 relative), so
// the sign of r is decided far above its error, and a correctly signed libm/ocml sin gives the same answer.
// x = +-0 gives sin = +-0 (product not negative); |x| > 1e5, inf and NaN fall back to the real sin.
template <typename R> __device__ inline int sin_sign(R x) { // -1, 0, +1
    const R s = Real<R>::sin_(x);
    return s < R(0) ? -1 : (s > R(0) ? 1 : 0);
}
template <> __device__ inline int sin_sign<double>(double x) {
    if (x == 0.0) return 0;
    if (!(fabs(x) <= 1.0e5)) { const double s = ::sin(x); return s < 0.0 ? -1 : (s > 0.0 ? 1 : 0); }
    const double k = ::rint(x * 0.3183098861837907);
    double r = ::fma(-k, 3.141592653589793, x);
    r = ::fma(-k, 1.2246467991473532e-16, r);
    const bool odd = ((int)k) & 1; // |k| <= 1e5 / pi: one v_cvt_i32_f64 (a 64-bit integer conversion is five instructions)
    return (odd != (r < 0.0)) ? -1 : 1;
}

// perlin.clj:19-50: noise = trilinear hermite blend of the dot products with the 8 surrounding lattice vectors
template <typename R> __device__ inline R perlin_noise(SceneRef sc, R px, R py, R pz) {
    const R fi = (R)::floor((double)px), fj = (R)::floor((double)py), fk = (R)::floor((double)pz);
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    const R u = px - fi, v = py - fj, w = pz - fk;
    const R uu = (u * u) * (R(3.0) - R(2.0) * u), vv = (v * v) * (R(3.0) - R(2.0) * v), ww = (w * w) * (R(3.0) - R(2.0) * w);
    R acc = R(0);
#pragma unroll
    for (int c = 0; c < 8; ++c) { // (for [di dj dk]) with dk fastest; (reduce + ...) in that order
        const int di = c >> 2, dj = (c >> 1) & 1, dk = c & 1;
        const int idx = sc.perlin_perm[(i + di) & 255] ^ sc.perlin_perm[256 + ((j + dj) & 255)] ^ sc.perlin_perm[512 + ((k + dk) & 255)];
        const double *g = sc.perlin_vec + (size_t)idx * 3;
        const R A = di ? uu : R(1.0) - uu, B = dj ? vv : R(1.0) - vv, C = dk ? ww : R(1.0) - ww; // i*uu + (1-i)*(1-uu), exactly
        const R d = dot3(u - (R)di, v - (R)dj, w - (R)dk, (R)g[0], (R)g[1], (R)g[2]);
        const R term = ((A * B) * C) * d;
        acc = c ? acc + term : term;
    }
    return acc;
}
// perlin.clj:52-64
template <typename R> __device__ inline R perlin_turbulence(SceneRef sc, R px, R py, R pz, int depth) {
    R acc = R(0), w = R(1.0);
    for (int i = 0; i < depth; ++i) {
        acc = acc + w * perlin_noise<R>(sc, px, py, pz);
        px = R(2.0) * px; py = R(2.0) * py; pz = R(2.0) * pz;
        w = w / R(2.0);
    }
    return acc < R(0) ? -acc : acc;
}

// Turbulence of the few lanes that need it, evaluated BY THE WAVE.  A marble / turbulence texture is depth x 8 independent lattice terms (perlin.clj:20-64) that
// the lane's own loop evaluates one after the other -- ~150 instructions per octave while the other lanes of the wave wait: make-final's marble sphere (depth 4, a few
// per cent of the screen) is hit by one or two lanes of most waves and cost 7 % of the frame.  Here every requesting lane is served in turn (wave-uniform loop over
// the request mask): its point is broadcast, the ACTIVE lanes each evaluate one (octave, corner) term -- the same operations perlin_noise performs for that corner --
// into the wave's exchange row in LDS, and the requesting lane adds the terms up in perlin_noise's order (corners dk fastest, the first term assigned) and the octaves
// in perlin_turbulence's (acc = acc + w * noise from acc = 0, w halved, |acc| at the end): the same additions of the same values, bit for bit.  Four octaves per
// round (32 terms: the row holds 32 doubles per wave); a point's octave o is 2^o p exactly (v_ldexp: the loop's repeated doubling).
#define RTMI_TURB_ROUND 4
#ifndef RTMI_TURB_WAVE_MAX
#define RTMI_TURB_WAVE_MAX 4 // requests per wave up to which the wave serves them (above: every lane's own loop)
#endif
__shared__ double g_turb_xch[4][8 * RTMI_TURB_ROUND]; // [wave of the workgroup: kernels that shade run 256 threads, RTMI_TRACE_BLOCK]
__device__ inline double perlin_term(SceneRef sc, double px, double py, double pz, int c) { // term c of perlin_noise(p)
    const double fi = ::floor(px), fj = ::floor(py), fk = ::floor(pz);
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    const double u = px - fi, v = py - fj, w = pz - fk;
    const double uu = (u * u) * (3.0 - 2.0 * u), vv = (v * v) * (3.0 - 2.0 * v), ww = (w * w) * (3.0 - 2.0 * w);
    const int di = c >> 2, dj = (c >> 1) & 1, dk = c & 1;
    const int idx = sc.perlin_perm[(i + di) & 255] ^ sc.perlin_perm[256 + ((j + dj) & 255)] ^ sc.perlin_perm[512 + ((k + dk) & 255)];
    const double *g = sc.perlin_vec + (size_t)idx * 3;
    const double A = di ? uu : 1.0 - uu, B = dj ? vv : 1.0 - vv, C = dk ? ww : 1.0 - ww;
    const double d = dot3(u - (double)di, v - (double)dj, w - (double)dk, g[0], g[1], g[2]);
    return ((A * B) * C) * d;
}
__device__ inline double bcast_lane(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// every ACTIVE lane of the wave must call this together (wave-uniform control flow); lanes with want = false only help.  Returns perlin_turbulence(q, depth) in the
// lanes that want it.
__device__ inline double perlin_turbulence_wave(SceneRef sc, bool want, double qx, double qy, double qz, int depth) {
    unsigned long long req = __ballot(want);
    if (req == 0) return 0.0;
    const unsigned long long act = __ballot(1);
    const int lane = threadIdx.x & 63, nact = __popcll(act), rank = __popcll(act & ((1ull << lane) - 1ull));
    double *xch = g_turb_xch[threadIdx.x >> 6];
    double result = 0.0;
    while (req) { // (wave-uniform)
        const int r = __ffsll((long long)req) - 1;
        req &= req - 1;
        const double bx = bcast_lane(qx, r), by = bcast_lane(qy, r), bz = bcast_lane(qz, r);
        const int dep = __builtin_amdgcn_readlane(depth, r);
        double acc = 0.0, wgt = 1.0;
        for (int o0 = 0; o0 < dep; o0 += RTMI_TURB_ROUND) { // (wave-uniform)
            const int items = 8 * min(RTMI_TURB_ROUND, dep - o0);
            for (int w = rank; w < items; w += nact) { // (one trip unless fewer than `items` lanes are active)
                const int o = o0 + (w >> 3);
                xch[w] = perlin_term(sc, __builtin_ldexp(bx, o), __builtin_ldexp(by, o), __builtin_ldexp(bz, o), w & 7);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == r) {
                for (int q = 0; q < items; q += 8) {
                    double n = xch[q];
                    for (int c = 1; c < 8; ++c) n = n + xch[q + c];
                    acc = acc + wgt * n;
                    wgt = wgt / 2.0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier(); // (the next round overwrites the row)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane == r) result = acc < 0.0 ? -acc : acc;
    }
    return result;
}

// texture.clj:14-16, 26-34, 44-50 (Texture.sample); checker children are walked iteratively.
// F4 (only in the EXT kernel instantiations): Perlin noise / turbulence / marble, FlipTextureU/V, ImageMap (texture.clj:60-138)
template <typename R, bool F4 = false> __device__ inline void tex_sample(SceneRef sc, int t, R u, R v, R px, R py, R pz, R &r, R &g, R &b) {
    r = g = b = R(0);
    for (int guard = 0; guard <= sc.n_tex; ++guard) {
        const double *tp = sc.tex_param + (size_t)t * RTMI_TEX_STRIDE;
        const int kind = sc.tex_kind[t];
        if (kind == RTMI_TEX_CONSTANT) {
            r = (R)tp[0]; g = (R)tp[1]; b = (R)tp[2];
            return;
        }
        if (kind == RTMI_TEX_UVGRADIENT) {
            const R omu = R(1.0) - u, omv = R(1.0) - v;
            // a = cu*(1-u) + co*u ; b = cuv*(1-u) + cv*u ; out = b*(1-v) + a*v
            const R a0 = (R)tp[3] * omu + (R)tp[0] * u, a1 = (R)tp[4] * omu + (R)tp[1] * u, a2 = (R)tp[5] * omu + (R)tp[2] * u;
            const R b0 = (R)tp[9] * omu + (R)tp[6] * u, b1 = (R)tp[10] * omu + (R)tp[7] * u, b2 = (R)tp[11] * omu + (R)tp[8] * u;
            r = b0 * omv + a0 * v; g = b1 * omv + a1 * v; b = b2 * omv + a2 * v;
            return;
        }
        if (kind == RTMI_TEX_CHECKER) {
            const R scale = (R)tp[0];
            // (neg? (ereduce * (emap sin (mul scale p)))): negative <=> no factor is zero and an odd number are negative
            const int sx = sin_sign<R>(scale * px), sy = sin_sign<R>(scale * py), sz = sin_sign<R>(scale * pz);
            t = (sx * sy * sz < 0) ? sc.tex_child[2 * t] : sc.tex_child[2 * t + 1];
            continue;
        }
        if (F4) {
            if (kind == RTMI_TEX_PERLIN_NOISE) { // (mul (vec3 1 1 1) (* 0.5 (inc (noise (mul scale p)))))
                const R s = (R)tp[0];
                r = g = b = R(0.5) * (perlin_noise<R>(sc, s * px, s * py, s * pz) + R(1.0));
                return;
            }
            if (kind == RTMI_TEX_PERLIN_TURB) {
                const R s = (R)tp[0];
                r = g = b = R(0.5) * (perlin_turbulence<R>(sc, s * px, s * py, s * pz, (int)tp[1]) + R(1.0));
                return;
            }
            if (kind == RTMI_TEX_MARBLE) { // 0.5 * (inc (sin (+ (* scale pz) (* 10.0 (turbulence p depth)))))
                r = g = b = R(0.5) * (Real<R>::sin_((R)tp[0] * pz + R(10.0) * perlin_turbulence<R>(sc, px, py, pz, (int)tp[1])) + R(1.0));
                return;
            }
            if (kind == RTMI_TEX_FLIP_U) { u = R(1.0) - u; t = sc.tex_child[2 * t]; continue; }
            if (kind == RTMI_TEX_FLIP_V) { v = R(1.0) - v; t = sc.tex_child[2 * t]; continue; }
            if (kind == RTMI_TEX_IMAGE) { // i = (int (* u width)), j = (int (* v height)), rgb / 255.0 (clamped: u = 1.0 throws in the reference)
                const int im = (int)tp[0];
                const int w = sc.image_wh[2 * im], h = sc.image_wh[2 * im + 1];
                int i = (int)(u * (R)w), j = (int)(v * (R)h);
                i = i < 0 ? 0 : (i >= w ? w - 1 : i); j = j < 0 ? 0 : (j >= h ? h - 1 : j);
                const unsigned char *q = sc.image_rgb + sc.image_off[im] + ((size_t)j * w + i) * 3;
                r = (R)q[0] / R(255.0); g = (R)q[1] / R(255.0); b = (R)q[2] / R(255.0);
                return;
            }
        }
        return;
    }
}

// ---- geometry in LDS: {cx, cy, cz, r*r} per static sphere, broadcast-read by the whole wave ---------
template <typename R> struct alignas(16) Prim4 { R cx, cy, cz, r2; };

// hitable.clj:180-207 (Sphere.hit?) x hitable.clj:15-26 (Hitlist reduce), closest-hit form:
// only (t, index) is kept per lane; p / normal / uv are rebuilt once for the winner.
// `ok` uses strict comparisons on both ends (hitable.clj:195,203); the running t-max is the best t
// so far (hitable.clj:20), so among equal t the first item in list order wins.
//
// Exact reduced quadratic.  The reference evaluates b = 2(oc.d), disc = b*b - (4a)c, t = (-b -+ sqrt(disc))/(2a).
// Scaling by 2 and 4 commutes with IEEE rounding (no overflow/underflow at scene scales), so with
// b' = oc.d:  disc = 4*fl(fl(b'b') - fl(a c)) = 4 disc',  sqrt(disc) = 2 sqrt(disc'),  t = fl((-b' -+ sqrt(disc'))/a)
// bit for bit.  The kernels evaluate the primed form (two multiplies fewer per test); scan variant 0 keeps the
// literal form so the parity tests can compare the two on the device.
enum { SCAN_LDS_LITERAL = 0, SCAN_LDS_PIPE = 1, SCAN_SGPR = 2, SCAN_SGPR_CULL = 3, SCAN_BVH = 4 };

template <typename R>
__device__ inline void scan_static(const Prim4<R> *__restrict__ lds, int n, int idx_base, const Path<R> &P, R a, R tmin, R &best_t, int &best_i) {
    const R a2 = R(2.0) * a, a4 = R(4.0) * a;
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
        const Prim4<R> s = lds[i];
        const R ocx = P.ox - s.cx, ocy = P.oy - s.cy, ocz = P.oz - s.cz;
        const R b = R(2.0) * dot3(ocx, ocy, ocz, P.dx, P.dy, P.dz);
        const R c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - s.r2;
        const R disc = b * b - a4 * c;
        if (disc >= R(0)) {
            const R sq = Real<R>::sqrt_lib(disc);
            R t = (-b - sq) / a2;
            bool ok = (t > tmin) && (t < best_t);
            if (!ok) {
                t = (-b + sq) / a2;
                ok = (t > tmin) && (t < best_t);
            }
            if (ok) { best_t = t; best_i = idx_base + i; }
        }
    }
}

// The two roots of one sphere for the lanes whose line meets it (disc' >= 0): hitable.clj:190-207.
// Exact early-out (only used when t-min >= 0): with the origin outside the sphere (c > 0) and the centre behind
// the ray (b' = oc.d > 0) both roots are <= 0: -b'-sq < 0, and sq = sqrt(fl(fl(b'b') - fl(ac))) <= sqrt(fl(b'b')) = b'
// (a correctly rounded sqrt of a correctly rounded square returns |x|), so -b'+sq <= 0: `t > t-min` fails for both.
template <typename R, typename A>
__device__ inline void sphere_roots(R bq, R cq, R disc, const A &a, R tmin, bool behind_ok, R &best_t, int &best_i, int idx) {
    if (behind_ok && bq > R(0) && cq > R(0)) return;
    const R sq = Real<R>::sqrt_(disc);
    R t = quot<R>(-bq - sq, a);
    bool ok = (t > tmin) && (t < best_t);
    if (!ok) {
        t = quot<R>(-bq + sq, a);
        ok = (t > tmin) && (t < best_t);
    }
    if (ok) { best_t = t; best_i = idx; }
}

template <typename R> __device__ inline void sphere_test(const Prim4<R> &s, const Path<R> &P, R a, R &bq, R &cq, R &disc) {
    const R ocx = P.ox - s.cx, ocy = P.oy - s.cy, ocz = P.oz - s.cz;
    bq = dot3(ocx, ocy, ocz, P.dx, P.dy, P.dz);
    cq = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - s.r2;
    disc = bq * bq - a * cq;
}

// Sphere data sources: LDS (per-lane VGPR copies of a broadcast read) or the scalar data cache (wave-uniform
// SGPRs: constant address space + uniform index -> s_load_dwordx8 / x4).
template <typename R> struct LdsPrims {
    const Prim4<R> *p;
    __device__ inline Prim4<R> operator()(int i) const { return p[i]; }
};
template <typename R> struct ScalarPrims {
    typedef R v4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) v4 *cptr;
    cptr p;
    __device__ explicit ScalarPrims(const R *g) : p((cptr)(g)) {}
    __device__ inline Prim4<R> operator()(int i) const {
        const v4 v = p[i];
        Prim4<R> s; s.cx = v.x; s.cy = v.y; s.cz = v.z; s.r2 = v.w;
        return s;
    }
};

// Software-pipelined scan: 4 spheres per group, two groups per trip in ping-pong registers, so the loads of the next
// group are in flight while the current group's 4 x 17 independent VALU ops issue; one wave-level branch per group
// guards the rare root computations.  The source must be readable up to index round_up(n,8)+7 (LDS: indices are
// clamped; HBM arrays are padded with copies of the last sphere): a duplicate of the last sphere can never win
// (strict t < best_t) and is reported under the last sphere's index.
template <typename R> struct Group4 { Prim4<R> s0, s1, s2, s3; };

template <typename R, bool CLAMP, typename Src> __device__ inline Group4<R> load_group(const Src &src, int g, int last) {
    Group4<R> G;
    G.s0 = src(CLAMP ? min(g, last) : g);
    G.s1 = src(CLAMP ? min(g + 1, last) : g + 1);
    G.s2 = src(CLAMP ? min(g + 2, last) : g + 2);
    G.s3 = src(CLAMP ? min(g + 3, last) : g + 3);
    return G;
}

template <typename R>
__device__ inline void test_group(const Group4<R> &G, int g, int last, int idx_base, const Path<R> &P, R a, R tmin, bool behind_ok, R &best_t, int &best_i) {
    R b0, q0, d0, b1, q1, d1, b2, q2, d2, b3, q3, d3;
    sphere_test(G.s0, P, a, b0, q0, d0);
    sphere_test(G.s1, P, a, b1, q1, d1);
    sphere_test(G.s2, P, a, b2, q2, d2);
    sphere_test(G.s3, P, a, b3, q3, d3);
    if ((d0 >= R(0)) | (d1 >= R(0)) | (d2 >= R(0)) | (d3 >= R(0))) {
        if (d0 >= R(0)) sphere_roots(b0, q0, d0, a, tmin, behind_ok, best_t, best_i, idx_base + min(g, last));
        if (d1 >= R(0)) sphere_roots(b1, q1, d1, a, tmin, behind_ok, best_t, best_i, idx_base + min(g + 1, last));
        if (d2 >= R(0)) sphere_roots(b2, q2, d2, a, tmin, behind_ok, best_t, best_i, idx_base + min(g + 2, last));
        if (d3 >= R(0)) sphere_roots(b3, q3, d3, a, tmin, behind_ok, best_t, best_i, idx_base + min(g + 3, last));
    }
}

template <typename R, bool CLAMP, typename Src>
__device__ inline void scan_static_pipe(const Src &src, int n, int idx_base, const Path<R> &P, R a, R tmin, R &best_t, int &best_i) {
    if (n <= 0) return;
    const int last = n - 1;
    const bool behind_ok = tmin >= R(0);
    Group4<R> A = load_group<R, CLAMP>(src, 0, last);
    for (int g = 0; g < n; g += 8) {
        const Group4<R> B = load_group<R, CLAMP>(src, g + 4, last);
        test_group<R>(A, g, last, idx_base, P, a, tmin, behind_ok, best_t, best_i);
        A = load_group<R, CLAMP>(src, g + 8, last);
        test_group<R>(B, g + 4, last, idx_base, P, a, tmin, behind_ok, best_t, best_i);
    }
}

// ---- conservative FP32 cull in front of the exact FP64 test (scan variant SCAN_SGPR_CULL) ------------------------
// The exact test needs 17 FP64 VALU ops per (ray, sphere) (4 SIMD cycles each on CDNA4); almost all of them only
// establish "the line misses this sphere".  The cull evaluates the same discriminant D = (oc.d)^2 - |d|^2 (|oc|^2 - r^2)
// in FP32 (13 ops, 2 cycles each, FMAs allowed) and rejects the sphere only when D_f + tol < 0, where tol bounds
// |D_f - D| + |D_fp64 - D|, so a sphere whose FP64 discriminant is >= 0 is NEVER rejected and the closest hit, its t
// and everything downstream stay bit-identical to the un-culled scan (asserted on the device by the parity tests).
//
// Error bound (u = 2^-24; S = |o|_1 + |c|_1 >= |oc|_2; a = |d|^2; first order, constants rounded up):
//   inputs rounded to float:  |d(oc_k)| <= 2u(|o_k| + |c_k|)  =>  |d(oc)|_2 <= 2uS ;  |d(d)|_2 <= u|d|
//   b = oc.d   (3 ops):        |db| <= (2+1+3) u S|d| = 6u S|d|        =>  |d(b^2)| <= 12u S^2 a
//   q = |oc|^2 - r2 (4 ops):   |dq| <= 4uS^2 + u r2 + 4u(S^2 + r2)     =   8u S^2 + 5u r2
//   a (conversion + 3 ops):    |da| <= 5u a
//   a q (1 op):                |d(aq)| <= 5u a(S^2+r2) + a(8uS^2+5u r2) + u a(S^2+r2) <= 15u a(S^2 + r2)
//   final fma:                 <= 2u a(S^2 + r2)
//   total |D_f - D| <= 29u a(S^2 + r2); the FP64 evaluation's own error is ~2^-29 of that.
// The kernel uses tol = 64u a_f (2|o|_1^2 + 2|c|_1^2 + r2) >= 64u a (S^2 + r2) (2x margin for second-order terms and for
// the float evaluation of tol itself).  Rays whose float image could overflow/underflow (|o|_1 > 1e15, a outside
// [1e-30, 1e30]) or hold NaN take tol = FLT_MAX, i.e. every sphere goes to the exact test.
struct CullRay { float ox, oy, oz, dx, dy, dz, a, A0, A1; };

__device__ inline CullRay make_cull_ray(const Path<double> &P, double a, double t_lo, double t_hi) {
    CullRay c;
    c.ox = (float)P.ox; c.oy = (float)P.oy; c.oz = (float)P.oz;
    c.dx = (float)P.dx; c.dy = (float)P.dy; c.dz = (float)P.dz;
    c.a = fmaf(c.dz, c.dz, fmaf(c.dy, c.dy, c.dx * c.dx));
    const float on = (fabsf(c.ox) + fabsf(c.oy)) + fabsf(c.oz);
    const float ku = 64.0f * 5.9604645e-08f; // 64 * 2^-24
    c.A1 = ku * c.a * 1.0001f;
    c.A0 = c.A1 * (2.0f * on * on);
    const bool safe = (on < 1e15f) && (c.a > 1e-30f) && (c.a < 1e30f) && (a > 1e-30) && (a < 1e30) && (P.time >= t_lo) && (P.time <= t_hi);
    if (!safe) { // NaN compares false too
        c.ox = c.oy = c.oz = c.dx = c.dy = c.dz = 0.0f; c.a = 0.0f; c.A1 = 0.0f; c.A0 = 3.0e38f;
    }
    return c;
}

struct CullGroup { float cx[4], cy[4], cz[4], r2[4], w[4]; };

__device__ inline CullGroup load_cull_group(const float *base, int group) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) f4 *cptr;
    const cptr p = (cptr)(base) + (size_t)group * 5;
    const f4 x = p[0], y = p[1], z = p[2], r = p[3], w = p[4];
    CullGroup G;
    G.cx[0] = x.x; G.cx[1] = x.y; G.cx[2] = x.z; G.cx[3] = x.w;
    G.cy[0] = y.x; G.cy[1] = y.y; G.cy[2] = y.z; G.cy[3] = y.w;
    G.cz[0] = z.x; G.cz[1] = z.y; G.cz[2] = z.z; G.cz[3] = z.w;
    G.r2[0] = r.x; G.r2[1] = r.y; G.r2[2] = r.z; G.r2[3] = r.w;
    G.w[0] = w.x; G.w[1] = w.y; G.w[2] = w.z; G.w[3] = w.w;
    return G;
}

// D_f + tol for sphere k of the group (>= 0: the sphere may be hit, run the exact test)
__device__ inline float cull_disc(const CullGroup &G, int k, const CullRay &c) {
    const float ocx = c.ox - G.cx[k], ocy = c.oy - G.cy[k], ocz = c.oz - G.cz[k];
    const float b = fmaf(ocz, c.dz, fmaf(ocy, c.dy, ocx * c.dx));
    const float q = fmaf(ocz, ocz, fmaf(ocy, ocy, fmaf(ocx, ocx, -G.r2[k])));
    const float t1 = fmaf(c.A1, G.w[k], c.A0);
    const float t2 = fmaf(-c.a, q, t1);
    return fmaf(b, b, t2);
}

// exact test of primitive i (original index; wave-uniform: scalar loads) for the lanes that survived the cull: Sphere /
// UVSphere (hitable.clj:180-207 / 141-168) or MovingSphere (hitable.clj:219-252, centre = lerp(c0, c1, (time-t0)/(t1-t0)))
template <typename R, typename A>
__device__ inline void exact_prim_test(const double *exact12, int i, int idx, const Path<R> &P, const A &a, R tmin, bool behind_ok, R &best_t, int &best_i) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) d4 *cptr;
    const cptr p = (cptr)(exact12) + (size_t)i * 3;
    const d4 q0 = p[0], q2 = p[2];
    Prim4<R> s;
    s.cx = (R)q0.x; s.cy = (R)q0.y; s.cz = (R)q0.z;
    if (sizeof(R) == sizeof(double)) s.r2 = (R)q0.w;
    else { const R rad = (R)q2.z; s.r2 = rad * rad; }
    if (q2.y != 0.0) { // moving (wave-uniform branch)
        const d4 q1 = p[1];
        const R t0 = (R)q1.w, t1 = (R)q2.x;
        const R f = (P.time - t0) / (t1 - t0), omf = R(1.0) - f;
        s.cx = (R)q0.x * omf + (R)q1.x * f; s.cy = (R)q0.y * omf + (R)q1.y * f; s.cz = (R)q0.z * omf + (R)q1.z * f;
    }
    R bq, cq, disc;
    sphere_test(s, P, coef<R>(a), bq, cq, disc);
    if (disc >= R(0)) sphere_roots<R>(bq, cq, disc, a, tmin, behind_ok, best_t, best_i, idx);
}

__device__ inline void cull_test_group(const CullGroup &G, int g, int last, const double *exact12, const Path<double> &P, const CullRay &c,
                                       double a, double tmin, bool behind_ok, double &best_t, int &best_i) {
    const float d0 = cull_disc(G, 0, c), d1 = cull_disc(G, 1, c), d2 = cull_disc(G, 2, c), d3 = cull_disc(G, 3, c);
    if (fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)) >= 0.0f) {
        if (d0 >= 0.0f) exact_prim_test<double>(exact12, g, min(g, last), P, a, tmin, behind_ok, best_t, best_i);
        if (d1 >= 0.0f) exact_prim_test<double>(exact12, g + 1, min(g + 1, last), P, a, tmin, behind_ok, best_t, best_i);
        if (d2 >= 0.0f) exact_prim_test<double>(exact12, g + 2, min(g + 2, last), P, a, tmin, behind_ok, best_t, best_i);
        if (d3 >= 0.0f) exact_prim_test<double>(exact12, g + 3, min(g + 3, last), P, a, tmin, behind_ok, best_t, best_i);
    }
}

// the culled scan over ALL primitives in Hitlist order; best_i is the ORIGINAL primitive index
__device__ inline void scan_all_cull(SceneRef sc, const Path<double> &P, double a, double tmin, double &best_t, int &best_i) {
    const int n = sc.n_all;
    if (n <= 0) return;
    const int last = n - 1;
    const bool behind_ok = tmin >= 0.0;
    const float *cull20 = sc.cull20;
    const double *exact12 = sc.exact12;
    const CullRay c = make_cull_ray(P, a, sc.cull_t_lo, sc.cull_t_hi);
    CullGroup A = load_cull_group(cull20, 0);
    for (int g = 0; g < n; g += 8) {
        const CullGroup B = load_cull_group(cull20, (g >> 2) + 1);
        cull_test_group(A, g, last, exact12, P, c, a, tmin, behind_ok, best_t, best_i);
        A = load_cull_group(cull20, (g >> 2) + 2);
        cull_test_group(B, g + 4, last, exact12, P, c, a, tmin, behind_ok, best_t, best_i);
    }
}

// ---- RTMI_ACCEL_BVH: per-lane BVH traversal with conservative float boxes, exact FP64 leaves -------------------------
// Closest hit is order independent: candidate(prim) = first root if > t-min else second root (hitable.clj:192-207 with the
// running t-max of hitable.clj:20 only ever rejecting non-minimal candidates); ties -> lowest Hitlist index (first wins).
#ifndef RTMI_EXIT_IBALLOT
#define RTMI_EXIT_IBALLOT 1
#endif
#ifndef RTMI_GRID_CHUNK
#define RTMI_GRID_CHUNK 1 // 1: a segment that crosses more than 2 x 2 grid cells is walked in pieces of at most 2 x 2 cells, nearest piece first (0: it descends from the root of the whole tree)
#endif
#define RTMI_BVH_EMPTY ((int)0x80000000) // no child / traversal done; negative like the leaf codes, so "inner node" is one sign test (a leaf code ~(idx | moving << 30) never equals it)
#ifndef RTMI_BVH_STACK
#define RTMI_BVH_STACK 32
#endif
// threads per workgroup of EVERY kernel that traverses the tree (trace and probe kernels are launched with exactly this many): the stride of the
// stack / parked-cursor columns in LDS is then a constant, i.e. an immediate offset of the ds_ instructions instead of address arithmetic
#ifndef RTMI_TRACE_BLOCK
#define RTMI_TRACE_BLOCK 256
#endif
#define RTMI_BVH_STRIDE RTMI_TRACE_BLOCK

template <typename R, typename A>
__device__ inline void sphere_roots_any_order(R bq, R cq, R disc, const A &a, R tmin, bool behind_ok, R &best_t, int &best_i, int idx) {
    if (behind_ok && bq > R(0) && cq > R(0)) return;
    const R sq = Real<R>::sqrt_(disc);
    R t = quot<R>(-bq - sq, a);
    if (!(t > tmin)) t = quot<R>(-bq + sq, a);
    if ((t > tmin) && ((t < best_t) || (t == best_t && idx < best_i))) { best_t = t; best_i = idx; }
}

// exact test of the primitive of leaf code `code` (per-lane: vector loads; a static sphere needs only its first 32 bytes).
// RTMI_F32 reads the same FP64 records and rounds them to float exactly as the flat float scan does: centre (float)c,
// r*r = one float multiply of (float)r (record slot 10), MovingSphere centre lerped in float.
template <typename R, typename A>
__device__ inline void exact_prim_test_lane(const double *exact12, int code, const Path<R> &P, const A &a, R tmin, bool behind_ok, R &best_t, int &best_i) {
    const int bits = ~code;
    const int idx = bits & 0x3fffffff;
    const double2 *g = reinterpret_cast<const double2 *>(exact12 + (size_t)idx * 12);
    const double2 g0 = g[0], g1 = g[1];
    Prim4<R> s;
    s.cx = (R)g0.x; s.cy = (R)g0.y; s.cz = (R)g1.x;
    if (sizeof(R) == sizeof(double)) s.r2 = (R)g1.y;
    else { const R rad = (R)g[5].x; s.r2 = rad * rad; }
    if (bits & 0x40000000) { // MovingSphere
        const double2 g2 = g[2], g3 = g[3], g4 = g[4];
        const R t0 = (R)g3.y, t1 = (R)g4.x;
        const R f = (P.time - t0) / (t1 - t0), omf = R(1.0) - f;
        s.cx = (R)g0.x * omf + (R)g2.x * f; s.cy = (R)g0.y * omf + (R)g2.y * f; s.cz = (R)g1.x * omf + (R)g3.x * f;
    }
    R bq, cq, disc;
    sphere_test(s, P, coef<R>(a), bq, cq, disc);
    if (disc >= R(0)) sphere_roots_any_order<R>(bq, cq, disc, a, tmin, behind_ok, best_t, best_i, idx);
}

// Conservative slab test in float, FMA form: t = plane * inv_d + (-o * inv_d).  What the boxes must absorb, in units of
// U = 2^-24 * obound (obound >= |o|_inf, |c|_inf, r for every ray that starts inside the scene bound):
//   * |fl32(o) - o|, the rounding of c = fl(-o_f * inv) and the rounding of the fma: each equivalent to moving a plane by <= 1 U;
//   * the exact test's own "phantom" zone: the FP64 discriminant b^2 - a q carries an absolute error of up to ~40 u64 a (|oc|^2 +
//     r^2), so it can report a hit for a line passing the centre at p with p - r <= sqrt(40 u64 (|oc|^2 + r^2)) = 6.7e-8 * sqrt(13)
//     * obound = 4 U in the worst case (origin and sphere in opposite corners);
// together <= 7 U: the host rounds the box planes outward and inflates them by 2^-21 * obound = 8 U.  A ray that starts OUTSIDE
// the scene bound (far: e.g. after scattering in a scene-sized ConstantMedium) moves every plane outward by its own
// e = 2^-20 |o|_inf instead (|oc| <= 2 sqrt(3) |o|_inf there, plus one more rounding): the slack e |inv| is folded into the additive
// constant per plane (c_lo for the planes x = lo, c_hi for x = hi; which of the two is the entry plane is the sign of inv).
// The remaining relative error on t (<= 4u with the rounding of inv) moves the computed crossing of a plane along its own axis by
// <= 4u |t d_k| = 4u |plane - o_k| <= 4u (cbound + |o|_inf): every ray moves every plane outward by 8u (cbound + |o|_inf) itself
// (folded into the same per-plane constants, so a node visit pays nothing for it).
// Both children of a node are tested together; the x/y planes go through packed FMAs (v_pk_fma_f32), z as (lo, hi) pairs.
typedef float v2f __attribute__((ext_vector_type(2)));
struct BvhRay { v2f ixy, izz, clxy, chxy, czz; float tmin_lo; bool ok, far, time_ok;
                // Node16 path: which half of a {lo, hi} pair is the ENTRY plane is the sign of the direction, so the pair is rotated
                // by sh (0 or 16 bits) and entry / exit distances come out of the fma directly -- no min / max per axis
                unsigned shx, shy, shz; float cex, cey, cez, cxx, cxy, cxz; };
// 32-byte node format (DevScene::bvh_node16): the planes as IEEE half, rounded outward on the host, consumed by v_fma_mix_f32
// without a conversion instruction: half the bytes per step through the texture-address / L1 path (C2 -3 %, C3 -6 %).  The host
// picks it when rounding to half grows the boxes' total area by less than a quarter (coordinates small against object sizes).
struct __attribute__((aligned(16))) Node16 { _Float16 p[12]; int cl, cr; }; // l: lo.x hi.x lo.y hi.y lo.z hi.z, r: the same, left, right
// hit  <=>  max(entry distances, t-min) <= min(exit distances, best t so far)
__device__ inline bool slab_hit6(float ax, float bx, float ay, float by, float az, float bz, const float tmin_lo, const float best_hi, float &tnear) {
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin_lo));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best_hi));
    tnear = tn;
    return tn <= tf;
}
__device__ inline bool slab_hit(const v2f a, const v2f b, const v2f z, const float tmin_lo, const float best_hi, float &tnear) {
    const float tn = fmaxf(fmaxf(fminf(a.x, b.x), fminf(a.y, b.y)), fmaxf(fminf(z.x, z.y), tmin_lo));
    const float tf = fminf(fminf(fmaxf(a.x, b.x), fmaxf(a.y, b.y)), fminf(fmaxf(z.x, z.y), best_hi));
    tnear = tn;
    return tn <= tf;
}

__device__ inline float min_raw(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// a float >= x within two ulps (any finite x; values beyond FLT_MAX give FLT_MAX, as the closest hit so far never exceeds the caller's t-max = Float/MAX_VALUE):
// RN(x) is within 2^-24 |x| (2^-150 in the denormal range) of x, so RN(x) (1 + 2^-23) + 2^-120 is above it.  Used for the traversal's bound of the closest
// hit so far, refreshed after every exact-test phase: a bound needs no exactness, and the exact form (float_up) is a dozen instructions and a branch.
__device__ inline float float_above(double x) { const float f = (float)x; return fminf(fmaf(fabsf(f), 0x1p-23f, f) + 0x1p-120f, 3.4028235e38f); }
__device__ inline float float_up(double x) { // smallest float >= x (x finite, |x| < FLT_MAX)
    float f = (float)x;
    if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + (f >= 0.0f ? 1u : (unsigned)-1));
    return f;
}

// what the float traversal needs to know about a ray; ok = false: take the exact flat scan instead
template <typename R>
__device__ inline BvhRay make_bvh_ray(SceneRef sc, const Path<R> &P, R a, R tmin) {
    BvhRay r;
    const float ox = (float)P.ox, oy = (float)P.oy, oz = (float)P.oz;
    const float dx = (float)P.dx, dy = (float)P.dy, dz = (float)P.dz;
    const float dmax = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    const float omax = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
    // not boundable in float: non-finite values, a vanishing direction component (1/d would overflow), absurd magnitudes
    r.ok = (omax < 1e15f) && (dmax < 1e15f) && (dmax > 1e-15f) && (fminf(fminf(fabsf(dx), fabsf(dy)), fabsf(dz)) >= 1e-12f * dmax) &&
           (a > R(1e-30)) && (a < R(1e30)) && (tmin > R(-1e30)) && (tmin < R(1e30)) && (sc.bvh_obound >= 0.0f);
    r.far = omax > sc.bvh_obound;
    r.time_ok = (P.time >= sc.cull_t_lo) && (P.time <= sc.cull_t_hi); // else the MovingSphere boxes do not bound this ray's spheres
    // v_rcp_f32: <= 1 ulp = 2u of relative error; with the rounding of d (u) and of the fma (u) the "<= 4u" of slab_hit's budget
    const float ix = __builtin_amdgcn_rcpf(dx), iy = __builtin_amdgcn_rcpf(dy), iz = __builtin_amdgcn_rcpf(dz);
    const float cx = -ox * ix, cy = -oy * iy, cz = -oz * iz;
    // RTMI_F32: the boxes must also catch what the FLOAT sphere test calls a hit.  (a) The float spheres (rounded centres, float
    // r*r, float lerp) may stick out of the boxes built from the FP64 geometry by a few 2^-24 |c|: 2^-20 of the larger of |o|
    // and the scene bound.  (b) The float discriminant b^2 - a q carries an absolute error of up to ~40u a(|oc|^2 + r^2)
    // (u = 2^-24), so the test can report a hit for a line that passes the centre at p with p^2 <= r^2 + 40u(|oc|^2 + r^2), i.e.
    // up to 1.55e-3 sqrt(|oc|^2 + r^2) outside the sphere; with |oc| <= sqrt(3)(|o|_inf + cbound) and r <= cbound (cbound = the
    // largest coordinate of any box in the tree) that is < 3.1e-3 (|o|_inf + cbound): every plane moves out by 4e-3 of that.
    const float e = (sizeof(R) == sizeof(float) ? fmaxf(omax, sc.bvh_obound) * (1.0001f / 1048576.0f) + 4.0e-3f * (omax + sc.bvh_cbound)
                                                : (r.far ? omax * (1.0001f / 1048576.0f) : 0.0f)) +
                    (8.0f * 5.9604645e-08f * 1.0001f) * (omax + sc.bvh_cbound); // the relative error of t as a plane shift (see slab_hit)
    // signed slack: the plane x = lo is the entry plane when inv >= 0 (entry distances are lowered, exit distances raised)
    const float ex = e * ix, ey = e * iy, ez = e * iz; // = e |inv| * sign(inv)
    r.ixy = v2f{ix, iy}; r.izz = v2f{iz, iz};
    r.clxy = v2f{cx - ex, cy - ey}; r.chxy = v2f{cx + ex, cy + ey};
    r.czz = v2f{cz - ez, cz + ez};
    r.shx = ix < 0.0f ? 16u : 0u; r.shy = iy < 0.0f ? 16u : 0u; r.shz = iz < 0.0f ? 16u : 0u;
    r.cex = cx - fabsf(ex); r.cey = cy - fabsf(ey); r.cez = cz - fabsf(ez); // entry planes move towards the origin of the ray,
    r.cxx = cx + fabsf(ex); r.cxy = cy + fabsf(ey); r.cxz = cz + fabsf(ez); // exit planes away from it
    r.tmin_lo = -float_up(-(double)tmin); // (a compile-time constant in the render kernel: t-min = 0.001, core.clj:25)
    return r;
}

// Per-lane traversal, "while-while": every lane descends until it holds a leaf (or is done); then all lanes holding a
// leaf run the exact FP64 test together -- the expensive FP64 code is not interleaved with box tests.  Stack: LDS, one
// column per thread (stack[level * blockDim.x + tid]: conflict-free).  leaf(code) runs the exact test, best() returns the
// current float upper bound of the closest t.
#ifdef RTMI_STAMPS // diagnostic build only (make stamps)
__device__ unsigned long long g_phase[3 * PH_SLOTS]; // [0..PH_SLOTS) wave ticks per phase, then lane-ticks, then stamps executed, summed over the launch's waves
#endif
// Where a lane's traversal stands: node = the node to visit next (inner node: byte offset of its record (>= 0); leaf: ~(primitive |
// moving << 30) (< 0); RTMI_BVH_EMPTY: done), tos = the newest stack entry (a register), top = next free slot of the thread's LDS column.
// The older entries live in the column; the bottom entry is a sentinel (RTMI_BVH_EMPTY, the initial tos), so popping needs no
// emptiness test and the LDS read of a pop is only needed by the NEXT pop or push: its latency is off the critical path.  (The
// tree's depth is < RTMI_BVH_STACK - 1 by construction; level 0 is only ever READ, by the pop of the sentinel.)
struct BvhCursor { int node, tos; int *top; };
__device__ inline BvhCursor bvh_cursor_at_root(SceneRef sc, int *stack) { return BvhCursor{sc.bvh_root, RTMI_BVH_EMPTY, stack + RTMI_BVH_STRIDE + threadIdx.x}; }

// SLICE: TIME-SLICED traversal.  A few rays of a wave visit ten times the nodes the others do (C2: 42 % of the descent trips and 40 % of
// the exact-test phases served fewer than 8 lanes, 20 % a single lane).  Two rules, one threshold: (1) the descent loop stops as soon
// as fewer than min_lanes lanes are still descending -- the lanes that wait with a leaf get their exact test now, the stragglers
// descend on in the next round, next to the lanes that have popped their next node; (2) the whole loop returns, with the cursors
// of the unfinished lanes, as soon as fewer than min_lanes lanes still have nodes to visit (checked after each exact-test phase,
// so every call makes progress): the caller lets the finished lanes move on and calls again with the same cursors and stack
// columns.  Per lane the visiting order is the same depth-first order either way.  min_lanes = 0: plain while-while.
template <bool NODE16, bool COUNT, bool SLICE, typename Leaf, typename BestHi>
__device__ inline void bvh_traverse_fmt(SceneRef sc, const BvhRay &r, BvhCursor &cur, int min_lanes, Leaf leaf, BestHi best, unsigned *cnt) {
    int node = cur.node;
    if (node == RTMI_BVH_EMPTY) return;
    constexpr int stride = RTMI_BVH_STRIDE;
    int *top = cur.top;
    int tos = cur.tos;
    const char *nodes = reinterpret_cast<const char *>(sc.bvh_nodes);
    float best_hi = best();
    while (node != RTMI_BVH_EMPTY) {
#ifdef RTMI_HIST
        { const int c = __popcll(__ballot(1)); if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_hist[128 + c], 1ull); }
#endif
        bool descend = node >= 0;
#ifdef RTMI_STAMPS
        RTMI_PH(PH_BVH_POST) // loop control between the phases
        unsigned st_trips = 0, st_lane_trips = 0;
#endif
#ifdef RTMI_MARKERS
        RTMI_PH(PH_BVH_POST)
#endif
        while (descend) { // inner node: both child boxes come with it (one record)
#ifdef RTMI_MARKERS
            asm volatile("; PHASE_BEGIN PH_DESCENT");
#endif
#ifdef RTMI_STAMPS
            st_trips += 1; st_lane_trips += (unsigned)__popcll(__ballot(1));
#endif
#ifdef RTMI_HIST
            { const int c = __popcll(__ballot(1)); if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_hist[c], 1ull); }
#endif
            if (COUNT) cnt[0] += 2; // two AABB slab tests (the reference counts one per AABB.hit?, hitable.clj:39)
            float tl, tr;
            bool hl, hr;
            int cl, cr;
            if (NODE16) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const uint4 *q = reinterpret_cast<const uint4 *>(nodes + (unsigned)node);
                const uint4 n0 = q[0], n1 = q[1]; // l.x l.y l.z r.x | r.y r.z left right   (each plane pair = {lo, hi} halves)
                const h2 lx = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(n0.x, n0.x, r.shx)), ly = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(n0.y, n0.y, r.shy)),
                         lz = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(n0.z, n0.z, r.shz)), rx = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(n0.w, n0.w, r.shx)),
                         ry = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(n1.x, n1.x, r.shy)), rz = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(n1.y, n1.y, r.shz));
                // hit  <=>  max(entry distances, t-min) <= min(exit distances, best t so far)
                tl = fmaxf(fmaxf(fmaf((float)lx.x, r.ixy.x, r.cex), fmaf((float)ly.x, r.ixy.y, r.cey)), fmaxf(fmaf((float)lz.x, r.izz.x, r.cez), r.tmin_lo));
                tr = fmaxf(fmaxf(fmaf((float)rx.x, r.ixy.x, r.cex), fmaf((float)ry.x, r.ixy.y, r.cey)), fmaxf(fmaf((float)rz.x, r.izz.x, r.cez), r.tmin_lo));
                // (min_raw: best_hi is loop-carried, so the compiler would re-quiet it with a v_max x, x before every fminf -- it is never a NaN)
                const float fl = fminf(fminf(fmaf((float)lx.y, r.ixy.x, r.cxx), fmaf((float)ly.y, r.ixy.y, r.cxy)), min_raw(fmaf((float)lz.y, r.izz.x, r.cxz), best_hi));
                const float fr = fminf(fminf(fmaf((float)rx.y, r.ixy.x, r.cxx), fmaf((float)ry.y, r.ixy.y, r.cxy)), min_raw(fmaf((float)rz.y, r.izz.x, r.cxz), best_hi));
                hl = tl <= fl; hr = tr <= fr;
                cl = (int)n1.z; cr = (int)n1.w;
            } else { // 64-byte record: l.lo.xy l.hi.xy | r.lo.xy r.hi.xy | l.lo.z l.hi.z r.lo.z r.hi.z | left, right
                const float4 *q = reinterpret_cast<const float4 *>(nodes + (unsigned)node);
                const float4 n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[3];
                hl = slab_hit(__builtin_elementwise_fma(v2f{n0.x, n0.y}, r.ixy, r.clxy), __builtin_elementwise_fma(v2f{n0.z, n0.w}, r.ixy, r.chxy),
                              __builtin_elementwise_fma(v2f{n2.x, n2.y}, r.izz, r.czz), r.tmin_lo, best_hi, tl);
                hr = slab_hit(__builtin_elementwise_fma(v2f{n1.x, n1.y}, r.ixy, r.clxy), __builtin_elementwise_fma(v2f{n1.z, n1.w}, r.ixy, r.chxy),
                              __builtin_elementwise_fma(v2f{n2.z, n2.w}, r.izz, r.czz), r.tmin_lo, best_hi, tr);
                cl = __float_as_int(n3.x); cr = __float_as_int(n3.y);
            }
            if (hl && hr) { // push the far child, descend into the near one
                const bool left_first = tl <= tr;
                *top = tos;
                top += stride;
                tos = left_first ? cr : cl;
                node = left_first ? cl : cr;
            } else if (hl) node = cl;
            else if (hr) node = cr;
            else { node = tos; top -= stride; tos = *top; } // pop (the sentinel ends the traversal)
            descend = node >= 0;
            // time-sliced: the few lanes on long descents stop holding up the lanes that wait with a leaf (one exit condition per lane:
            // a wave-uniform `break` costs the structured loop a dozen scalar instructions per trip)
#if RTMI_EXIT_IBALLOT
            if (SLICE) { // the lane mask of the descending lanes IS the loop's next exec mask: no second compare to turn it back into a per-lane flag
                const unsigned long long dm = __ballot(descend);
                descend = __builtin_amdgcn_inverse_ballot_w64(__popcll(dm) >= min_lanes ? dm : 0ull);
            }
#else
            if (SLICE) descend = descend & (__popcll(__ballot(descend)) >= min_lanes);
#endif
#ifdef RTMI_STAMPS
            if (__ballot(descend) == 0) { RTMI_PH_LANES(PH_DESCENT, st_lane_trips, st_trips) } // the wave's last trip: these lanes were in every trip
#endif
#ifdef RTMI_MARKERS
            RTMI_PH(PH_DESCENT)
#endif
        }
        if (node < 0 && node != RTMI_BVH_EMPTY) { // leaf: one primitive, exact FP64 test
#ifdef RTMI_HIST
            { const int c = __popcll(__ballot(1)); if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_hist[64 + c], 1ull); }
#endif
            if (COUNT) cnt[1] += 1;
            leaf(node);
            best_hi = best();
            node = tos; top -= stride; tos = *top;
            RTMI_PH(PH_LEAF)
        }
        if (SLICE && __popcll(__ballot(node != RTMI_BVH_EMPTY)) < min_lanes) break; // wave-uniform
    }
    cur.node = node; cur.tos = tos; cur.top = top;
}

// Entry grid (DevScene::grid_*): where does this ray's traversal START?  Half of the node visits of a ray from the root only locate it -- the
// boxes of the top ten levels nearly all contain a ray that starts inside the scene.  The layer of primitives is cut into x-z cells with a
// BVH each; the ray's parameter range inside the layer's box (the slab test of one more box, with the constants of make_bvh_ray, cut at the
// closest hit so far -- the ground, usually) gives a segment, the segment's end points a rectangle of cells.  If that rectangle is at most
// 2 x 2 cells and grid_kmax cells, the traversal starts with those cells' roots and -- if the ray meets their box -- the tall primitives' tree on
// the stack; otherwise at the root of the whole tree.  Every primitive the ray can hit within its range overlaps one of those cells or is tall:
// the end points carry a float error below 2^-21 (|o| + cbound) and the rectangle is grown by 2^-16 of that; cells claim primitives by
// their inflated boxes.  A primitive two cells share is tested twice at worst (same t, same index: the any-order rule keeps one).
// e_rel: rectangle growth relative to |o| + cbound -- RTMI_F32's float sphere test calls a hit up to 3.1e-3 (|o| + cbound) outside the sphere
// (make_bvh_ray), so its rectangle grows by that much more.
// Long segments (RTMI_GRID_CHUNK): a segment whose rectangle is larger is cut at t_split, the parameter up to which it stays within the 2 x 2 cells
// around its start (1 .. 2 cells of travel along the tighter axis), and only [start, t_split] is entered now -- every primitive it can hit at a
// t <= t_split overlaps those cells.  The caller traverses them, and if the closest hit so far is not at or before t_split it calls again with
// t_from = t_split (first = false: the tall primitives' tree, entered with the first piece, is done): an ORDERED walk, nearest piece first, that
// ends at the first piece holding the hit -- the mean free path of a ray inside the layer is about a cell --, where the descent from the root of
// the whole tree spends a dozen node visits on locating the ray.  Returns t_split, +inf when this start covers all that remains of the segment.
__device__ inline float bvh_grid_entry(SceneRef sc, const BvhRay &r, float ox, float oy, float oz, float dx, float dy, float dz, float best_hi, float e_rel, BvhCursor &cur,
                                       float t_from = -__builtin_inff(), bool first = true) {
    const float kNoSplit = __builtin_inff();
    float t_split = kNoSplit;
    const int G = sc.grid_n;
    // The slab test of a node visit: per axis the {lo, hi} half pair is rotated by the ray's direction sign (entry plane first) and v_fma_mix_f32 gives
    // entry and exit distances with the ray's own conservative constants -- 3 + 6 instructions per box, where selecting float planes by sign took 18.
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    auto box_range = [&](const unsigned w0, const unsigned w1, const unsigned w2, float far_hi, float &tn_, float &tf_) {
        const h2 hx = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(w0, w0, r.shx)), hy = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(w1, w1, r.shy)),
                 hz = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(w2, w2, r.shz));
        tn_ = fmaxf(fmaxf(fmaf((float)hx.x, r.ixy.x, r.cex), fmaf((float)hy.x, r.ixy.y, r.cey)), fmaxf(fmaf((float)hz.x, r.izz.x, r.cez), r.tmin_lo));
        tf_ = fminf(fminf(fmaf((float)hx.y, r.ixy.x, r.cxx), fmaf((float)hy.y, r.ixy.y, r.cxy)), fminf(fmaf((float)hz.y, r.izz.x, r.cxz), far_hi));
    };
    float tn, tf;
    box_range(sc.grid_box_h[0], sc.grid_box_h[1], sc.grid_box_h[2], best_hi, tn, tf);
    tn = fmaxf(tn, t_from);
    int tall = first ? sc.grid_tall : RTMI_BVH_EMPTY;
    if (tall != RTMI_BVH_EMPTY) { // (wave-uniform) the tall primitives' tree: only if the ray meets their box within its range
        float un, uf;
        box_range(sc.grid_tall_box_h[0], sc.grid_tall_box_h[1], sc.grid_tall_box_h[2], best_hi, un, uf);
        tall = un > uf ? RTMI_BVH_EMPTY : tall; // (a NaN keeps the tree)
    }
    constexpr int stride = RTMI_BVH_STRIDE;
    int node = RTMI_BVH_EMPTY, tos = cur.tos, *top = cur.top;
    if (tn <= tf) {
        const float omax = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
        const float e = (omax + sc.bvh_cbound) * e_rel;
        const float p0x = fmaf(tn, dx, ox), p0z = fmaf(tn, dz, oz);
        float p1x = fmaf(tf, dx, ox), p1z = fmaf(tf, dz, oz);
        const float gmax = (float)(G - 1);
        // cell coordinate = (p - lo) * inv as one fma (p * inv + (-lo * inv)); its rounding (a few 2^-24 of a coordinate <= grid_n) is far below
        // the margin e * inv, which is at least 2^-16 (|o| + cbound) * inv >= 2^-16 of the layer's extent in cells
        const float ax = sc.grid_inv_x, az = sc.grid_inv_z, bx = -sc.grid_lo_x * ax, bz = -sc.grid_lo_z * az, ex = e * ax, ez = e * az;
        const float sx = fmaf(p0x, ax, bx), sz = fmaf(p0z, az, bz); // where the range starts, in cells
        int i0, i1, j0, j1, wi, wj;
        auto rect = [&]() { // the cells the rectangle of the range's two end points touches
            const float qx = fmaf(p1x, ax, bx), qz = fmaf(p1z, az, bz);
            const float fx0 = fminf(sx, qx) - ex, fx1 = fmaxf(sx, qx) + ex, fz0 = fminf(sz, qz) - ez, fz1 = fmaxf(sz, qz) + ez;
            i0 = (int)__builtin_amdgcn_fmed3f(floorf(fx0), 0.0f, gmax); i1 = (int)__builtin_amdgcn_fmed3f(floorf(fx1), 0.0f, gmax); // (med3 of a NaN: the smaller bound)
            j0 = (int)__builtin_amdgcn_fmed3f(floorf(fz0), 0.0f, gmax); j1 = (int)__builtin_amdgcn_fmed3f(floorf(fz1), 0.0f, gmax);
            wi = i1 - i0; wj = j1 - j0;
            return wi <= 1 && wj <= 1 && (wi + 1) * (wj + 1) <= sc.grid_kmax; // (a NaN: false)
        };
        if (!rect()) { // too many cells
#if RTMI_GRID_CHUNK
            // the first piece: as far as the range stays inside the two cells per axis that hold its start (and the start's own margin), less 2^-8 of
            // a cell for the rounding of this very computation -- the rectangle of the piece is then taken from its end points like any other
            const float gx = dx * ax, gz = dz * az; // cells per unit of t
            const float lx = gx >= 0.0f ? __builtin_amdgcn_fmed3f(floorf(sx - ex), 0.0f, gmax) + (2.0f - 1.0f / 256.0f) - ex : __builtin_amdgcn_fmed3f(floorf(sx + ex), 0.0f, gmax) - (1.0f - 1.0f / 256.0f) + ex;
            const float lz = gz >= 0.0f ? __builtin_amdgcn_fmed3f(floorf(sz - ez), 0.0f, gmax) + (2.0f - 1.0f / 256.0f) - ez : __builtin_amdgcn_fmed3f(floorf(sz + ez), 0.0f, gmax) - (1.0f - 1.0f / 256.0f) + ez;
            const float ts = tn + fminf((lx - sx) * __builtin_amdgcn_rcpf(gx), (lz - sz) * __builtin_amdgcn_rcpf(gz));
            p1x = fmaf(ts, dx, ox); p1z = fmaf(ts, dz, oz);
            if (!(ts > tn && ts < tf && sc.grid_walk && rect())) return kNoSplit; // (cannot happen for margins far below a cell; if it does: the whole tree, from its root)
            t_split = ts;
#else
            return kNoSplit; // the whole tree, from its root
#endif
        }
        // ONE root: the host built a tree for every rectangle of cells (1 x 1, 2 x 1, 1 x 2, 2 x 2: family wi + 2 wj, indexed by the low corner) over the union
        // of the cells' primitives -- nothing to push, no root per cell to visit, and a primitive two of the cells share is tested once
        node = sc.grid_cells[((wi + 2 * wj) * G + j0) * G + i0];
    } else if (!(tn > tf)) return kNoSplit; // (a NaN cannot arise for a ray make_bvh_ray accepted; if it did: the whole tree)
    // else: the ray does not meet the layer's box within its range -- only the tall primitives remain
    if (tall != RTMI_BVH_EMPTY) { // (few lanes: the tall primitives' box is small) visited first
        if (node != RTMI_BVH_EMPTY) { *top = tos; top += stride; tos = node; }
        node = tall;
    }
    cur.node = node; cur.tos = tos; cur.top = top;
    return t_split;
}

// one loop per record format (a format test inside the loop costs 5 %)
template <bool COUNT = false, bool SLICE = false, typename Leaf, typename BestHi>
__device__ inline void bvh_traverse(SceneRef sc, const BvhRay &r, BvhCursor &cur, int min_lanes, Leaf leaf, BestHi best, unsigned *cnt = nullptr) {
    if (sc.bvh_node16) bvh_traverse_fmt<true, COUNT, SLICE>(sc, r, cur, min_lanes, leaf, best, cnt);
    else bvh_traverse_fmt<false, COUNT, SLICE>(sc, r, cur, min_lanes, leaf, best, cnt);
}

// what a suspended lane keeps between two calls of scan_bvh (time-sliced traversal): LDS, one column per thread like the stack
#define RTMI_BVH_SUSPEND_WORDS 6 // node, tos, top (byte offset into the stack), best_t (2 words), best_i
template <typename R> __device__ inline void best_to_words(R t, int &w0, int &w1);
template <> __device__ inline void best_to_words<double>(double t, int &w0, int &w1) { w0 = __double2loint(t); w1 = __double2hiint(t); }
template <> __device__ inline void best_to_words<float>(float t, int &w0, int &w1) { w0 = __float_as_int(t); w1 = 0; }
template <typename R> __device__ inline R best_from_words(int w0, int w1);
template <> __device__ inline double best_from_words<double>(int w0, int w1) { return __hiloint2double(w1, w0); }
template <> __device__ inline float best_from_words<float>(int w0, int) { return __int_as_float(w0); }

// flat(): the exact flat scan of the same precision, for rays the float boxes cannot bound
// COUNT (diagnostic instantiation, RTMI option "count_traversal"): cnt[0] += AABB slab tests, cnt[1] += exact primitive tests
// Time-sliced use (susp != nullptr; min_lanes as in bvh_traverse_fmt): returns false when the lane's traversal was suspended -- its
// cursor and closest hit so far are then in susp (RTMI_BVH_SUSPEND_WORDS columns of blockDim.x words) -- and the caller passes
// resume = true on the next call with the SAME ray; best_t / best_i are only final when the function returns true.
// CHUNK = false: the instantiation for scenes WITHOUT an entry grid (the plain render kernel: it is never launched on a scene that has one)
template <typename R, bool COUNT = false, bool SLICE = false, bool CHUNK = true, typename Flat>
__device__ inline bool scan_bvh(SceneRef sc, int *stack, const Path<R> &P, R a, R tmin, R &best_t, int &best_i, Flat flat, unsigned *cnt = nullptr,
                                int *susp = nullptr, bool resume = false, int min_lanes = 0) {
    constexpr bool WALK = CHUNK && RTMI_GRID_CHUNK;
    const bool behind_ok = tmin >= R(0);
    const double *exact12 = sc.exact12;
    const BvhRay r = make_bvh_ray<R>(sc, P, a, tmin); // a function of the ray: a resumed lane gets the values it had
    constexpr int stride = RTMI_BVH_STRIDE;
    int *sw = susp + threadIdx.x;
    BvhCursor cur;
    // RTMI_GRID_CHUNK: a long segment is walked piece by piece (bvh_grid_entry).  t_split = where the piece being traversed ends (+inf: it is the
    // last one); it rides in a register while the lane is in the tree and in level 0 of the lane's stack column (a slot the traversal only ever
    // reads, for the pop of the sentinel) while the lane is parked.  A lane whose piece is finished without the hit being settled parks with
    // `top` = -1 and resumes with the next piece: the grid entry then runs at full width next to the new segments of the other lanes.
    float t_split = __builtin_inff(), t_from = -__builtin_inff();
    bool reenter = false;
    if (SLICE && resume) {
        cur.node = sw[0]; cur.tos = sw[stride];
        const int top_off = sw[2 * stride];
        cur.top = reinterpret_cast<int *>(reinterpret_cast<char *>(stack) + top_off);
        best_t = best_from_words<R>(sw[3 * stride], sw[4 * stride]);
        best_i = sw[5 * stride];
        if (WALK) {
            t_split = __int_as_float(stack[threadIdx.x]);
            if (top_off < 0) { reenter = true; t_from = t_split; t_split = __builtin_inff(); cur = bvh_cursor_at_root(sc, stack); }
        }
    } else {
        if (!r.ok) { // rays the float traversal cannot bound take the exact flat scan (all primitives, original order) instead
            if (COUNT) cnt[1] += (unsigned)sc.n_all;
            flat();
            return true;
        }
        cur = bvh_cursor_at_root(sc, stack);
    }
    const Quot<R> qa = make_quot<R>(a, tmin, best_t); // every root of this ray divides by a
    RTMI_PH(PH_BVH_SETUP)
    if (!(SLICE && resume)) { // 1. the big primitives (sky dome, ground, ...): exact test, ascending Hitlist index
        if (COUNT) cnt[1] += (unsigned)sc.n_big;
        for (int k = 0; k < sc.n_big; ++k) exact_prim_test<R>(exact12, sc.big_idx[k], sc.big_idx[k], P, qa, tmin, behind_ok, best_t, best_i);
        RTMI_PH(PH_BIG)
    }
    // 2. the tree
    auto leaf = [&](int code) { exact_prim_test_lane<R>(exact12, code, P, qa, tmin, behind_ok, best_t, best_i); };
    auto best = [&]() { return float_above((double)best_t); };
    const float e_rel = sizeof(R) == sizeof(float) ? 4.0e-3f + 1.0f / 65536.0f : 1.0f / 65536.0f;
    bool enter = CHUNK && sc.grid_n && (!(SLICE && resume) || reenter), first = !reenter;
    for (;;) {
        if (enter) // where the traversal starts: the roots of the grid cells the ray's segment (or its next piece) crosses, or the root of the whole tree
        {
            t_split = bvh_grid_entry(sc, r, (float)P.ox, (float)P.oy, (float)P.oz, (float)P.dx, (float)P.dy, (float)P.dz, best(), e_rel, cur, t_from, first);
            RTMI_PH(PH_GRID)
        }
        bvh_traverse<COUNT, SLICE>(sc, r, cur, min_lanes, leaf, best, cnt);
        if (SLICE && cur.node != RTMI_BVH_EMPTY) { // suspended
            int w0, w1;
            best_to_words<R>(best_t, w0, w1);
            sw[0] = cur.node; sw[stride] = cur.tos;
            sw[2 * stride] = (int)(reinterpret_cast<char *>(cur.top) - reinterpret_cast<char *>(stack));
            sw[3 * stride] = w0; sw[4 * stride] = w1; sw[5 * stride] = best_i;
            if (WALK) stack[threadIdx.x] = __float_as_int(t_split);
            return false;
        }
        if (!WALK) break;
        if (!(best_t > (R)t_split)) break; // the hit is settled: whatever the later pieces hold lies beyond t_split (always, when there is no later piece: +inf)
        if (SLICE) { // the next piece is entered next trip, at full width
            int w0, w1;
            best_to_words<R>(best_t, w0, w1);
            sw[0] = RTMI_BVH_EMPTY; sw[stride] = RTMI_BVH_EMPTY; sw[2 * stride] = -1;
            sw[3 * stride] = w0; sw[4 * stride] = w1; sw[5 * stride] = best_i;
            stack[threadIdx.x] = __float_as_int(t_split);
            return false;
        }
        cur = bvh_cursor_at_root(sc, stack);
        t_from = t_split; first = false; enter = true; // (probes: the next piece right away)
    }
    // 3. a ray outside the shutter interval: the MovingSphere boxes were built for [t_lo, t_hi], so test every moving sphere exactly
    if (!r.time_ok)
        for (int k = 0; k < sc.n_moving_all; ++k) exact_prim_test_lane<R>(exact12, ~(sc.moving_all[k] | 0x40000000), P, qa, tmin, behind_ok, best_t, best_i);
    return true;
}

// ==== section 8(f3): RectXY/XZ/YZ (hitable.clj:269-363), Triangle (548-571), FlipNormals (375-381), Translate (391-396),
// ==== RotateY (410-450), Box = Hitlist of six (flipped) rectangles (491-511) -- FP64 only ==================================
// Closest hit over MIXED kinds, any visiting order.  In the reference's sequential Hitlist scan (running t-max = closest
// so far) a sphere replaces the current hit only if t < closest (hitable.clj:195) while rectangles and triangles use
// t <= closest (hitable.clj:278, 564): among primitives tied at the minimal t the winner is the LAST "inclusive" one after
// the first tied primitive, else the first.  F = lowest tied index, W = highest tied inclusive index: winner = max(F, W).
struct ExtHit { double t; int F, W; __device__ inline bool any() const { return F != 0x7fffffff; } }; // F = 0x7fffffff: no hit yet, t = the caller's t-max
__device__ inline int ext_winner(const ExtHit &H) { return H.any() ? max(H.F, H.W) : -1; }
// Branch-free (selects under lane masks): as nested `if`s the candidate tests of a mixed-kind scan compiled to three to five levels of divergent control flow per
// primitive, each level with copies of the whole state at its join (a rectangle test: 13 arithmetic instructions among ~40 moves and ~45 scalar mask operations).
// ok = false: the candidate is no hit at all (t outside the interval, the point outside the rectangle): nothing changes.
__device__ inline void ext_update(ExtHit &H, double t, int idx, bool incl, bool ok = true) {
    const bool clt = t < H.t, ceq = t == H.t; // (H.t is the caller's t-max until the first hit)
    const bool any = H.any();
    const bool take = ok & (clt | (incl & ceq & !any));
    const bool tie = ok & ceq & any; // (never together with `take`)
    H.F = take ? idx : H.F;
    H.W = take ? (incl ? idx : -1) : H.W;
    H.t = take ? t : H.t;
    if (__builtin_expect(__any(tie), 0)) { // an exact tie in some lane of the wave: one scalar test in front of five vector instructions that almost never have work
        H.F = tie ? min(H.F, idx) : H.F;
        if (incl) H.W = tie ? max(H.W, idx) : H.W;
    }
}

struct LocalRay { double ox, oy, oz, dx, dy, dz; };
// Scene arrays are read-only for the whole launch.  UNIFORM = the index is the same for every lane of the wave (the big
// primitives, the flat scan, a medium's boundary list): read through the constant address space, i.e. the scalar data cache
// into SGPRs, and every kind / chain branch is a scalar branch.  Per-lane indices (BVH leaves, the winner) use vector loads.
template <bool UNIFORM> __device__ inline double ext_ld(const double *p, size_t i) {
    if (UNIFORM) return reinterpret_cast<const __attribute__((address_space(4))) double *>((const __attribute__((address_space(4))) void *)(p))[i];
    return p[i];
}
template <bool UNIFORM> __device__ inline int4 ext_ld_info(const int *p, int idx) {
    if (UNIFORM) {
        typedef int i4 __attribute__((ext_vector_type(4)));
        const i4 v = reinterpret_cast<const __attribute__((address_space(4))) i4 *>((const __attribute__((address_space(4))) void *)(p))[idx];
        return make_int4(v.x, v.y, v.z, v.w);
    }
    return reinterpret_cast<const int4 *>(p)[idx];
}
// the ray as the innermost record sees it: Translate subtracts its offset from the origin (hitable.clj:394), RotateY
// pre-rotates origin and direction (hitable.clj:423-429); outermost wrapper first
template <bool UNIFORM = false>
__device__ inline LocalRay ext_local_ray(SceneRef sc, int first, int count, const Path<double> &P) {
    LocalRay r = {P.ox, P.oy, P.oz, P.dx, P.dy, P.dz};
    for (int k = 0; k < count; ++k) {
        const size_t q = (size_t)(first + k) * 4;
        const double q0 = ext_ld<UNIFORM>(sc.ext_xf, q), q1 = ext_ld<UNIFORM>(sc.ext_xf, q + 1), q2 = ext_ld<UNIFORM>(sc.ext_xf, q + 2);
        if (q0 == 0.0) { const double q3 = ext_ld<UNIFORM>(sc.ext_xf, q + 3); r.ox = r.ox - q1; r.oy = r.oy - q2; r.oz = r.oz - q3; }
        else {
            const double sn = q1, cs = q2;
            const double ox = cs * r.ox - sn * r.oz, oz = sn * r.ox + cs * r.oz;
            const double dx = cs * r.dx - sn * r.dz, dz = sn * r.dx + cs * r.dz;
            r.ox = ox; r.oz = oz; r.dx = dx; r.dz = dz;
        }
    }
    return r;
}
__device__ inline void cross3(double ax, double ay, double az, double bx, double by, double bz, double &x, double &y, double &z) {
    x = ay * bz - az * by; y = az * bx - ax * bz; z = ax * by - ay * bx;
}
// Moeller-Trumbore exactly as hitable.clj:551-563 (one sided: det > 1e-8); returns false when there is no candidate
__device__ inline bool tri_mt(double g0, double g1, double g2, double g3, double g4, double g5, double g6, double g7, double g8, const LocalRay &r,
                              double &u, double &v, double &t) {
    const double e1x = g3 - g0, e1y = g4 - g1, e1z = g5 - g2;
    const double e2x = g6 - g0, e2y = g7 - g1, e2z = g8 - g2;
    double px, py, pz;
    cross3(r.dx, r.dy, r.dz, e2x, e2y, e2z, px, py, pz);
    const double det = dot3(e1x, e1y, e1z, px, py, pz);
    if (!(det > 0.00000001)) return false;
    const double inv_det = 1.0 / det;
    const double tx = r.ox - g0, ty = r.oy - g1, tz = r.oz - g2;
    u = dot3(tx, ty, tz, px, py, pz) * inv_det;
    if (!(u > 0.0 && u <= 1.0)) return false;
    double qx, qy, qz;
    cross3(tx, ty, tz, e1x, e1y, e1z, qx, qy, qz);
    v = dot3(r.dx, r.dy, r.dz, qx, qy, qz) * inv_det;
    if (!(v > 0.0 && u + v <= 1.0)) return false;
    t = dot3(e2x, e2y, e2z, qx, qy, qz) * inv_det;
    return true;
}
__device__ inline double pick3(int k, double x, double y, double z) { return k == 0 ? x : (k == 1 ? y : z); } // no runtime-indexed arrays: they go to scratch
__device__ inline void rect_axes(int kind, int &ax, int &ua, int &va) {
    ax = kind == RTMI_PRIM_RECT_XY ? 2 : (kind == RTMI_PRIM_RECT_XZ ? 1 : 0);
    ua = kind == RTMI_PRIM_RECT_YZ ? 1 : 0;
    va = kind == RTMI_PRIM_RECT_XY ? 1 : 2;
}

// ---- Box (hitable.clj:491-511) as ONE leaf --------------------------------------------------------------------------------------------
// (box :p0 :p1) is a Hitlist of six rectangles -- RectXY k = z1, (flipped) RectXY k = z0, RectXZ k = y1, (flipped) RectXZ k = y0, RectYZ k = x1,
// (flipped) RectYZ k = x0 -- that the flattener splices into the world as six consecutive primitives with one instance chain.  The host recognises
// such a run (scene creation: kinds, chain and all thirty parameters must agree) and hands the tree ONE leaf for it (leaf-code bit RTMI_LEAF_BOX,
// index = the first rectangle; z0 rides in the otherwise unused slot 5 of that rectangle's record): a ray that meets the box pays one exact-test
// phase and no node visits among the faces, where six flat leaves cost it three to five visits and two to four phases.  The six tests below are
// the six RectXY/XZ/YZ.hit? bodies (hitable.clj:269-363: t = (k - o_a) / d_a, t-min <= t, the point inside the rectangle, bounds inclusive), each
// folded into the any-order state under ITS OWN primitive index -- exactly what six calls of ext_prim_test do --, evaluated on the chain's local ray
// computed once instead of six times.
//
// The three divisors are the local direction's components, two faces each: RefinedRcp keeps the refined reciprocal of the IEEE division sequence
// (Quot's argument, rtmi_device.h above; signed divisors: every step of the sequence is odd in a) and a face's t costs the sequence's last three
// operations.  Same bits whenever the division would not have rescaled (2^-200 <= |d_a| <= 2^200, 2^-969 <= |k - o_a| < 2^568); outside the
// numerator's range |t| < 2^-768 or >= 2^368 or NaN with either form, which fails t-min <= t or t <= closest-so-far alike for t-min >= 2^-300
// and a closest hit <= 2^200 (checked per lane: `fast`).
#define RTMI_LEAF_BOX 0x20000000
struct RefinedRcp { double a, r; };
__device__ inline RefinedRcp refined_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = ::fma(r, ::fma(-a, r, 1.0), r);
    return RefinedRcp{a, ::fma(r, ::fma(-a, r, 1.0), r)};
}
__device__ inline bool rcp_in_range(double a) { return ((unsigned)__double2hiint(a) & 0x7fffffffu) - (823u << 20) < (400u << 20); } // biased exponent in [823, 1223)
__device__ inline double div_by(double n, const RefinedRcp &d, bool fast) {
    if (__builtin_expect(!fast, 0)) return n / d.a;
    const double q = n * d.r;
    return ::fma(::fma(-d.a, q, n), d.r, q);
}
// one face: the rectangle in the plane x_a = k, in-plane coordinates (u, v) within [u0, u1] x [v0, v1]
// fast_all (wave-uniform): every lane of the wave is `fast` -- one scalar branch instead of a lane mask around the plain division
__device__ inline void box_face(double k, double oa, const RefinedRcp &da, bool fast, double ou, double du, double ov, double dv, double u0, double u1, double v0, double v1,
                                double tmin, int idx, ExtHit &H, bool fast_all = false) {
    double t;
    if (fast_all) { const double n = k - oa, q = n * da.r; t = ::fma(::fma(-da.a, q, n), da.r, q); }
    else t = div_by(k - oa, da, fast);
    const double x = ou + t * du, y = ov + t * dv; // (for every lane: a t below t-min or not a number fails the first comparison)
    ext_update(H, t, idx, true, (t >= tmin) & (x >= u0) & (x <= u1) & (y >= v0) & (y <= v1));
}
// r: the ray in the box's frame (ext_local_ray of the six rectangles' common chain); rec: the first rectangle's record (x0 y0 x1 y1 z1 z0)
__device__ inline void ext_box_faces(const LocalRay &r, double x0, double y0, double x1, double y1, double z1, double z0, double tmin, int idx, ExtHit &H) {
    const bool fast = rcp_in_range(r.dx) && rcp_in_range(r.dy) && rcp_in_range(r.dz) && tmin >= 0x1p-300 && H.t <= 0x1p200;
    const RefinedRcp qx = refined_rcp(r.dx), qy = refined_rcp(r.dy), qz = refined_rcp(r.dz);
    box_face(z1, r.oz, qz, fast, r.ox, r.dx, r.oy, r.dy, x0, x1, y0, y1, tmin, idx, H);     // RectXY x0 y0 x1 y1 z1
    box_face(z0, r.oz, qz, fast, r.ox, r.dx, r.oy, r.dy, x0, x1, y0, y1, tmin, idx + 1, H); // RectXY x0 y0 x1 y1 z0
    box_face(y1, r.oy, qy, fast, r.ox, r.dx, r.oz, r.dz, x0, x1, z0, z1, tmin, idx + 2, H); // RectXZ x0 z0 x1 z1 y1
    box_face(y0, r.oy, qy, fast, r.ox, r.dx, r.oz, r.dz, x0, x1, z0, z1, tmin, idx + 3, H); // RectXZ x0 z0 x1 z1 y0
    box_face(x1, r.ox, qx, fast, r.oy, r.dy, r.oz, r.dz, y0, y1, z0, z1, tmin, idx + 4, H); // RectYZ y0 z0 y1 z1 x1
    box_face(x0, r.ox, qx, fast, r.oy, r.dy, r.oz, r.dz, y0, y1, z0, z1, tmin, idx + 5, H); // RectYZ y0 z0 y1 z1 x0
}
template <bool UNIFORM = false>
__device__ inline void ext_box_test(SceneRef sc, int idx, const Path<double> &P, double tmin, ExtHit &H) {
    const int4 info = ext_ld_info<UNIFORM>(sc.ext_info, idx);
    const LocalRay r = ext_local_ray<UNIFORM>(sc, info.z, info.w, P);
    const size_t gi = (size_t)idx * 12;
    if (UNIFORM) {
        ext_box_faces(r, ext_ld<true>(sc.exact12, gi), ext_ld<true>(sc.exact12, gi + 1), ext_ld<true>(sc.exact12, gi + 2), ext_ld<true>(sc.exact12, gi + 3),
                      ext_ld<true>(sc.exact12, gi + 4), ext_ld<true>(sc.exact12, gi + 5), tmin, idx, H);
    } else {
        const double2 *g = reinterpret_cast<const double2 *>(sc.exact12 + gi);
        const double2 g0 = g[0], g1 = g[1], g2 = g[2];
        ext_box_faces(r, g0.x, g0.y, g1.x, g1.y, g2.x, g2.y, tmin, idx, H);
    }
}

// hit? of primitive idx (any kind) on the ray r already taken through its instance chain, folded into the any-order state.
// q (optional): the refined reciprocals of r's direction components (ext_box_faces) -- a rectangle's t = (k - o_a) / d_a then costs the division's last
// three operations (`fast`, per lane: see RefinedRcp); nullptr: the plain division.
template <bool UNIFORM = false>
__device__ inline void ext_prim_test_local(SceneRef sc, int idx, int kind, const LocalRay &r, double time, double tmin, ExtHit &H, const RefinedRcp *q = nullptr, bool fast = false,
                                           bool fast_all = false) {
    const size_t gi = (size_t)idx * 12;
    const double g0 = ext_ld<UNIFORM>(sc.exact12, gi), g1 = ext_ld<UNIFORM>(sc.exact12, gi + 1), g2 = ext_ld<UNIFORM>(sc.exact12, gi + 2),
                 g3 = ext_ld<UNIFORM>(sc.exact12, gi + 3);
    if (kind <= RTMI_PRIM_MOVING) {
        Prim4<double> s;
        s.cx = g0; s.cy = g1; s.cz = g2; s.r2 = g3;
        if (kind == RTMI_PRIM_MOVING) {
            const double g4 = ext_ld<UNIFORM>(sc.exact12, gi + 4), g5 = ext_ld<UNIFORM>(sc.exact12, gi + 5), g6 = ext_ld<UNIFORM>(sc.exact12, gi + 6);
            const double t0 = ext_ld<UNIFORM>(sc.exact12, gi + 7), t1 = ext_ld<UNIFORM>(sc.exact12, gi + 8);
            const double f = (time - t0) / (t1 - t0), omf = 1.0 - f;
            s.cx = g0 * omf + g4 * f; s.cy = g1 * omf + g5 * f; s.cz = g2 * omf + g6 * f;
        }
        Path<double> L; L.ox = r.ox; L.oy = r.oy; L.oz = r.oz; L.dx = r.dx; L.dy = r.dy; L.dz = r.dz;
        const double a = dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz);
        double bq, cq, disc;
        sphere_test(s, L, a, bq, cq, disc);
        if (disc >= 0.0 && !(tmin >= 0.0 && bq > 0.0 && cq > 0.0)) {
            const double sq = rt_sqrt(disc);
            double t = (-bq - sq) / a;
            if (!(t > tmin)) t = (-bq + sq) / a;
            ext_update(H, t, idx, false, t > tmin);
        }
    } else if (kind <= RTMI_PRIM_RECT_YZ) {
        const double gk = ext_ld<UNIFORM>(sc.exact12, gi + 4);
        // the rectangle's plane axis a and its two in-plane axes u, v
        auto rect = [&](double oa, double da, double ou, double du, double ov, double dv) {
            const double t = (gk - oa) / da;
            const double x = ou + t * du, y = ov + t * dv;
            ext_update(H, t, idx, true, (t >= tmin) & (x >= g0) & (x <= g2) & (y >= g1) & (y <= g3));
        };
        if (UNIFORM) { // scalar branches on the kind, one copy of the test per kind: no selects, no indexed temporaries
            if (q) { // g = (u0 v0 u1 v1)
                if (kind == RTMI_PRIM_RECT_XY) box_face(gk, r.oz, q[2], fast, r.ox, r.dx, r.oy, r.dy, g0, g2, g1, g3, tmin, idx, H, fast_all);
                else if (kind == RTMI_PRIM_RECT_XZ) box_face(gk, r.oy, q[1], fast, r.ox, r.dx, r.oz, r.dz, g0, g2, g1, g3, tmin, idx, H, fast_all);
                else box_face(gk, r.ox, q[0], fast, r.oy, r.dy, r.oz, r.dz, g0, g2, g1, g3, tmin, idx, H, fast_all);
            }
            else if (kind == RTMI_PRIM_RECT_XY) rect(r.oz, r.dz, r.ox, r.dx, r.oy, r.dy);
            else if (kind == RTMI_PRIM_RECT_XZ) rect(r.oy, r.dy, r.ox, r.dx, r.oz, r.dz);
            else rect(r.ox, r.dx, r.oy, r.dy, r.oz, r.dz);
        } else {
            int ax, ua, va;
            rect_axes(kind, ax, ua, va);
            rect(pick3(ax, r.ox, r.oy, r.oz), pick3(ax, r.dx, r.dy, r.dz), pick3(ua, r.ox, r.oy, r.oz), pick3(ua, r.dx, r.dy, r.dz),
                 pick3(va, r.ox, r.oy, r.oz), pick3(va, r.dx, r.dy, r.dz));
        }
    } else {
        double u, v, t;
        if (tri_mt(g0, g1, g2, g3, ext_ld<UNIFORM>(sc.exact12, gi + 4), ext_ld<UNIFORM>(sc.exact12, gi + 5), ext_ld<UNIFORM>(sc.exact12, gi + 6),
                   ext_ld<UNIFORM>(sc.exact12, gi + 7), ext_ld<UNIFORM>(sc.exact12, gi + 8), r, u, v, t)) ext_update(H, t, idx, true, t >= tmin);
    }
}
// hit? of primitive idx (any kind, through its instance chain) folded into the any-order state
template <bool UNIFORM = false>
__device__ inline void ext_prim_test(SceneRef sc, int idx, const Path<double> &P, double tmin, ExtHit &H) {
    const int4 info = ext_ld_info<UNIFORM>(sc.ext_info, idx);
    if (info.x == RTMI_PRIM_MEDIUM) return; // media are evaluated after the surfaces, in index order (ext_medium_test)
    ext_prim_test_local<UNIFORM>(sc, idx, info.x, ext_local_ray<UNIFORM>(sc, info.z, info.w, P), P.time, tmin, H);
}

// The scan of a SMALL mixed-kind world (a Cornell box's 18 rectangles; what option flat_below sends here): all primitives in Hitlist order, every index
// wave-uniform -- scalar loads, scalar branches on kind and chain.  Consecutive primitives that share an instance chain (a Box's six rectangles under its
// Translate / RotateY wrappers, hitable.clj:391-486; the chain-less walls) share the LOCAL RAY -- evaluated once per run of equal chains instead of once per
// rectangle -- and the refined reciprocals of its three direction components: a rectangle's t costs three operations instead of the division's eleven.  No cull:
// a rectangle test at that price (~15 instructions) costs what the FP32 cull of its bounding sphere does.  Bit-identical to scan_all_cull_ext (RefinedRcp).
#ifndef RTMI_SMALL_SCAN_MAX
#define RTMI_SMALL_SCAN_MAX 64
#endif
// The records of primitive i + 1 (info, first five geometry slots: a rectangle's whole record) are requested BEFORE primitive i is tested: the scalar loads'
// latency runs under the test's vector instructions instead of in front of them (RTMI_SMALL_SCAN_PREFETCH).
#ifndef RTMI_SMALL_SCAN_PREFETCH
#define RTMI_SMALL_SCAN_PREFETCH 0 // measured: Cornell box 37.3 ms with the prefetch against 34.3 without -- four waves per SIMD already cover the scalar loads, and the extra SGPRs cost more
#endif
struct SmallRec { int4 info; double g0, g1, g2, g3, g4; };
__device__ inline SmallRec small_rec(SceneRef sc, int i) {
    SmallRec r;
    r.info = ext_ld_info<true>(sc.ext_info, i);
    const size_t gi = (size_t)i * 12;
    r.g0 = ext_ld<true>(sc.exact12, gi); r.g1 = ext_ld<true>(sc.exact12, gi + 1); r.g2 = ext_ld<true>(sc.exact12, gi + 2); r.g3 = ext_ld<true>(sc.exact12, gi + 3);
    r.g4 = ext_ld<true>(sc.exact12, gi + 4);
    return r;
}
__device__ inline void scan_small_ext(SceneRef sc, const Path<double> &P, double tmin, ExtHit &H, int lo = 0, int hi = 0x7fffffff) {
    const int n = min(sc.n_all, hi);
    int cf = -1, cc = -1; // the chain the cached local ray belongs to
    LocalRay lr = {P.ox, P.oy, P.oz, P.dx, P.dy, P.dz};
    RefinedRcp q[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    bool fast = false, fast_all = false;
    int i = max(lo, 0);
    if (i >= n) return;
#if RTMI_SMALL_SCAN_PREFETCH
    SmallRec nxt = small_rec(sc, i); // (the arrays are padded: index n is readable)
    for (; i < n; ++i) {
        const SmallRec rec = nxt;
        nxt = small_rec(sc, i + 1);
        const int4 info = rec.info;
#else
    for (; i < n; ++i) {
        const SmallRec rec = small_rec(sc, i); // info and the first five geometry slots (a rectangle's whole record) requested together: one wait per primitive, not two
        const int4 info = rec.info;
#endif
        if (info.x == RTMI_PRIM_MEDIUM) continue;
        if (info.z != cf || info.w != cc) { // (wave-uniform)
            cf = info.z; cc = info.w;
            lr = ext_local_ray<true>(sc, cf, cc, P);
            q[0] = refined_rcp(lr.dx); q[1] = refined_rcp(lr.dy); q[2] = refined_rcp(lr.dz);
            fast = rcp_in_range(lr.dx) && rcp_in_range(lr.dy) && rcp_in_range(lr.dz) && tmin >= 0x1p-300 && H.t <= 0x1p200;
            fast_all = __all(fast) != 0;
        }
        if (info.x >= RTMI_PRIM_RECT_XY && info.x <= RTMI_PRIM_RECT_YZ) { // a rectangle: its record is already here
            if (info.x == RTMI_PRIM_RECT_XY) box_face(rec.g4, lr.oz, q[2], fast, lr.ox, lr.dx, lr.oy, lr.dy, rec.g0, rec.g2, rec.g1, rec.g3, tmin, i, H, fast_all);
            else if (info.x == RTMI_PRIM_RECT_XZ) box_face(rec.g4, lr.oy, q[1], fast, lr.ox, lr.dx, lr.oz, lr.dz, rec.g0, rec.g2, rec.g1, rec.g3, tmin, i, H, fast_all);
            else box_face(rec.g4, lr.ox, q[0], fast, lr.oy, lr.dy, lr.oz, lr.dz, rec.g0, rec.g2, rec.g1, rec.g3, tmin, i, H, fast_all);
            continue;
        }
        ext_prim_test_local<true>(sc, i, info.x, lr, P.time, tmin, H, q, fast, fast_all);
    }
}

// ConstantMedium.hit? (hitable.clj:518-541): closest boundary hit on the whole line, closest boundary hit after it, the
// segment between them clipped to [t-min, t-max]; then ONE draw of the path's stream: hit-distance = -(log xi)/density
// against the length of the segment decides whether (where) the ray scatters inside.  t-min/t-max are the caller's
// un-narrowed interval, as in the reference's bvh-node descent (hitable.clj:99-105).
// The boundary part is a function of (medium, ray) alone -- no draw, no t-min / t-max --, so a medium the reference asks twice in a row (make-bvh stores a lone
// item as bvh-node(L, L) and hit? evaluates both children, hitable.clj:113-114: make-final's haze is media call 2 AND 3 of every ray) computes it once:
// MediumChord caches it for the next call with the same index (the index sequence is wave-uniform).
// The divisions of a medium's hit? -- two boundary roots by a = d.d, -(log xi) by the density, the hit distance by |d| (hitable.clj:183-207, 529-537) -- take the
// refined-reciprocal form (RefinedRcp: the IEEE sequence's own last three operations, the same bits while the division would not rescale its operands: checked per
// lane, numerator and divisor; anything else divides).  For a medium without an instance chain a, |d| and their reciprocals are the RAY's: medium_chord_begin
// evaluates them once per segment for all media calls (make-final: three calls, two media; 12 divisions and 2 square roots per segment before).
struct MediumChord { int idx; bool ok; double t1, t2, mag, rmag, rden, density; double ray_a, ray_ra, ray_mag, ray_rmag; };
__device__ inline bool num_in_range(double n) { return ((unsigned)__double2hiint(n) & 0x7fffffffu) - (54u << 20) < (1537u << 20); } // 2^-969 <= |n| < 2^568
__device__ inline MediumChord medium_chord_begin(const Path<double> &P) {
    MediumChord C;
    C.idx = -1; C.ok = false; C.t1 = C.t2 = C.mag = C.rmag = C.rden = C.density = 0.0;
    C.ray_a = dot3(P.dx, P.dy, P.dz, P.dx, P.dy, P.dz);
    C.ray_ra = refined_rcp(C.ray_a).r;
    C.ray_mag = rt_sqrt(C.ray_a);
    C.ray_rmag = refined_rcp(C.ray_mag).r;
    return C;
}
#define RTMI_MEDIA_FAST_MAX 8
__device__ inline void ext_medium_chord(SceneRef sc, int idx, const Path<double> &P, MediumChord &C, unsigned *cnt = nullptr, int k = -1) {
    const double FMAX = 3.4028234663852886e38;
    if (k >= 0 && k < RTMI_MEDIA_FAST_MAX && sc.media_fast[k][0] != 0.0) { // (wave-uniform) the one-sphere branch below with its operands from the descriptor
        C.idx = idx; C.ok = false; C.t1 = C.t2 = C.mag = C.rmag = 0.0;
        C.density = sc.media_fast[k][1];
        C.rden = refined_rcp(C.density).r;
        Prim4<double> s;
        s.cx = sc.media_fast[k][2]; s.cy = sc.media_fast[k][3]; s.cz = sc.media_fast[k][4]; s.r2 = sc.media_fast[k][5];
        const RefinedRcp qa = {C.ray_a, C.ray_ra};
        const bool fa = rcp_in_range(qa.a);
        double bq, cq, disc;
        sphere_test(s, P, qa.a, bq, cq, disc);
        if (!(disc >= 0.0)) { if (cnt) cnt[1] += 1u; return; }
        const double sq = rt_sqrt(disc);
        const double nA = -bq - sq;
        const double tA = div_by(nA, qa, fa && num_in_range(nA));
        const double tmin2 = tA + 0.0001;
        if (tA > -FMAX && tA < FMAX && !(tA > tmin2)) {
            if (cnt) cnt[1] += 2u;
            if (tmin2 >= 0.0 && bq > 0.0 && cq > 0.0) return;
            const double nB = -bq + sq;
            const double tB = div_by(nB, qa, fa && num_in_range(nB));
            if (!(tB > tmin2 && tB < FMAX)) return;
            C.mag = C.ray_mag; C.rmag = C.ray_rmag;
            C.t1 = tA; C.t2 = tB; C.ok = true;
            return;
        }
        // (a root that is not finite: the generic scans below)
    }
    const size_t gi = (size_t)idx * 12; // idx comes from media_idx: wave-uniform
    const int first = (int)ext_ld<true>(sc.exact12, gi + 1), count = (int)ext_ld<true>(sc.exact12, gi + 2);
    C.idx = idx; C.ok = false; C.t1 = C.t2 = C.mag = C.rmag = 0.0;
    C.density = ext_ld<true>(sc.exact12, gi);
    C.rden = refined_rcp(C.density).r;
    const int4 info = ext_ld_info<true>(sc.ext_info, idx);
    auto own_mag = [&]() { // |d| of the ray as the medium sees it (its own chain; none: the ray's)
        if (info.w == 0) { C.mag = C.ray_mag; C.rmag = C.ray_rmag; return; }
        const LocalRay rm = ext_local_ray<true>(sc, info.z, info.w, P);
        C.mag = rt_sqrt(dot3(rm.dx, rm.dy, rm.dz, rm.dx, rm.dy, rm.dz));
        C.rmag = refined_rcp(C.mag).r;
    };
    if (count == 1) { // (wave-uniform) a boundary that is ONE plain sphere -- make-final's two media, make-subsurface-sphere --: both closest-hit scans evaluate the same
        // quadratic (hitable.clj:183-207), so it is evaluated once: the scan over the whole line takes the first root, the scan from first root + 0.0001 the second
        // (the first cannot exceed itself + 0.0001).  The same operations on the same operands as the two generic scans below; anything unusual (a root that is
        // not finite, a moving sphere) takes them.
        const int4 binfo = ext_ld_info<true>(sc.ext_info, first);
        if (binfo.x == RTMI_PRIM_SPHERE || binfo.x == RTMI_PRIM_UVSPHERE) {
            const LocalRay r = ext_local_ray<true>(sc, binfo.z, binfo.w, P);
            const size_t bi = (size_t)first * 12;
            Prim4<double> s;
            s.cx = ext_ld<true>(sc.exact12, bi); s.cy = ext_ld<true>(sc.exact12, bi + 1); s.cz = ext_ld<true>(sc.exact12, bi + 2); s.r2 = ext_ld<true>(sc.exact12, bi + 3);
            Path<double> L; L.ox = r.ox; L.oy = r.oy; L.oz = r.oz; L.dx = r.dx; L.dy = r.dy; L.dz = r.dz;
            RefinedRcp qa;
            if (binfo.w == 0) { qa.a = C.ray_a; qa.r = C.ray_ra; } // (wave-uniform) the boundary stands in the world's frame: r is the ray itself
            else qa = refined_rcp(dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz));
            const double a = qa.a;
            const bool fa = rcp_in_range(a);
            double bq, cq, disc;
            sphere_test(s, L, a, bq, cq, disc);
            if (!(disc >= 0.0)) { if (cnt) cnt[1] += 1u; return; } // the line misses the boundary: no first hit
            const double sq = rt_sqrt(disc);
            const double nA = -bq - sq;
            const double tA = div_by(nA, qa, fa && num_in_range(nA));
            const double tmin2 = tA + 0.0001;
            if (tA > -FMAX && tA < FMAX && !(tA > tmin2)) { // the first scan's hit, and not the second scan's (else, i.e. never for finite roots: the generic scans)
                if (cnt) cnt[1] += 2u;
                if (tmin2 >= 0.0 && bq > 0.0 && cq > 0.0) return; // ext_prim_test's exact early-out of the second scan: both roots <= 0
                const double nB = -bq + sq;
                const double tB = div_by(nB, qa, fa && num_in_range(nB));
                if (!(tB > tmin2 && tB < FMAX)) return; // the second scan finds nothing
                own_mag();
                C.t1 = tA; C.t2 = tB; C.ok = true;
                return;
            }
        }
    }
    ExtHit h1 = {FMAX, 0x7fffffff, -1};
    for (int k = 0; k < count; ++k) ext_prim_test<true>(sc, first + k, P, -FMAX, h1); // idx (a medium of media_idx) is wave-uniform
    if (cnt) cnt[1] += (unsigned)count; // exact tests of the boundary's primitives
    if (!h1.any()) return;
    if (cnt) cnt[1] += (unsigned)count;
    ExtHit h2 = {FMAX, 0x7fffffff, -1};
    for (int k = 0; k < count; ++k) ext_prim_test<true>(sc, first + k, P, h1.t + 0.0001, h2);
    if (!h2.any()) return;
    own_mag();
    C.t1 = h1.t; C.t2 = h2.t; C.ok = true;
}
// hit_t (optional): receives the medium's hit parameter when it hits (the return value says whether)
// k (optional): the call's position in the media sequence (media_fast)
__device__ inline bool ext_medium_test(SceneRef sc, int idx, Path<double> &P, double tmin, double tmax, ExtHit &H, MediumChord &C, unsigned *cnt = nullptr, double *hit_t = nullptr, int k = -1) {
    if (C.idx != idx) ext_medium_chord(sc, idx, P, C, cnt, k); // (wave-uniform)
    if (!C.ok) return false;
    double t1 = C.t1, t2 = C.t2;
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (!(t1 < t2)) return false;
    if (t1 < 0.0) t1 = 0.0;
    const RefinedRcp qd = {C.density, C.rden}, qm = {C.mag, C.rmag};
    const double dist_in = (t2 - t1) * C.mag;
    const double lg = rt_log_unit(next_uniform(P));
    const double hit_distance = -div_by(lg, qd, rcp_in_range(qd.a) && num_in_range(lg));
    const bool hit = hit_distance < dist_in;
    const double t = t1 + div_by(hit_distance, qm, rcp_in_range(qm.a) && num_in_range(hit_distance)); // (for every lane that drew; folded under `hit`: no branch, no state copies)
    ext_update(H, t, idx, true, hit);
    if (hit_t) *hit_t = t;
    return hit;
}

// ---- a tree leaf's exact test from ONE record ------------------------------------------------------------------------------------------------
// ext_prim_test reads a primitive the way the flattened scene stores it: its info (kind, chain slice), then the chain's transforms, then the geometry --
// three dependent trips to the vector cache per leaf, at the 18 lanes a mixed-kind leaf phase runs with, and four waves per SIMD do not cover them
// (make-final: 24 % of the waves' time).  The host packs what the common leaves need into one 112-byte record per primitive (seven 16-byte loads issued
// together): header {kind, generic?, transform 1, transform 2} (transform: 0 none, 1 Translate, 2 RotateY; outermost first), five geometry values (a sphere's
// c r^2, a rectangle's u0 v0 u1 v1 k), two transforms' parameters (offset.xyz | sin cos -), and z0 of a Box leaf.  Primitives that need more (MovingSphere,
// Triangle, chains longer than two wrappers) are flagged generic and take ext_prim_test.  The arithmetic is ext_local_ray's and ext_prim_test_local's.
#define RTMI_LEAF_REC_DOUBLES 14
#ifndef RTMI_LEAF_RECORDS
#define RTMI_LEAF_RECORDS 1
#endif
__device__ inline void leaf_xform(int xk, double a, double b, double c, LocalRay &r) {
    if (xk == 1) { r.ox = r.ox - a; r.oy = r.oy - b; r.oz = r.oz - c; } // Translate: origin - offset (hitable.clj:394)
    else if (xk == 2) {                                                  // RotateY: a = sin, b = cos (hitable.clj:423-429)
        const double ox = b * r.ox - a * r.oz, oz = a * r.ox + b * r.oz;
        const double dx = b * r.dx - a * r.dz, dz = a * r.dx + b * r.dz;
        r.ox = ox; r.oz = oz; r.dx = dx; r.dz = dz;
    }
}
__device__ inline void ext_leaf_test(SceneRef sc, int idx, bool box, const Path<double> &P, double tmin, ExtHit &H) {
    const double2 *q = reinterpret_cast<const double2 *>(sc.leaf_rec + (size_t)idx * RTMI_LEAF_REC_DOUBLES);
    const int4 hdr = *reinterpret_cast<const int4 *>(q);
    const double2 q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
    if (hdr.y) { ext_prim_test<false>(sc, idx, P, tmin, H); return; } // (rare kinds, long chains)
    LocalRay r = {P.ox, P.oy, P.oz, P.dx, P.dy, P.dz};
    if (hdr.z | hdr.w) { // (skipped by a wave whose leaves carry no instance chain)
        leaf_xform(hdr.z, q3.y, q4.x, q4.y, r);
        leaf_xform(hdr.w, q5.x, q5.y, q6.x, r);
    }
    const double g0 = q1.x, g1 = q1.y, g2 = q2.x, g3 = q2.y, g4 = q3.x;
    if (box) { ext_box_faces(r, g0, g1, g2, g3, g4, q6.y, tmin, idx, H); return; }
    const int kind = hdr.x & 0xff; // (bit 8: the FlipNormals parity, read by resolve_hit_ext)
    if (kind <= RTMI_PRIM_UVSPHERE) {
        Prim4<double> s;
        s.cx = g0; s.cy = g1; s.cz = g2; s.r2 = g3;
        Path<double> L; L.ox = r.ox; L.oy = r.oy; L.oz = r.oz; L.dx = r.dx; L.dy = r.dy; L.dz = r.dz;
        const double a = dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz);
        double bq, cq, disc;
        sphere_test(s, L, a, bq, cq, disc);
        if (disc >= 0.0 && !(tmin >= 0.0 && bq > 0.0 && cq > 0.0)) {
            const double sq = rt_sqrt(disc);
            double t = (-bq - sq) / a;
            if (!(t > tmin)) t = (-bq + sq) / a;
            ext_update(H, t, idx, false, t > tmin);
        }
    } else { // a rectangle: plane axis a, in-plane axes u, v
        int ax, ua, va;
        rect_axes(kind, ax, ua, va);
        const double oa = pick3(ax, r.ox, r.oy, r.oz), da = pick3(ax, r.dx, r.dy, r.dz);
        const double t = (g4 - oa) / da;
        const double x = pick3(ua, r.ox, r.oy, r.oz) + t * pick3(ua, r.dx, r.dy, r.dz), y = pick3(va, r.ox, r.oy, r.oz) + t * pick3(va, r.dx, r.dy, r.dz);
        ext_update(H, t, idx, true, (t >= tmin) & (x >= g0) & (x <= g2) & (y >= g1) & (y <= g3));
    }
}

// the flat scan (FP32 cull + exact test) over all primitives
// [lo, hi): only the primitives of this index range (RTMI_MEDIA_HITLIST scans the list in pieces, between its media); default: all
__device__ inline void scan_all_cull_ext(SceneRef sc, const Path<double> &P, double a, double tmin, ExtHit &H, int lo = 0, int hi = 0x7fffffff) {
    const int n = min(sc.n_all, hi);
    if (n <= 0) return;
    const int last = n - 1;
    const CullRay c = make_cull_ray(P, a, sc.cull_t_lo, sc.cull_t_hi);
    for (int g = lo & ~3; g < n; g += 4) {
        const CullGroup G = load_cull_group(sc.cull20, g >> 2);
        const float d0 = cull_disc(G, 0, c), d1 = cull_disc(G, 1, c), d2 = cull_disc(G, 2, c), d3 = cull_disc(G, 3, c);
        if (fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)) >= 0.0f) {
            if (d0 >= 0.0f && g >= lo) ext_prim_test<true>(sc, g, P, tmin, H);
            if (d1 >= 0.0f && g + 1 <= last && g + 1 >= lo) ext_prim_test<true>(sc, g + 1, P, tmin, H);
            if (d2 >= 0.0f && g + 2 <= last && g + 2 >= lo) ext_prim_test<true>(sc, g + 2, P, tmin, H);
            if (d3 >= 0.0f && g + 3 <= last && g + 3 >= lo) ext_prim_test<true>(sc, g + 3, P, tmin, H);
        }
    }
}

__device__ inline float ext_best_hi(const ExtHit &H) { return float_above(H.t); }

// Time-sliced like scan_bvh (susp: RTMI_BVH_SUSPEND_WORDS_EXT columns behind the stack; returns false when the lane's traversal was
// suspended): the mixed-kind scenes need it most -- make-final's descent trips ran at 12.7 of 64 lanes, 66 % of them below 8.
#define RTMI_BVH_SUSPEND_WORDS_EXT 7 // node, tos, top, H.t (2 words), H.F, H.W   (no hit yet <=> H.F = 0x7fffffff)
template <bool SLICE = false, bool COUNT = false>
__device__ inline bool scan_bvh_ext(SceneRef sc, int *stack, const Path<double> &P, double a, double tmin, ExtHit &H, int *susp = nullptr, bool resume = false,
                                    int min_lanes = 0, unsigned *cnt = nullptr, int lo = 0, int hi = 0x7fffffff) {
    const BvhRay r = make_bvh_ray(sc, P, a, tmin);
    constexpr int stride = RTMI_BVH_STRIDE;
    int *sw = susp + threadIdx.x;
    BvhCursor cur;
    if (SLICE && resume) {
        cur.node = sw[0]; cur.tos = sw[stride];
        cur.top = reinterpret_cast<int *>(reinterpret_cast<char *>(stack) + sw[2 * stride]);
        H.t = __hiloint2double(sw[4 * stride], sw[3 * stride]);
        H.F = sw[5 * stride]; H.W = sw[6 * stride];
    } else {
        if (!r.ok) { if (COUNT) cnt[1] += (unsigned)sc.n_all; scan_all_cull_ext(sc, P, a, tmin, H, lo, hi); return true; }
        RTMI_PH(PH_BVH_SETUP)
        if (COUNT) cnt[1] += (unsigned)sc.n_big;
        for (int k = 0; k < sc.n_big; ++k) if (sc.big_idx[k] >= lo && sc.big_idx[k] < hi) ext_prim_test<true>(sc, sc.big_idx[k], P, tmin, H);
        cur = bvh_cursor_at_root(sc, stack);
        if (sc.n_mloc > 0 && !sc.media_seq && lo == 0 && H.t < 1e30) { // (wave-uniform) a bounded segment: does it lie inside a medium's neighbourhood?
            const float ox = (float)P.ox, oy = (float)P.oy, oz = (float)P.oz, tb = (float)H.t;
            const float px = fmaf(tb, (float)P.dx, ox), py = fmaf(tb, (float)P.dy, oy), pz = fmaf(tb, (float)P.dz, oz); // where the bound cuts the ray
            // both end points carry a float error below 2^-21 (|o| + |p|): the box is shrunk by 2^-16 of that, like the entry grid's rectangles are grown
            const float e = (fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + fmaxf(fmaxf(fabsf(px), fabsf(py)), fabsf(pz))) * (1.0f / 65536.0f);
            for (int k = 0; k < sc.n_mloc; ++k) {
                const float lx = sc.mloc_box[k][0] + e, ly = sc.mloc_box[k][1] + e, lz = sc.mloc_box[k][2] + e, hx = sc.mloc_box[k][3] - e, hy = sc.mloc_box[k][4] - e, hz = sc.mloc_box[k][5] - e;
                const bool in = ox >= lx && ox <= hx && oy >= ly && oy <= hy && oz >= lz && oz <= hz && px >= lx && px <= hx && py >= ly && py <= hy && pz >= lz && pz <= hz; // (a NaN: false)
                if (in) cur.node = sc.mloc_root[k]; // convex: the whole segment is inside
            }
        }
        RTMI_PH(PH_BIG)
    }
    auto leaf = [&](int code) { // one primitive, or the six faces of a Box (RTMI_LEAF_BOX: one chain, one local ray, three divisors)
        const int bits = ~code, idx = bits & 0x1fffffff;
        if (idx >= lo && idx < hi) {
            if (COUNT && (bits & RTMI_LEAF_BOX)) cnt[1] += 5;
#if RTMI_LEAF_RECORDS
            ext_leaf_test(sc, idx, (bits & RTMI_LEAF_BOX) != 0, P, tmin, H);
#else
            if (bits & RTMI_LEAF_BOX) ext_box_test<false>(sc, idx, P, tmin, H);
            else ext_prim_test<false>(sc, idx, P, tmin, H);
#endif
        }
    };
    auto best = [&]() { return ext_best_hi(H); };
    bvh_traverse<COUNT, SLICE>(sc, r, cur, min_lanes, leaf, best, cnt);
    if (SLICE && cur.node != RTMI_BVH_EMPTY) { // suspended
        sw[0] = cur.node; sw[stride] = cur.tos;
        sw[2 * stride] = (int)(reinterpret_cast<char *>(cur.top) - reinterpret_cast<char *>(stack));
        sw[3 * stride] = __double2loint(H.t); sw[4 * stride] = __double2hiint(H.t);
        sw[5 * stride] = H.F; sw[6 * stride] = H.W;
        return false;
    }
    if (!r.time_ok) { // e.g. after Isotropic.scatter, which sets the ray's time to the hit's t (shader.clj:136)
        if (COUNT) cnt[1] += (unsigned)sc.n_moving_all;
        for (int k = 0; k < sc.n_moving_all; ++k) if (sc.moving_all[k] >= lo && sc.moving_all[k] < hi) ext_prim_test<true>(sc, sc.moving_all[k], P, tmin, H);
    }
    return true;
}

// hitable.clj:219-252 (MovingSphere.hit?): centre = lerp(c0, c1, (time-t0)/(t1-t0)) per ray.
// Moving spheres are scanned after the static ones, so ties are resolved by original index.
template <typename R>
__device__ inline void scan_moving(SceneRef sc, const Path<R> &P, R a, R tmin, R &best_t, int &best_i, int &best_orig) {
    const R a2 = R(2.0) * a, a4 = R(4.0) * a;
    for (int m = 0; m < sc.n_moving; ++m) {
        const double *g = sc.mov_geom + (size_t)m * RTMI_PRIM_STRIDE;
        const R t0 = (R)g[7], t1 = (R)g[8];
        const R f = (P.time - t0) / (t1 - t0), omf = R(1.0) - f;
        const R cx = (R)g[0] * omf + (R)g[4] * f, cy = (R)g[1] * omf + (R)g[5] * f, cz = (R)g[2] * omf + (R)g[6] * f;
        const R rad = (R)g[3];
        const R ocx = P.ox - cx, ocy = P.oy - cy, ocz = P.oz - cz;
        const R b = R(2.0) * dot3(ocx, ocy, ocz, P.dx, P.dy, P.dz);
        const R c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - rad * rad;
        const R disc = b * b - a4 * c;
        if (disc >= R(0)) {
            const R sq = Real<R>::sqrt_(disc);
            const int orig = sc.mov_orig[m];
            R t = (-b - sq) / a2;
            bool ok = (t > tmin) && ((t < best_t) || (t == best_t && orig < best_orig));
            if (!ok) {
                t = (-b + sq) / a2;
                ok = (t > tmin) && ((t < best_t) || (t == best_t && orig < best_orig));
            }
            if (ok) { best_t = t; best_i = sc.n_static + m; best_orig = orig; }
        }
    }
}

// optional per-segment record for the parity probes (layout RTMI_SEG_REC)
struct SegLog { double *rec; int max_seg; int n; };

// the hit record {:t :p :uv :normal :material} (hitable.clj:196-199) of the winning primitive
template <typename R> struct HitRec { R t, px, py, pz, nx, ny, nz, u, v; int orig, kind, mat; };

// Rebuild the hit record of primitive `orig` (original Hitlist index) at parameter t: centre (hitable.clj:219-222 for moving
// spheres), p = point-at-parameter (util.clj:18-22), normal = normalise(p - centre) (hitable.clj:194),
// uv = get-sphere-uv for UVSphere (hitable.clj:128-139) else [0 0].
// One record per material: kind, its parameter, the root texture and -- for the common constant texture -- the colour itself.
// tex_kind = RTMI_TEX_CHECKER2 (record-only code): a Checkerboard of two Constant textures -- the two colours and the scale are in
// the record as well (r,g,b = tex0 = the colour where the sine product is negative; c1 = tex1), texture.clj:44-50.
#define RTMI_TEX_CHECKER2 100
// tex_kind = RTMI_TEX_GRADIENT_REC (record-only code): a UVGradient (texture.clj:26-34) -- e.g. the cover scene's sky dome, a third of all
// segments end on it -- whose twelve parameters sit in mat_grad[material]: one address, known with the material index, instead of the
// chain material -> texture index -> tex_kind -> tex_param
#define RTMI_TEX_GRADIENT_REC 101
// Dielectric: inv_ri = 1/ri (shader.clj:89) and r0 = ((1-ri)/(1+ri))^2 (schlick, shader.clj:71-72) depend on the material only: the
// host evaluates the same IEEE operations once per material instead of two FP64 divisions per scatter (FP64 kernels; RTMI_F32 evaluates
// them in float as before)
struct __attribute__((aligned(16))) MatRec { int mat_kind, tex, tex_kind, pad; double param, r, g, b; double scale, c1r, c1g, c1b; double inv_ri, r0; };

// `all_uv`: the probes report uv of every UVSphere hit; the trace kernel computes it (atan2 + asin) only where the hit material's
// texture reads uv, and only the coordinate it reads (bits RTMI_PRIM_NEEDS_U / _V of the device copy of prim_kind): nothing for a
// constant-colour dome; v alone (asin; no atan2) for the cover scene's sky, a UVGradient whose corner colours do not vary with u.  The
// coordinate nobody reads is set to 0.5: the u-lerp of two EQUAL colours c (1 - u) + c u is then c exactly, where the reference's own u
// gives c within an ulp -- a colour difference of the size the <= 2 ulp of atan2 / asin against the JVM's already allow (uv never
// feeds geometry).
#define RTMI_PRIM_NEEDS_U 32
#define RTMI_PRIM_NEEDS_V 64
#define RTMI_PRIM_NEEDS_UV (RTMI_PRIM_NEEDS_U | RTMI_PRIM_NEEDS_V)
template <typename R> __device__ inline void resolve_hit(SceneRef sc, const Path<R> &P, R t, int orig, HitRec<R> &h, bool all_uv = true) {
    const double *g = sc.exact12 + (size_t)orig * 12;
    R cx = (R)g[0], cy = (R)g[1], cz = (R)g[2];
    h.orig = orig;
    const int2 km = reinterpret_cast<const int2 *>(sc.prim_km)[orig];
    if ((km.x & 15) == RTMI_PRIM_MOVING) { // MovingSphere: centre at the ray's time (hitable.clj:219-222)
        const R t0 = (R)g[7], t1 = (R)g[8];
        const R f = (P.time - t0) / (t1 - t0), omf = R(1.0) - f;
        cx = (R)g[0] * omf + (R)g[4] * f; cy = (R)g[1] * omf + (R)g[5] * f; cz = (R)g[2] * omf + (R)g[6] * f;
    }
    const int kind_flags = km.x;
    h.kind = kind_flags & 15;
    h.mat = km.y;
    h.t = t;
    h.px = P.dx * t + P.ox; h.py = P.dy * t + P.oy; h.pz = P.dz * t + P.oz;
    R nx = h.px - cx, ny = h.py - cy, nz = h.pz - cz;
    const R len = Real<R>::sqrt_(dot3(nx, ny, nz, nx, ny, nz));
    if (len > R(0)) { const R inv = R(1.0) / len; nx = nx * inv; ny = ny * inv; nz = nz * inv; }
    h.nx = nx; h.ny = ny; h.nz = nz;
    h.u = R(0); h.v = R(0);
    RTMI_PH(PH_HITREC)
    if (h.kind == RTMI_PRIM_UVSPHERE && (all_uv || (kind_flags & RTMI_PRIM_NEEDS_UV))) {
        h.u = h.v = R(0.5);
        if (all_uv || (kind_flags & RTMI_PRIM_NEEDS_U)) h.u = Real<R>::sphere_u(nx, nz);
        if (all_uv || (kind_flags & RTMI_PRIM_NEEDS_V)) h.v = Real<R>::sphere_v(ny);
        RTMI_PH(PH_UV)
    }
}

// the hit record of an f3 primitive: the innermost record's {:t :p :uv :normal} in ITS frame, then every wrapper's
// outward step in reverse order -- RotateY post-rotates p and the normal (hitable.clj:441-446), Translate adds its offset
// to p (hitable.clj:396), FlipNormals negates the normal (hitable.clj:380; negation commutes exactly with the rotation)
// all_uv = false (the render): a rectangle's / triangle's uv only where the hit material's texture reads it (device copy of prim_kind: RTMI_PRIM_NEEDS_UV)
// the outward step of one wrapper (see below): Translate adds its offset to p, RotateY post-rotates p and the normal; a, b, c = the leaf record's parameters
__device__ inline void leaf_xform_back(int xk, double a, double b, double c, double &px, double &py, double &pz, double &nx, double &nz) {
    if (xk == 1) { px = px + a; py = py + b; pz = pz + c; }
    else if (xk == 2) { // a = sin, b = cos
        const double rx = b * px + a * pz, rz = (-(a * px)) + b * pz;
        const double mx = b * nx + a * nz, mz = (-(a * nx)) + b * nz;
        px = rx; pz = rz; nx = mx; nz = mz;
    }
}
__device__ inline void resolve_hit_ext(SceneRef sc, const Path<double> &P, double t, int orig, HitRec<double> &h, bool all_uv = true) {
#if RTMI_LEAF_RECORDS
    { // The winner's LEAF RECORD (ext_leaf_test) holds what the common kinds need -- kind, flip parity, up to two wrappers with their parameters, the geometry -- at ONE
      // address known from the index: seven loads issued together, where the generic path below reads the primitive's info, then walks its chain (a dependent
      // load per wrapper, twice: inwards for the local ray, outwards for p and the normal), then reads the geometry.  The same operations on the same operands.
        const double2 *q = reinterpret_cast<const double2 *>(sc.leaf_rec + (size_t)orig * RTMI_LEAF_REC_DOUBLES);
        const int4 hdr = *reinterpret_cast<const int4 *>(q);
        if (!hdr.y) {
            const double2 q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
            const int kind = hdr.x & 0xff;
            h.orig = orig; h.kind = kind; h.mat = sc.prim_mat[orig]; h.t = t;
            const bool want_uv = all_uv || (sc.prim_kind[orig] & RTMI_PRIM_NEEDS_UV);
            LocalRay r = {P.ox, P.oy, P.oz, P.dx, P.dy, P.dz};
            if (hdr.z | hdr.w) {
                leaf_xform(hdr.z, q3.y, q4.x, q4.y, r);
                leaf_xform(hdr.w, q5.x, q5.y, q6.x, r);
            }
            double px = r.dx * t + r.ox, py = r.dy * t + r.oy, pz = r.dz * t + r.oz;
            double nx, ny, nz;
            h.u = 0.0; h.v = 0.0;
            if (kind <= RTMI_PRIM_UVSPHERE) {
                nx = px - q1.x; ny = py - q1.y; nz = pz - q2.x;
                const double len = rt_sqrt(dot3(nx, ny, nz, nx, ny, nz));
                if (len > 0.0) { const double inv = 1.0 / len; nx = nx * inv; ny = ny * inv; nz = nz * inv; }
                if (kind == RTMI_PRIM_UVSPHERE && want_uv) Real<double>::sphere_uv(nx, ny, nz, &h.u, &h.v);
            } else { // a rectangle: u0 v0 u1 v1 k
                int ax, ua, va;
                rect_axes(kind, ax, ua, va);
                if (want_uv) {
                    const double x = pick3(ua, r.ox, r.oy, r.oz) + t * pick3(ua, r.dx, r.dy, r.dz), y = pick3(va, r.ox, r.oy, r.oz) + t * pick3(va, r.dx, r.dy, r.dz);
                    h.u = (x - q1.x) / (q2.x - q1.x); h.v = (y - q1.y) / (q2.y - q1.y);
                }
                nx = ax == 0 ? 1.0 : 0.0; ny = ax == 1 ? 1.0 : 0.0; nz = ax == 2 ? 1.0 : 0.0;
            }
            if (hdr.x & 0x100) { nx = -nx; ny = -ny; nz = -nz; }
            if (hdr.z | hdr.w) {
                leaf_xform_back(hdr.w, q5.x, q5.y, q6.x, px, py, pz, nx, nz);
                leaf_xform_back(hdr.z, q3.y, q4.x, q4.y, px, py, pz, nx, nz);
            }
            h.px = px; h.py = py; h.pz = pz; h.nx = nx; h.ny = ny; h.nz = nz;
            return;
        }
    }
#endif
    const int4 info = reinterpret_cast<const int4 *>(sc.ext_info)[orig];
    const LocalRay r = ext_local_ray(sc, info.z, info.w, P);
    const double *g = sc.exact12 + (size_t)orig * 12;
    const int kind = info.x;
    h.orig = orig; h.kind = kind; h.mat = sc.prim_mat[orig]; h.t = t;
    double px = r.dx * t + r.ox, py = r.dy * t + r.oy, pz = r.dz * t + r.oz; // point-at-parameter of the LOCAL ray
    double nx, ny, nz;
    h.u = 0.0; h.v = 0.0;
    if (kind <= RTMI_PRIM_MOVING) {
        double cx = g[0], cy = g[1], cz = g[2];
        if (kind == RTMI_PRIM_MOVING) {
            const double t0 = g[7], t1 = g[8];
            const double f = (P.time - t0) / (t1 - t0), omf = 1.0 - f;
            cx = g[0] * omf + g[4] * f; cy = g[1] * omf + g[5] * f; cz = g[2] * omf + g[6] * f;
        }
        nx = px - cx; ny = py - cy; nz = pz - cz;
        const double len = rt_sqrt(dot3(nx, ny, nz, nx, ny, nz));
        if (len > 0.0) { const double inv = 1.0 / len; nx = nx * inv; ny = ny * inv; nz = nz * inv; }
        if (kind == RTMI_PRIM_UVSPHERE && (all_uv || (sc.prim_kind[orig] & RTMI_PRIM_NEEDS_UV))) { // (both coordinates whenever the texture reads either: nothing is replaced on this path)
            Real<double>::sphere_uv(nx, ny, nz, &h.u, &h.v);
        }
    } else if (kind == RTMI_PRIM_MEDIUM) { // hitable.clj:536-540: uv [0 0] and normal (1,0,0) are arbitrary
        nx = 1.0; ny = 0.0; nz = 0.0;
    } else if (kind <= RTMI_PRIM_RECT_YZ) {
        int ax, ua, va;
        rect_axes(kind, ax, ua, va);
        if (all_uv || (sc.prim_kind[orig] & RTMI_PRIM_NEEDS_UV)) {
            const double x = pick3(ua, r.ox, r.oy, r.oz) + t * pick3(ua, r.dx, r.dy, r.dz), y = pick3(va, r.ox, r.oy, r.oz) + t * pick3(va, r.dx, r.dy, r.dz);
            h.u = (x - g[0]) / (g[2] - g[0]); h.v = (y - g[1]) / (g[3] - g[1]);
        }
        nx = ax == 0 ? 1.0 : 0.0; ny = ax == 1 ? 1.0 : 0.0; nz = ax == 2 ? 1.0 : 0.0;
    } else {
        if (all_uv || (sc.prim_kind[orig] & RTMI_PRIM_NEEDS_UV)) {
            double u = 0.0, v = 0.0, tt;
            tri_mt(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], r, u, v, tt);
            h.u = u; h.v = v;
        }
        cross3(g[3] - g[0], g[4] - g[1], g[5] - g[2], g[6] - g[0], g[7] - g[1], g[8] - g[2], nx, ny, nz); // not normalised (hitable.clj:570)
    }
    if (info.y) { nx = -nx; ny = -ny; nz = -nz; }
    for (int k = info.w - 1; k >= 0; --k) {
        const double *q = sc.ext_xf + (size_t)(info.z + k) * 4;
        if (q[0] == 0.0) { px = px + q[1]; py = py + q[2]; pz = pz + q[3]; }
        else {
            const double sn = q[1], cs = q[2];
            const double rx = cs * px + sn * pz, rz = (-(sn * px)) + cs * pz;
            const double mx = cs * nx + sn * nz, mz = (-(sn * nx)) + cs * nz;
            px = rx; pz = rz; nx = mx; nz = mz;
        }
    }
    h.px = px; h.py = py; h.pz = pz; h.nx = nx; h.ny = ny; h.nz = nz;
}
template <typename R, bool EXT> __device__ inline void resolve_any(SceneRef sc, const Path<R> &P, R t, int orig, HitRec<R> &h, bool all_uv = true) { resolve_hit<R>(sc, P, t, orig, h, all_uv); }
template <> __device__ inline void resolve_any<double, true>(SceneRef sc, const Path<double> &P, double t, int orig, HitRec<double> &h, bool all_uv) { resolve_hit_ext(sc, P, t, orig, h, all_uv); }

// Shader.scatter + Shader.emitted for the hit record, and the atten/accum update of core.clj:27-39.
// Returns true when the path continues (the `recur` of core.clj:30) with P holding the scattered ray.
// `att` (optional) receives the attenuation of a successful scatter.  `emit` receives accum + atten * emitted of a path that
// ends here (core.clj:37-39) with accum = (0 0 0): only a path's LAST segment can emit (the emitting Shader never scatters).
template <typename R, bool F4 = false> __device__ inline bool scatter_emit(SceneRef sc, Path<R> &P, const HitRec<R> &h, R *att, R *emit, bool passenger = false) {
    // The material switch is laid out in PHASES shared by the materials that need them (one rejection-sampler loop,
    // one |d| normalisation, one texture evaluation per trip) instead of one inlined copy per material: the lanes of a
    // wave hold different materials, so every copy would be executed serially.  Per lane the operations and the draw
    // order are exactly those of the material's own scatter.
    const int mat = h.mat;
    const double2 *mq = reinterpret_cast<const double2 *>(sc.mat_rec + (size_t)mat * 12);
    const double2 m0 = mq[0], m1 = mq[1], m2 = mq[2]; // 48 bytes: header, param + r, g + b
    const int mk = (int)__double2loint(m0.x), mtex = (int)__double2hiint(m0.x), mtk = (int)__double2loint(m0.y);
    const R mparam = (R)m1.x;
    const R px = h.px, py = h.py, pz = h.pz, nx = h.nx, ny = h.ny, nz = h.nz;
    const bool is_light = !passenger && mk == RTMI_MAT_DIFFUSE_LIGHT;
    const bool live = !passenger && !is_light && P.depth > 0; // core.clj:27: (and (pos? depth) (scatter ...))
    const bool is_lamb = live && mk == RTMI_MAT_LAMBERTIAN, is_metal = live && mk == RTMI_MAT_METAL, is_diel = live && mk == RTMI_MAT_DIELECTRIC;
    const bool is_iso = F4 && live && mk == RTMI_MAT_ISOTROPIC; // shader.clj:129-138, the phase function of ConstantMedium
    bool scat = false;
    emit[0] = emit[1] = emit[2] = R(0);
    R sdx = R(0), sdy = R(0), sdz = R(0); // scattered direction
    R atr = R(1), atg = R(1), atb = R(1);  // attenuation

    // phase 1 -- |d| and normalise(d): Metal (shader.clj:48) and refract (shader.clj:14) / Dielectric (shader.clj:84-88)
    R dmag = R(0), ux = P.dx, uy = P.dy, uz = P.dz;
    if (is_metal || is_diel) {
        dmag = Real<R>::sqrt_(dot3(ux, uy, uz, ux, uy, uz));
        if (dmag > R(0)) { const R inv = R(1.0) / dmag; ux = ux * inv; uy = uy * inv; uz = uz * inv; }
        RTMI_PH(PH_DNORM)
    }
    // phase 2 -- rand-in-unit-sphere: Lambertian (shader.clj:32) and Metal (shader.clj:53; drawn even when fuzz = 0)
    R rx = R(0), ry = R(0), rz = R(0);
    RTMI_PH(PH_DIRS) // (the material record fetch and the flags)
    rand_in_unit_sphere_wave(P, is_lamb || is_metal || is_iso, rx, ry, rz);
    RTMI_PH(PH_SAMPLER)
    // phase 3 -- directions
    if (is_lamb) { // shader.clj:29-34: target = (p + normal) + rand; dir = target - p
        const R tx = (px + nx) + rx, ty = (py + ny) + ry, tz = (pz + nz) + rz;
        sdx = tx - px; sdy = ty - py; sdz = tz - pz;
        scat = true;
        RTMI_PH(PH_DIRS)
    } else if (is_iso) { // (ray p (rand-in-unit-sphere) t): the scattered ray's TIME is the hit's t (shader.clj:136)
        sdx = rx; sdy = ry; sdz = rz;
        P.time = h.t;
        scat = true;
        RTMI_PH(PH_DIRS)
    } else if (is_metal) { // shader.clj:46-57 + reflect shader.clj:6-9
        const R fuzz = mparam;
        const R k = R(2.0) * dot3(ux, uy, uz, nx, ny, nz);
        const R rfx = ux - k * nx, rfy = uy - k * ny, rfz = uz - k * nz;
        sdx = rfx + fuzz * rx; sdy = rfy + fuzz * ry; sdz = rfz + fuzz * rz;
        scat = dot3(sdx, sdy, sdz, nx, ny, nz) > R(0);
        RTMI_PH(PH_DIRS)
    } else if (is_diel) { // shader.clj:76-102, refract 11-20, schlick 69-74
        const R ri = mparam;
        const R dn = dot3(P.dx, P.dy, P.dz, nx, ny, nz);
        R onx, ony, onz, eta, cosine;
        if (dn > R(0)) { onx = -nx; ony = -ny; onz = -nz; eta = ri; cosine = ri * (dn / dmag); }
        else { onx = nx; ony = ny; onz = nz; eta = sizeof(R) == sizeof(double) ? (R)mq[5].x : R(1.0) / ri; cosine = -(dn / dmag); }
        const R dt = dot3(ux, uy, uz, onx, ony, onz);
        const R disc = R(1.0) - (eta * eta) * (R(1.0) - dt * dt);
        // reflect(ray-direction, normal): un-normalised d, original normal
        const R k = R(2.0) * dn;
        sdx = P.dx - k * nx; sdy = P.dy - k * ny; sdz = P.dz - k * nz;
        if (disc > R(0)) {
            R r0;
            if (sizeof(R) == sizeof(double)) r0 = (R)mq[5].y;
            else { r0 = (R(1.0) - ri) / (R(1.0) + ri); r0 = r0 * r0; }
            const R prob = r0 + (R(1.0) - r0) * pow5(R(1.0) - cosine);
            if (!(next_uniform(P) < prob)) { // one draw, only when refraction is possible (shader.clj:91-93)
                const R sq = Real<R>::sqrt_(disc);
                sdx = eta * (ux - onx * dt) - onx * sq;
                sdy = eta * (uy - ony * dt) - ony * sq;
                sdz = eta * (uz - onz * dt) - onz * sq;
            }
        }
        scat = true;
        RTMI_PH(PH_DIRS)
    }
    // phase 4 -- ONE texture evaluation: emitted of DiffuseLight (shader.clj:118-119) or the albedo of a successful
    // Lambertian / Metal scatter (shader.clj:34,57)
    const bool want_tex = is_light || is_lamb || is_iso || (is_metal && scat);
    double turb = 0.0, turb_p0 = 0.0; // F4: a material whose texture IS a Marble / PerlinTurbulence gets its turbulence from the wave (perlin_turbulence_wave)
    bool turb_done = false;
    if (F4 && sizeof(R) == sizeof(double)) { // (wave-uniform: every active lane passes here)
        const bool want_turb = want_tex && (mtk == RTMI_TEX_MARBLE || mtk == RTMI_TEX_PERLIN_TURB);
        // ... when FEW lanes ask: a wave whose 64 camera rays all hit the marble sphere evaluates 64 turbulences in its lanes' own loops at full width (~600
        // instructions for all of them), where serving them one by one would cost 64 rounds; the rounds win below about five requests
        const int n_turb = __popcll(__ballot(want_turb));
        if (n_turb > 0 && n_turb <= RTMI_TURB_WAVE_MAX) {
            double p1 = 0.0;
            if (want_turb) { const double *tp = sc.tex_param + (size_t)mtex * RTMI_TEX_STRIDE; turb_p0 = tp[0]; p1 = tp[1]; }
            const double s = mtk == RTMI_TEX_PERLIN_TURB ? turb_p0 : 1.0; // (turbulence (mul scale p) depth) | marble: (turbulence p depth), texture.clj:70-84
            turb = perlin_turbulence_wave(sc, want_turb, s * (double)px, s * (double)py, s * (double)pz, (int)p1);
            turb_done = want_turb;
        }
    }
    if (want_tex) {
        R tr, tg, tb;
        if (mtk == RTMI_TEX_CONSTANT) { tr = (R)m1.y; tg = (R)m2.x; tb = (R)m2.y; } // texture.clj:14-16, the colour came with the record
        else if (F4 && turb_done) {
            if (mtk == RTMI_TEX_PERLIN_TURB) tr = tg = tb = R(0.5) * ((R)turb + R(1.0));
            else tr = tg = tb = R(0.5) * (Real<R>::sin_((R)turb_p0 * pz + R(10.0) * (R)turb) + R(1.0));
        }
        else if (mtk == RTMI_TEX_CHECKER2) { // texture.clj:44-50 with both children Constant
            const double2 m3 = mq[3], m4 = mq[4];
            const R scale = (R)m3.x;
            const int sx = sin_sign<R>(scale * px), sy = sin_sign<R>(scale * py), sz = sin_sign<R>(scale * pz);
            const bool neg = sx * sy * sz < 0;
            tr = neg ? (R)m1.y : (R)m3.y; tg = neg ? (R)m2.x : (R)m4.x; tb = neg ? (R)m2.y : (R)m4.y;
        } else if (mtk == RTMI_TEX_GRADIENT_REC) { // the operations of tex_sample's UVGradient case, operands from the material's second record
            const double2 *gq = reinterpret_cast<const double2 *>(sc.mat_grad + (size_t)mat * 12);
            const double2 g0 = gq[0], g1 = gq[1], g2 = gq[2], g3 = gq[3], g4 = gq[4], g5 = gq[5]; // co.rg co.b|cu.r cu.gb cv.rg cv.b|cuv.r cuv.gb
            const R u = h.u, v = h.v, omu = R(1.0) - u, omv = R(1.0) - v;
            const R a0 = (R)g1.y * omu + (R)g0.x * u, a1 = (R)g2.x * omu + (R)g0.y * u, a2 = (R)g2.y * omu + (R)g1.x * u;
            const R b0 = (R)g4.y * omu + (R)g3.x * u, b1 = (R)g5.x * omu + (R)g3.y * u, b2 = (R)g5.y * omu + (R)g4.x * u;
            tr = b0 * omv + a0 * v; tg = b1 * omv + a1 * v; tb = b2 * omv + a2 * v;
        } else tex_sample<R, F4>(sc, mtex, h.u, h.v, px, py, pz, tr, tg, tb);
        if (is_light) { emit[0] = R(0) + P.ar * tr; emit[1] = R(0) + P.ag * tg; emit[2] = R(0) + P.ab * tb; } // core.clj:37-39: (add accum (mul atten emitted))
        else { atr = tr; atg = tg; atb = tb; }
        RTMI_PH(PH_TEXTURE)
    }
    if (!scat) return false;
    if (att) { att[0] = atr; att[1] = atg; att[2] = atb; }
    // recur: scattered ray (time inherited), depth-1, atten*attenuation; accum unchanged (emitted of these is 0)
    P.ox = px; P.oy = py; P.oz = pz; P.dx = sdx; P.dy = sdy; P.dz = sdz;
    P.ar = P.ar * atr; P.ag = P.ag * atg; P.ab = P.ab * atb;
    P.depth -= 1;
    return true;
}

// One iteration of `color`'s loop after hit? has returned (core.clj:25-41).
template <typename R, bool EXT = false>
// passenger: a lane with no hit to shade in this trip (its segment is suspended in the tree, see trace_kernel).  It stays in the wave
// through scatter_emit -- the cooperative rand-in-unit-sphere wants all 64 lanes -- with every material flag off: P is untouched.
__device__ inline bool shade_segment(SceneRef sc, Path<R> &P, R t, int orig, SegLog *lg, R *emit, bool passenger = false) {
    emit[0] = emit[1] = emit[2] = R(0);
    if (!passenger && orig < 0) return false; // miss: (color) returns accum, core.clj:40-41
    HitRec<R> h;
    h.px = h.py = h.pz = h.nx = h.ny = h.nz = h.u = h.v = h.t = R(0); h.mat = 0; h.orig = 0; h.kind = 0;
    if (!passenger) {
        resolve_any<R, EXT>(sc, P, t, orig, h, false);
        RTMI_PH(PH_HITREC)
    }
    const bool scat = scatter_emit<R, EXT>(sc, P, h, nullptr, emit, passenger);
    if (lg && lg->n < lg->max_seg) {
        double *q = lg->rec + (size_t)lg->n * RTMI_SEG_REC;
        q[0] = h.orig; q[1] = h.t; q[2] = h.px; q[3] = h.py; q[4] = h.pz; q[5] = h.nx; q[6] = h.ny; q[7] = h.nz;
        q[8] = scat ? P.dx : 0; q[9] = scat ? P.dy : 0; q[10] = scat ? P.dz : 0; q[11] = scat ? 1.0 : 0.0;
        lg->n++;
    }
    return scat;
}

} // namespace rtmi
