// valu_cost.hip -- SIMD issue cost (cycles per wave64 instruction per SIMD) of the instruction classes trace_kernel issues,
// measured in register-only loops of 8 independent chains at 1 / 2 / 4 / 8 waves per SIMD.  The per-class prices bench.py's
// `roofline.frac` uses come from the 4-waves-per-SIMD column (the occupancy of the FP64 trace kernels).
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost            (prints a table + one JSON line)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

enum Op {
    OP_FMA_F32, OP_ADD_F32, OP_MUL_F32, OP_MAX3_F32, OP_MIN3_F32, OP_FMA_MIX, OP_PK_FMA_F32, OP_RCP_F32, OP_CVT_F32_F64, OP_CVT_F64_F32, OP_CVT_F64_U32,
    OP_FMA_F64, OP_ADD_F64, OP_MUL_F64, OP_RCP_F64, OP_RSQ_F64, OP_DIV_SCALE_F64, OP_DIV_FMAS_F64, OP_DIV_FIXUP_F64, OP_CMP_F64, OP_CMP_F32,
    OP_MUL_LO_U32, OP_MUL_HI_U32, OP_MAD_U64_U32, OP_ADD_U32, OP_ADDC_U32, OP_XOR_B32, OP_LSHR_B64, OP_LSHL_B64, OP_LSHR_B32, OP_ALIGNBIT, OP_CNDMASK,
    OP_MOV_B32, OP_BFE_U32, OP_AND_OR_B32, OP_MAD_U32_U24, OP_MAX_F32, OP_CMP_F32_SGPR, OP_CVT_F32_F16, OP_ADD3_U32, OP_LSHL_ADD_U64, OP_READFIRSTLANE, OP_BPERMUTE, OP_READLANE, OP_SALU_ADD, OP_VALU_SALU_MIX, OP_LDS_READ_B32, OP_LDS_WRITE_B32, OP_COUNT
};
static const char *kNames[OP_COUNT] = {
    "v_fma_f32", "v_add_f32", "v_mul_f32", "v_max3_f32", "v_min3_f32", "v_fma_mix_f32", "v_pk_fma_f32", "v_rcp_f32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_cvt_f64_u32",
    "v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_rsq_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_cmp_lt_f64", "v_cmp_lt_f32",
    "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_add_u32", "v_addc_co_u32", "v_xor_b32", "v_lshrrev_b64", "v_lshlrev_b64", "v_lshrrev_b32", "v_alignbit_b32", "v_cndmask_b32",
    "v_mov_b32", "v_bfe_u32", "v_and_or_b32", "v_mad_u32_u24", "v_max_f32", "v_cmp_lt_f32 -> sgpr pair", "v_cvt_f32_f16", "v_add3_u32", "v_lshl_add_u64", "v_readfirstlane_b32", "ds_bpermute_b32", "v_readlane_b32", "s_add_u32 (scalar only)", "v_fma_f32 + s_add_u32 (1:1)", "ds_read_b32", "ds_write_b32"};
// PMC class the instruction is counted in (SQ_INSTS_VALU_*), for bench.py's per-class pricing
static const char *kClass[OP_COUNT] = {
    "FMA_F32", "ADD_F32", "MUL_F32", "other", "other", "FMA_F32?", "FMA_F32?", "TRANS_F32", "CVT", "CVT", "CVT",
    "FMA_F64", "ADD_F64", "MUL_F64", "TRANS_F64", "TRANS_F64", "other_f64", "FMA_F64?", "other_f64", "other_f64", "other",
    "INT32", "INT32", "INT64", "INT32", "INT32", "INT32", "INT64", "INT64", "INT32", "INT32", "other",
    "other", "INT32", "INT32", "INT32", "other", "other", "CVT", "INT32", "INT64", "other", "lds", "other", "salu", "mix", "lds", "lds"};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ void __launch_bounds__(256) k(double *out, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    float f[8];
    double d[8];
    unsigned u[8];
    unsigned long long q[8];
    unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    unsigned long long smask = 0x5555555555555555ull * (seed | 1u), sm2 = 0;
    __shared__ unsigned lds[256 * 8];
    for (int i = 0; i < 8; ++i) {
        f[i] = 1.0f + 0.001f * (float)(lane + i + (int)seed);
        d[i] = 1.0 + 0.001 * (double)(lane + i + (int)seed);
        u[i] = (unsigned)(lane * 2654435761u + i * 40503u + seed);
        q[i] = (unsigned long long)u[i] * 0x9E3779B97F4A7C15ull + i;
        lds[threadIdx.x * 8 + i] = u[i];
    }
    __syncthreads();
    const float fm = 1.0000001f, fc = 0.5f;
    const double dm = 1.0000001, dc = 0.5;
    const unsigned um = 0x9E3779B9u, sh = 7u;
    const unsigned ldsa = (unsigned)(threadIdx.x * 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define B(i)                                                                                                                                    \
    if (OP == OP_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fc));                                             \
    if (OP == OP_ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fc));                                                          \
    if (OP == OP_MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fm));                                                          \
    if (OP == OP_MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fc));                                           \
    if (OP == OP_MIN3_F32) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fc));                                           \
    if (OP == OP_FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]" : "+v"(f[i]) : "v"(u[i]), "v"(fc));                     \
    if (OP == OP_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dm));                                                \
    if (OP == OP_RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));                                                                        \
    if (OP == OP_CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));                                                    \
    if (OP == OP_CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));                                                    \
    if (OP == OP_CVT_F64_U32) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));                                                    \
    if (OP == OP_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(dc));                                             \
    if (OP == OP_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));                                                          \
    if (OP == OP_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));                                                          \
    if (OP == OP_RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));                                                                        \
    if (OP == OP_RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));                                                                        \
    if (OP == OP_DIV_SCALE_F64) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[i]) : "v"(dm) : "vcc");                             \
    if (OP == OP_DIV_FMAS_F64) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(dc) : "vcc");                           \
    if (OP == OP_DIV_FIXUP_F64) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(dc));                                 \
    if (OP == OP_CMP_F64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(dm) : "vcc");                                              \
    if (OP == OP_CMP_F32) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(fm) : "vcc");                                              \
    if (OP == OP_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));                                                    \
    if (OP == OP_MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));                                                    \
    if (OP == OP_MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(um) : "vcc");                      \
    if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));                                                          \
    if (OP == OP_ADDC_U32) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(um) : "vcc");                                   \
    if (OP == OP_XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(um));                                                          \
    if (OP == OP_LSHR_B64) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(sh));                                                     \
    if (OP == OP_LSHL_B64) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(sh));                                                     \
    if (OP == OP_LSHR_B32) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(u[i]) : "v"(sh));                                                     \
    if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(u[i]) : "v"(sh));                                                \
    if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(um), "s"(smask));                                   \
    if (OP == OP_MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fm));                                                          \
    if (OP == OP_CMP_F32_SGPR) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm2) : "v"(f[i]), "v"(fm));                                    \
    if (OP == OP_CVT_F32_F16) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[i]) : "v"(u[i]));                                                    \
    if (OP == OP_ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(um), "v"(sh));                                           \
    if (OP == OP_LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));                                 \
    if (OP == OP_READFIRSTLANE) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(u[i]));                                              \
    if (OP == OP_MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(um));                                                              \
    if (OP == OP_BFE_U32) asm volatile("v_bfe_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(sh));                                                      \
    if (OP == OP_AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(um), "v"(sh));                                       \
    if (OP == OP_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(sh), "v"(um));                                     \
    if (OP == OP_BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ldsa));                                                 \
    if (OP == OP_READLANE) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(u[i]));                                                     \
    if (OP == OP_SALU_ADD) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");                                                   \
    if (OP == OP_VALU_SALU_MIX) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, %4" : "+v"(f[i]), "+s"(s0) : "v"(fm), "v"(fc), "s"(s1) : "scc"); \
    if (OP == OP_LDS_READ_B32) asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"(ldsa + 1024u * i));                                         \
    if (OP == OP_LDS_WRITE_B32) asm volatile("ds_write_b32 %1, %0" : : "v"(u[i]), "v"(ldsa + 1024u * i) : "memory");
            REP8(B)
#undef B
            if (OP == OP_BPERMUTE || OP == OP_LDS_READ_B32 || OP == OP_LDS_WRITE_B32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    double acc = 0.0;
    for (int i = 0; i < 8; ++i) acc += (double)f[i] + d[i] + (double)u[i] + (double)q[i];
    acc += (double)(s0 + s2 + s3) + (double)(sm2 & 1ull);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*kern_t)(double *, int, unsigned);
template <int OP> struct Tab { static void fill(kern_t *t) { t[OP] = k<OP>; Tab<OP + 1>::fill(t); } };
template <> struct Tab<OP_COUNT> { static void fill(kern_t *) {} };

int main(int argc, char **argv) {
    kern_t tab[OP_COUNT];
    Tab<0>::fill(tab);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = 2.4;
    double *out;
    hipMalloc(&out, sizeof(double) * (size_t)cus * 8 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int wps[4] = {1, 2, 4, 8};
    const int iters = 2048;
    printf("%-30s %-10s %8s %8s %8s %8s   (cycles per wave64 instruction per SIMD at %.1f GHz nominal; %d CUs)\n", "instruction", "pmc class", "1 w/SIMD", "2", "4", "8", ghz, cus);
    std::string json = "{";
    for (int op = 0; op < OP_COUNT; ++op) {
        double cyc[4];
        for (int w = 0; w < 4; ++w) {
            const int blocks = cus * wps[w]; // 256 threads = one wave per SIMD per block
            hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, 64, 1u);
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double inst_per_simd = (double)wps[w] * iters * 32 * (op == OP_VALU_SALU_MIX ? 1 : 1); // wave-instructions issued on one SIMD
            cyc[w] = best * 1e-3 * ghz * 1e9 / inst_per_simd;
        }
        printf("%-30s %-10s %8.2f %8.2f %8.2f %8.2f\n", kNames[op], kClass[op], cyc[0], cyc[1], cyc[2], cyc[3]);
        char buf[256];
        snprintf(buf, sizeof buf, "%s\"%s\": [%.3f, %.3f, %.3f, %.3f]", op ? ", " : "", kNames[op], cyc[0], cyc[1], cyc[2], cyc[3]);
        json += buf;
    }
    json += "}";
    printf("JSON %s\n", json.c_str());
    hipFree(out);
    return 0;
}
