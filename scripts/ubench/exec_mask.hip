// exec_mask.hip -- does a wave64 VALU instruction cost less when half of EXEC is zero?  (gfx950: SIMD-32, a wave64
// instruction takes two passes.)  Times a register-only FMA loop under four lane masks at full occupancy.
//   hipcc --offload-arch=gfx950 -O3 -o exec_mask exec_mask.hip && ./exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T> __global__ void __launch_bounds__(256) k(T *out, int iters, unsigned long long mask, T seed) {
    const int lane = threadIdx.x & 63;
    T a0 = seed + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const T m = (T)1.0000001, c = (T)0.5;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
                a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename T> void run(const char *name) {
    const int blocks = 256 * 8, iters = 4096;
    T *out;
    hipMalloc(&out, sizeof(T) * blocks * 256);
    const unsigned long long masks[5] = {~0ull, 0xffffffffull, 0xffffffff00000000ull, 0x5555555555555555ull, 0xffffull};
    const char *names[5] = {"all 64", "lanes 0-31", "lanes 32-63", "even lanes", "lanes 0-15"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 5; ++v) {
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, out, 64, masks[v], (T)1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, out, iters, masks[v], (T)1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)blocks * 4 * iters * 64; // wave-instructions (fma)
        printf("%s %-12s %8.3f ms  %.2f cycles/wave-instr/SIMD at 2.4 GHz\n", name, names[v], ms, ms * 1e-3 * 2.4e9 * 1024 / inst);
    }
    hipFree(out);
}
int main() { run<float>("f32"); run<double>("f64"); return 0; }
