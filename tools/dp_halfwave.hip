// Developer probe: does a wave64 with only 32 (or 16) active lanes issue v_fma_f64 faster than a full wave?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ void fma_chain(double *out, int iters, double a, double b) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    for (int lanes : {64, 32, 16}) {
        for (int waves_per_simd : {1, 2, 4}) {
            const int blocks = 1024 * waves_per_simd;  // one wave per block
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(fma_chain<16>, dim3(blocks), dim3(lanes), 0, 0, out, iters, 1.0000001, 1e-9);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double instr_per_wave = (double)iters * 8 * 16;
            printf("active lanes %2d, waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD slot (at 2.4 GHz), lane-FMA rate %.1f T/s\n",
                   lanes, waves_per_simd, best, best * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd),
                   instr_per_wave * blocks * lanes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
