// Developer probe: sustained v_fma_f64 issue rate on this GPU (independent accumulators, no memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 tools/dp_peak.hip -o tools/dp_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NACC>
__global__ void __launch_bounds__(256) fma_chain(double *out, int iters, double a, double b) {
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

template <int NACC>
void run(int waves_per_simd, int iters) {
    const int threads = 256 * 4 * 64 * waves_per_simd;  // CUs * SIMDs * lanes * waves
    double *out;
    hipMalloc(&out, threads * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fma_chain<NACC>, dim3(threads / 256), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double fma = (double)threads * iters * 8.0 * NACC;
        if (rep == 2)
            printf("NACC=%2d waves/SIMD=%d: %.2f ms  %.2f TFLOP/s  (%.3f DP wave-instr/cycle/SIMD at 2.4 GHz)\n", NACC,
                   waves_per_simd, ms, 2 * fma / ms / 1e9, fma / 64.0 / (ms * 1e-3) / 1024.0 / 2.4e9);
    }
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<1>(w, 40000);
        run<2>(w, 40000);
        run<4>(w, 40000);
        run<8>(w, 40000);
        run<16>(w, 20000);
    }
    return 0;
}
