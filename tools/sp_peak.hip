// Developer probe: sustained float32 VALU issue rate on this GPU -- v_fma_f32 (one float per lane) and v_pk_fma_f32 (two per
// lane) -- against waves per SIMD and independent chains per wave, no memory traffic.  Decides what "float32 vector peak" means
// for a kernel that gives each SIMD ONE wave (BASELINE config 4's per-GPU shard with two points per lane).
// Build: hipcc --offload-arch=gfx950 -O3 tools/sp_peak.hip -o tools/sp_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int NACC, bool PACKED>
__global__ void __launch_bounds__(256) fma_chain(float *out, int iters, float a, float b) {
    f2 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f2){threadIdx.x * 1e-3f + i, 1.0f + i};
    const f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (PACKED) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
                else acc[i].x = __builtin_fmaf(acc[i].x, a, b);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool PACKED>
void run(int waves_per_simd, int iters) {
    const int threads = 256 * 4 * 64 * waves_per_simd;
    float *out;
    (void)hipMalloc(&out, threads * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((fma_chain<NACC, PACKED>), dim3(threads / 256), dim3(256), 0, 0, out, iters, 1.0000001f, 1e-9f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)threads / 64 * iters * 8.0 * NACC;   // wave instructions
        if (rep == 2)
            printf("%s NACC=%2d waves/SIMD=%d: %7.2f ms  %6.1f TFLOP/s  %.3f wave-instr/cycle/SIMD at 2.4 GHz\n",
                   PACKED ? "v_pk_fma_f32" : "v_fma_f32   ", NACC, waves_per_simd, ms,
                   instr * 64 * (PACKED ? 4 : 2) / ms / 1e9, instr / (ms * 1e-3) / 1024.0 / 2.4e9);
    }
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<4, false>(w, 40000);
        run<16, false>(w, 20000);
        run<4, true>(w, 40000);
        run<16, true>(w, 20000);
    }
    return 0;
}
