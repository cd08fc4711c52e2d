// Developer probe: what the chip does when FP64 issue and HBM stores are both saturated -- the regime of the sweep's
// trajectory mode (save_every = 1: ~300 FP64 instructions and 64 B of stores per point per step, right at the ridge).
// Each lane runs NACC independent v_fma_f64 chains (`fmas` instructions per row in total) and writes four 16-B pairs per
// row in the trajectory layout.  Reports the sustained time per launch (200 launches back to back, mean of the last 100)
// for: FMAs only, stores only, both -- i.e. how well this chip overlaps the two streams when neither is ours.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_store_mix.hip -o tools/fp64_store_mix ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool FMA, bool STORE>
__global__ void __launch_bounds__(256) mix_kernel(d2 *traj, long long n, int rows, int reps, unsigned long long *ticks) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const unsigned long long t0 = __builtin_readcyclecounter();
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (double)idx * 1e-9 + k;
    for (int r = 0; r < rows; ++r) {
        if (FMA) {
#pragma unroll 19
            for (int it = 0; it < reps; ++it) {   // 19 x 8 = 152 straight-line FMAs per loop trip, like the unrolled RK4 step
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], 0.9999999, 1e-9);
            }
        }
        if (STORE) {
            d2 *dst = traj + (long long)r * 4 * n + idx;
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_nontemporal_store((d2){a[2 * j], a[2 * j + 1]}, dst + (long long)j * n);
        }
    }
    if (!STORE) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a[k];
        if (s == 12345.678) traj[idx] = (d2){s, s};
    }
    if (threadIdx.x == 0) ticks[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

template <bool FMA, bool STORE>
static void run(const char *name, d2 *buf, unsigned long long *d_ticks, long long n, int rows, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // 200 launches back to back with no host synchronisation in between (like bench.py --mode trajectory): the clock has
    // ~20 ms to settle; the mean of the LAST 100 launches is reported.
    const int launches = 200;
    unsigned long long h_ticks[1024];
    double tick_mean = 0;
    for (int rep = 0; rep < launches; ++rep) {
        if (rep == launches / 2) (void)hipEventRecord(e0);
        hipLaunchKernelGGL((mix_kernel<FMA, STORE>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, buf, n, rows, reps, d_ticks);
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms_total;
    (void)hipEventElapsedTime(&ms_total, e0, e1);
    (void)hipMemcpy(h_ticks, d_ticks, sizeof(h_ticks), hipMemcpyDeviceToHost);
    for (int i = 0; i < 1024; ++i) tick_mean += (double)h_ticks[i] / 1024;
    const float ms = ms_total / (launches - launches / 2);
    const double bytes = STORE ? (double)rows * 4 * n * 16 : 0.0;
    const double fmas = FMA ? (double)n / 64 * rows * reps * 8 : 0.0;   // wave instructions
    (void)tick_mean;
    printf("%-12s %7.3f ms per launch | %6.0f GB/s of stores | %.3f FP64 wave-instructions per cycle per SIMD (at 2.4 GHz; 0.25 = peak)\n",
           name, ms, bytes / ms / 1e6, fmas / (ms * 1e-3) / 1024 / 2.4e9);
}

int main() {
    const long long n = 262144;
    const int rows = 401;
    d2 *buf;
    unsigned long long *d_ticks;
    if (hipMalloc(&buf, (size_t)rows * 4 * n * sizeof(d2)) != hipSuccess) return 1;
    (void)hipMalloc(&d_ticks, 1024 * sizeof(unsigned long long));
    for (int reps : {19, 38}) {   // 152 / 304 FP64 instructions per row (the sweep kernel: 302-322)
        printf("-- %d FP64 instructions per row, 64 B per point per row, %lld points x %d rows, 4 waves per SIMD\n", reps * 8, n, rows);
        run<true, false>("fma only", buf, d_ticks, n, rows, reps);
        run<false, true>("stores only", buf, d_ticks, n, rows, reps);
        run<true, true>("both", buf, d_ticks, n, rows, reps);
    }
    (void)hipFree(buf);
    return 0;
}
