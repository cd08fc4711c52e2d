// Developer probe: the store-only ceiling of the trajectory layout on this GPU.  Every lane writes one 16-byte
// (re, im) pair per wave per row into [row][wave][N] -- the same 1 KiB-per-wave-instruction stream the sweep kernel
// emits in trajectory mode (save_every = 1) -- with no arithmetic in between.  Compares default, non-temporal and
// "slc|glc"-style stores and different numbers of resident waves.
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_write_peak.hip -o tools/hbm_write_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

// layout experiment: [row][point block of 64][wave][64] -- a wave's four stores of a row are ONE contiguous 4 KiB run
template <bool NT>
__global__ void __launch_bounds__(256) rows_blocked_kernel(d2 *traj, long long n, int rows, int spin) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    double x = (double)idx * 1e-9, y = 1.0;
    const long long blk = idx >> 6, lane = idx & 63, nblk = (n + 63) >> 6;
    for (int r = 0; r < rows; ++r) {
        for (int k = 0; k < spin; ++k) {
            x = __builtin_fma(x, 1.0000001, 1e-9);
            y = __builtin_fma(y, 0.9999999, x);
        }
        d2 *dst = traj + (((long long)r * nblk + blk) * 4) * 64 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const d2 v = {x + j, y};
            if (NT) __builtin_nontemporal_store(v, dst + j * 64);
            else dst[j * 64] = v;
        }
    }
}

// layout experiment: the shipped [row][wave][N] order with a PADDED leading dimension (ld = n + pad points), so that the four
// wave regions of a row -- and consecutive rows -- are not a power of two apart
template <bool NT>
__global__ void __launch_bounds__(256) rows_padded_kernel(d2 *traj, long long n, long long ld, int rows, int spin) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    double x = (double)idx * 1e-9, y = 1.0;
    for (int r = 0; r < rows; ++r) {
        for (int k = 0; k < spin; ++k) {
            x = __builtin_fma(x, 1.0000001, 1e-9);
            y = __builtin_fma(y, 0.9999999, x);
        }
        d2 *dst = traj + (long long)r * 4 * ld + idx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const d2 v = {x + j, y};
            if (NT) __builtin_nontemporal_store(v, dst + (long long)j * ld);
            else dst[(long long)j * ld] = v;
        }
    }
}

template <bool NT>
__global__ void __launch_bounds__(256) rows_kernel(d2 *traj, long long n, int rows, int spin) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    double x = (double)idx * 1e-9, y = 1.0;
    for (int r = 0; r < rows; ++r) {
        for (int k = 0; k < spin; ++k) {  // stand-in for the FP64 work of a step (dependent chain)
            x = __builtin_fma(x, 1.0000001, 1e-9);
            y = __builtin_fma(y, 0.9999999, x);
        }
        d2 *dst = traj + (long long)r * 4 * n + idx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const d2 v = {x + j, y};
            if (NT) __builtin_nontemporal_store(v, dst + (long long)j * n);
            else dst[(long long)j * n] = v;
        }
    }
}

int main(int argc, char **argv) {   // [points = 262144] [rows = 401]: the two-lane trajectory shape is 32768 x 3201
    const long long n = argc > 1 ? atoll(argv[1]) : 262144;
    const int rows = argc > 2 ? atoi(argv[2]) : 401;
    printf("-- store-only ceiling, %lld points x %d rows x 4 waves x 16 B\n", n, rows);
    d2 *buf;
    const size_t bytes = (size_t)rows * 4 * n * sizeof(d2);
    if (hipMalloc(&buf, bytes + (size_t)rows * 4 * 4096 * sizeof(d2)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const bool quick = getenv("PEAK_QUICK") != nullptr;   // layout sweeps: non-temporal stores only, no FMA variants
    for (int nt = quick ? 1 : 0; nt < 2; ++nt)
        for (int spin : {0, 16}) {
            if (quick && spin) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(rows_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, buf, n, rows, spin);
                else hipLaunchKernelGGL(rows_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, buf, n, rows, spin);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            // the same launch 100 times back to back, no host synchronisation in between (sustained rate)
            hipEventRecord(e0);
            for (int rep = 0; rep < 100; ++rep) {
                if (nt) hipLaunchKernelGGL(rows_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, buf, n, rows, spin);
                else hipLaunchKernelGGL(rows_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, buf, n, rows, spin);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms100;
            hipEventElapsedTime(&ms100, e0, e1);
            printf("%s stores, %d dependent FMA pairs per row: isolated launch %.3f ms -> %.0f GB/s | 100 launches back to back "
                   "%.3f ms each -> %.0f GB/s (%.2f GB)\n", nt ? "non-temporal" : "default", spin, best, bytes / best / 1e6,
                   ms100 / 100, bytes / (ms100 / 100) / 1e6, bytes / 1e9);
        }
    for (int nt = quick ? 2 : 0; nt < 2; ++nt)
        for (int spin : {0, 16}) {
            const dim3 g((unsigned)((n + 255) / 256));
            for (int rep = 0; rep < 3; ++rep) {
                if (nt) hipLaunchKernelGGL(rows_blocked_kernel<true>, g, dim3(256), 0, 0, buf, n, rows, spin);
                else hipLaunchKernelGGL(rows_blocked_kernel<false>, g, dim3(256), 0, 0, buf, n, rows, spin);
            }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 100; ++rep) {
                if (nt) hipLaunchKernelGGL(rows_blocked_kernel<true>, g, dim3(256), 0, 0, buf, n, rows, spin);
                else hipLaunchKernelGGL(rows_blocked_kernel<false>, g, dim3(256), 0, 0, buf, n, rows, spin);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms100;
            hipEventElapsedTime(&ms100, e0, e1);
            printf("BLOCKED layout [row][block][wave][64], %s stores, %d dependent FMA pairs per row: 100 launches back to back %.3f ms "
                   "each -> %.0f GB/s\n", nt ? "non-temporal" : "default", spin, ms100 / 100, bytes / (ms100 / 100) / 1e6);
        }
    for (long long pad : {64LL, 272LL, 1040LL, 4096LL - 64})
        for (int nt = quick ? 1 : 0; nt < 2; ++nt) {
            const dim3 g((unsigned)((n + 255) / 256));
            const long long ld = n + pad;
            for (int rep = 0; rep < 3; ++rep) {
                if (nt) hipLaunchKernelGGL(rows_padded_kernel<true>, g, dim3(256), 0, 0, buf, n, ld, rows, 0);
                else hipLaunchKernelGGL(rows_padded_kernel<false>, g, dim3(256), 0, 0, buf, n, ld, rows, 0);
            }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 100; ++rep) {
                if (nt) hipLaunchKernelGGL(rows_padded_kernel<true>, g, dim3(256), 0, 0, buf, n, ld, rows, 0);
                else hipLaunchKernelGGL(rows_padded_kernel<false>, g, dim3(256), 0, 0, buf, n, ld, rows, 0);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms100;
            hipEventElapsedTime(&ms100, e0, e1);
            printf("PADDED ld = n + %lld points, %s stores: 100 launches back to back %.3f ms each -> %.0f GB/s\n", pad,
                   nt ? "non-temporal" : "default", ms100 / 100, bytes / (ms100 / 100) / 1e6);
        }
    hipFree(buf);
    return 0;
}
