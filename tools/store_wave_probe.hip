// Developer probe: a trajectory launch with ONE arithmetic wave per SIMD (the two-lane layout at 32 768 points: 1 024 waves,
// ~197 FP64 instructions and 2 KiB of row per wave and step) loses a third of its time to stores that find the memory
// queue full and stall the only wave.  Does a second, store-only wave per SIMD, fed through an LDS ring, take
// that stall off the arithmetic?
//   direct : 256-thread workgroups, every wave computes a row and stores it itself (what the product does)
//   ring   : 512-thread workgroups, waves 0-3 compute and write each row into an LDS ring (R slots of 2 KiB per wave),
//            waves 4-7 drain the ring to HBM; head / tail counters in LDS, no barrier after the first
// Same arithmetic (8 dependent chains x 24 v_fma_f64 per row), same bytes, same addresses.
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_wave_probe.hip -o tools/store_wave_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void row_math(double (&a)[8], int reps) {
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], 0.9999999, 1e-9);
        }
    }
}

// lane `gid` of the launch owns point gid >> 1, role gid & 1; it writes the pairs of waves 2*role and 2*role + 1
__device__ __forceinline__ d2 *row_ptr(d2 *traj, long long ld, int r, long long gid, int j) {
    const long long idx = gid >> 1;
    const int role = (int)(gid & 1);
    return traj + ((long long)r * 4 + 2 * role + j) * ld + idx;
}

__global__ void __launch_bounds__(256) direct_kernel(d2 *traj, long long ld, int rows, int reps) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (double)gid * 1e-9 + k;
    for (int r = 0; r < rows; ++r) {
        row_math(a, reps);
        __builtin_nontemporal_store((d2){a[0], a[1]}, row_ptr(traj, ld, r, gid, 0));
        __builtin_nontemporal_store((d2){a[2], a[3]}, row_ptr(traj, ld, r, gid, 1));
    }
}

template <int R>
__global__ void __launch_bounds__(512) ring_kernel(d2 *traj, long long ld, int rows, int reps, int *simd_of_wave, int *timeouts) {
    __shared__ d2 ring[4][R][2][64];
    __shared__ int head[4][64], tail[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, cw = w & 3;
    const bool storer = w >= 4;
    const long long gid = (long long)blockIdx.x * 256 + cw * 64 + lane;     // the arithmetic lane this thread is or mirrors
    if (!storer) head[cw][lane] = 0, tail[cw][lane] = 0;
    __syncthreads();
    if (blockIdx.x == 0 && lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        simd_of_wave[w] = (int)((hw >> 4) & 3);
    }
    // LDS byte offsets (the low half of a generic pointer into LDS is its offset) and hand-placed DS instructions: plain or
    // volatile C++ accesses through a generic pointer become FLAT instructions with a full wait after each one
    const unsigned slot0 = (unsigned)(unsigned long long)&ring[cw][0][0][lane];
    const unsigned my_head = (unsigned)(unsigned long long)&head[cw][lane], my_tail = (unsigned)(unsigned long long)&tail[cw][lane];
    auto lds_read_int = [](unsigned addr) -> int {
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        return __builtin_amdgcn_readfirstlane(v);
    };
    if (!storer) {
        double a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = (double)gid * 1e-9 + k;
        int tail_seen = 0;
        for (int r = 0; r < rows; ++r) {
            row_math(a, reps);
            if (r - tail_seen >= R) {                                  // ring full as far as this wave knows: look again
                int spins = 0;
                do {
                    tail_seen = lds_read_int(my_tail);
                    if (r - tail_seen < R) break;
                    __builtin_amdgcn_s_sleep(1);
                } while (++spins < (1 << 18));
                if (spins >= (1 << 18)) {                               // never hang: give up waiting for good
                    if (lane == 0) atomicAdd(timeouts, 1);
                    tail_seen = 0x3fffffff;
                }
            }
            const unsigned s = slot0 + (unsigned)(r & (R - 1)) * 2048u;
            const d2 p0 = (d2){a[0], a[1]}, p1 = (d2){a[2], a[3]};
            const int h = r + 1;
            // LDS serves a wave's operations in order: the row is in the ring before the counter says so.  No wait here.
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\tds_write_b32 %3, %4\n\ts_nop 0\n\ts_nop 0"
                         : : "v"(s), "v"(p0), "v"(p1), "v"(my_head), "v"(h) : "memory");
        }
    } else {
        int head_seen = 0;
        for (int r = 0; r < rows; ++r) {
            if (head_seen <= r) {
                int spins = 0;
                do {
                    head_seen = lds_read_int(my_head);
                    if (head_seen > r) break;
                    __builtin_amdgcn_s_sleep(2);
                } while (++spins < (1 << 18));
                if (spins >= (1 << 18)) {
                    if (lane == 0) atomicAdd(timeouts, 1);
                    return;
                }
            }
            const unsigned s = slot0 + (unsigned)(r & (R - 1)) * 2048u;
            d2 p0, p1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(p0), "=v"(p1) : "v"(s) : "memory");
            __builtin_nontemporal_store(p0, row_ptr(traj, ld, r, gid, 0));
            __builtin_nontemporal_store(p1, row_ptr(traj, ld, r, gid, 1));
            const int t = r + 1;
            asm volatile("ds_write_b32 %0, %1" : : "v"(my_tail), "v"(t) : "memory");
        }
    }
}

template <typename F>
static float time_launches(F launch, int launches) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < launches; ++rep) {
        if (rep == launches / 2) (void)hipEventRecord(e0);
        launch();
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / (launches - launches / 2);
}

int main(int argc, char **argv) {
    const long long points = argc > 1 ? atoll(argv[1]) : 32768;
    const int rows = argc > 2 ? atoi(argv[2]) : 3200;
    const long long lanes = 2 * points, ld = points;
    d2 *buf;
    int *d_simd, *d_to;
    if (hipMalloc(&buf, (size_t)rows * 4 * ld * sizeof(d2)) != hipSuccess) return 1;
    (void)hipMalloc(&d_simd, 8 * sizeof(int));
    (void)hipMalloc(&d_to, sizeof(int));
    (void)hipMemset(d_to, 0, sizeof(int));
    const double gb = (double)rows * 4 * ld * 16 / 1e9;
    printf("%lld points x %d rows, two lanes per point (%lld arithmetic waves), %.2f GB per launch\n", points, rows, lanes / 64, gb);
    for (int reps : {8, 4, 0}) {   // 192 / 96 / 0 FP64 instructions per row
        const float td = time_launches([&] { hipLaunchKernelGGL(direct_kernel, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, buf, ld, rows, reps); }, 60);
        const float t8 = time_launches([&] { hipLaunchKernelGGL(ring_kernel<8>, dim3((unsigned)(lanes / 256)), dim3(512), 0, 0, buf, ld, rows, reps, d_simd, d_to); }, 60);
        const float t16 = time_launches([&] { hipLaunchKernelGGL(ring_kernel<16>, dim3((unsigned)(lanes / 256)), dim3(512), 0, 0, buf, ld, rows, reps, d_simd, d_to); }, 60);
        printf("%3d FP64 instructions per row:  direct %.3f ms = %.0f GB/s | ring of 8 rows %.3f ms = %.0f GB/s | ring of 16 rows %.3f ms = %.0f GB/s\n",
               reps * 24, td, gb / td * 1e3, t8, gb / t8 * 1e3, t16, gb / t16 * 1e3);
    }
    int h_simd[8], h_to;
    (void)hipMemcpy(h_simd, d_simd, sizeof(h_simd), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&h_to, d_to, sizeof(int), hipMemcpyDeviceToHost);
    printf("SIMD of waves 0..7 of workgroup 0: %d %d %d %d | %d %d %d %d ; timeouts %d\n", h_simd[0], h_simd[1], h_simd[2], h_simd[3],
           h_simd[4], h_simd[5], h_simd[6], h_simd[7], h_to);
    // the last launch was the ring of 16 with no arithmetic: spot-check a few rows against what the lanes hold
    (void)hipFree(buf);
    return 0;
}
