// Developer probe (optional): FOUR lanes per sweep point for the 4-wave float64 model -- is a third lane
// layout worth building for sweeps of <= 16 384 points (BASELINE config 1's single run)?  Lane r = lane & 3 of a quad holds
// one wave (0: pump 1, 1: pump 2, 2: signal, 3: idler); its pair partner is lane ^ 1, the other pair lane ^ 2 (DPP quad_perm).
// Same RK4 regrouping, phase recurrence (seeded every 64 steps) and arithmetic order per wave as rk4_sweep_split_kernel; no
// trajectory, no finite test: the point is the step rate and that the numbers agree.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -align-all-blocks=3 \
//        tools/quad_lane_probe.hip -o tools/libquad_probe.so ;  driven by tools/quad_lane_probe.py
#include <hip/hip_runtime.h>

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <int CTRL> __device__ __forceinline__ double xchg(const double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int PAIR = 0xB1;    // quad_perm:[1,0,3,2]
constexpr int OTHER = 0x4E;   // quad_perm:[2,3,0,1]

// out = base + c * dA/dz for the lane's own wave (c folded into g, tg, ha, E)
__device__ __forceinline__ void quad_stage(const double x, const double y, const double bx, const double by, const double Er,
                                           const double Ei, const double g, const double tg, const double ha, double &ox,
                                           double &oy) {
    const double p = fma_(x, x, y * y);
    const double s1 = p + xchg<PAIR>(p);
    const double s = s1 + xchg<OTHER>(s1);
    const double gj = fma_(-g, p, tg * s);
    const double X = xchg<PAIR>(x), Y = xchg<PAIR>(y);
    const double qr = fma_(x, X, -(y * Y)), qi = fma_(x, Y, y * X);
    const double Qr = xchg<OTHER>(qr), Qi = xchg<OTHER>(qi);
    const double Fr = fma_(Er, Qr, -(Ei * Qi)), Fi = fma_(Er, Qi, Ei * Qr);
    ox = fma_(Y, Fr, fma_(-X, Fi, fma_(-gj, y, fma_(ha, x, bx))));
    oy = fma_(X, Fr, fma_(Y, Fi, fma_(gj, x, fma_(ha, y, by))));
}

__global__ void __launch_bounds__(64) quad_kernel(const double *dbeta, const double *a0 /*[8]*/, double gamma, double alpha,
                                                  double z_max, int n_steps, long long n_points, double *a_end /*[N][8]*/) {
    const long long gid = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long idx = gid >> 2;
    const int role = (int)(gid & 3);
    if (idx >= n_points) return;
    double x = a0[2 * role], y = a0[2 * role + 1];
    const double g = gamma, tg = g + g, ha = -0.5 * alpha;
    const double dbd = (role < 2) ? dbeta[idx] : -dbeta[idx];
    const double hd = z_max / (double)n_steps, hh = 0.5 * hd;
    const double g_d = hh * g, tg_d = hh * tg, ha_d = hh * ha, g_h = hd * g, tg_h = hd * tg, ha_h = hd * ha;
    const double e_amp = tg_d, third = 1.0 / 3.0;
    double rc, rs, Er = e_amp, Ei = 0.0;
    sincos(dbd * hh, &rs, &rc);
    for (int i = 0; i < n_steps; ++i) {
        if ((i & 63) == 0) {
            double c, s;
            sincos(dbd * ((double)i * hd), &s, &c);
            Er = e_amp * c;
            Ei = e_amp * s;
        }
        double x2, y2, x3, y3, x4, y4, dx, dy;
        quad_stage(x, y, x, y, Er, Ei, g_d, tg_d, ha_d, x2, y2);
        double nr = fma_(Er, rc, -(Ei * rs)), ni = fma_(Er, rs, Ei * rc);
        Er = nr; Ei = ni;
        quad_stage(x2, y2, x, y, Er, Ei, g_d, tg_d, ha_d, x3, y3);
        quad_stage(x3, y3, x, y, Er + Er, Ei + Ei, g_h, tg_h, ha_h, x4, y4);
        const double tx = fma_(2.0, x3, fma_(-4.0, x, x2)) + x4, ty = fma_(2.0, y3, fma_(-4.0, y, y2)) + y4;
        nr = fma_(Er, rc, -(Ei * rs)); ni = fma_(Er, rs, Ei * rc);
        Er = nr; Ei = ni;
        quad_stage(x4, y4, tx, ty, Er, Ei, g_d, tg_d, ha_d, dx, dy);
        x = fma_(dx, third, x);
        y = fma_(dy, third, y);
    }
    a_end[idx * 8 + 2 * role] = x;
    a_end[idx * 8 + 2 * role + 1] = y;
}

extern "C" int quad_probe(long long n_points, int n_steps, double z_max, const double *dbeta, const double *a0, double gamma,
                          double alpha, double *a_end, double *kernel_ms, int reps) {
    double *d_db, *d_a0, *d_out;
    if (hipMalloc(&d_db, n_points * 8) != hipSuccess || hipMalloc(&d_a0, 64) != hipSuccess ||
        hipMalloc(&d_out, n_points * 64) != hipSuccess) return 1;
    hipMemcpy(d_db, dbeta, n_points * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_a0, a0, 64, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned grid = (unsigned)((4 * n_points + 63) / 64);
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(quad_kernel, dim3(grid), dim3(64), 0, 0, d_db, d_a0, gamma, alpha, z_max, n_steps, n_points, d_out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    hipMemcpy(a_end, d_out, n_points * 64, hipMemcpyDeviceToHost);
    *kernel_ms = best;
    hipFree(d_db); hipFree(d_a0); hipFree(d_out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
