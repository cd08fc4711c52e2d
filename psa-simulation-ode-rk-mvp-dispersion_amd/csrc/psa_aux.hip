// psa_aux.hip -- small gfx950 kernels around the sweep: layout changes between the NumPy-facing AoS
// buffers and the SoA device layout, a batched single RHS evaluation, and the gain summary reduction.
#include "psa_internal.h"

namespace psa {

// ---- AoS [n][nc]  <->  SoA [nc][n] -------------------------------------------------------------------
// One thread per point; SoA side coalesced.  Tiny (O(N*nc)), runs once per sweep in the host-buffer API.
template <typename T>
__global__ void __launch_bounds__(256) aos_to_soa_kernel(const T *__restrict__ aos, T *__restrict__ soa,
                                                         long long n, int nc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < nc; ++c) soa[(long long)c * n + i] = aos[i * nc + c];
}
template <typename T>
__global__ void __launch_bounds__(256) soa_to_aos_kernel(const T *__restrict__ soa, T *__restrict__ aos,
                                                         long long n, int nc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < nc; ++c) aos[i * nc + c] = soa[(long long)c * n + i];
}

// traj device layout [rows][nw][n] of (re, im) pairs -> NumPy layout [n][rows][nw] of pairs: a 2-D transpose of
// 16-B (f64) / 8-B (f32) elements.  A 32-point x 32-(row,wave) tile goes through LDS so that both the global reads
// (n fastest) and the global writes ((row, wave) fastest) are contiguous runs; +1 column: conflict-free.
// The host-buffer API transposes a CHUNK of points at a time into a bounded staging buffer: `soa` points at the chunk's
// first point inside the full [rows*nw][ld] device buffer, `aos` receives [n][rows*nw].
template <typename P2>
__global__ void __launch_bounds__(256) traj_to_aos_kernel(const P2 *__restrict__ soa, P2 *__restrict__ aos,
                                                          long long n, long long ld, long long rw_total) {
    __shared__ P2 tile[32][33];
    const long long p0 = (long long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 8 rows of 32 threads
    for (long long rw0 = (long long)blockIdx.y * 32; rw0 < rw_total; rw0 += (long long)gridDim.y * 32) {
        for (int r = ty; r < 32; r += 8) {
            const long long rw = rw0 + r;
            if (rw < rw_total && p0 + tx < n) tile[r][tx] = soa[rw * ld + p0 + tx];
        }
        __syncthreads();
        for (int p = ty; p < 32; p += 8) {
            const long long rw = rw0 + tx;
            if (rw < rw_total && p0 + p < n) aos[(p0 + p) * rw_total + rw] = tile[tx][p];
        }
        __syncthreads();
    }
}

template <typename T>
static hipError_t launch_a2s(hipStream_t s, const T *aos, T *soa, long long n, int nc) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((aos_to_soa_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, aos, soa, n, nc);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_s2a(hipStream_t s, const T *soa, T *aos, long long n, int nc) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((soa_to_aos_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, soa, aos, n, nc);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_t2a(hipStream_t s, const T *soa, T *aos, long long n, long long ld, long long rows, int nc) {
    if (n == 0 || rows == 0) return hipSuccess;
    typedef T P2 __attribute__((ext_vector_type(2)));
    const long long rw_total = rows * (nc / 2);
    const long long chunks = (rw_total + 31) / 32;
    const unsigned gy = (unsigned)(chunks < 64 ? chunks : 64);
    hipLaunchKernelGGL((traj_to_aos_kernel<P2>), dim3((unsigned)((n + 31) / 32), gy), dim3(256), 0, s,
                       reinterpret_cast<const P2 *>(soa), reinterpret_cast<P2 *>(aos), n, ld, rw_total);
    return hipGetLastError();
}

hipError_t launch_aos_to_soa_f64(hipStream_t s, const double *a, double *b, long long n, int nc) { return launch_a2s(s, a, b, n, nc); }
hipError_t launch_soa_to_aos_f64(hipStream_t s, const double *a, double *b, long long n, int nc) { return launch_s2a(s, a, b, n, nc); }
hipError_t launch_aos_to_soa_f32(hipStream_t s, const float *a, float *b, long long n, int nc) { return launch_a2s(s, a, b, n, nc); }
hipError_t launch_soa_to_aos_f32(hipStream_t s, const float *a, float *b, long long n, int nc) { return launch_s2a(s, a, b, n, nc); }
hipError_t launch_traj_to_aos_f64(hipStream_t s, const double *a, double *b, long long n, long long ld, long long r, int nc) { return launch_t2a(s, a, b, n, ld, r, nc); }
hipError_t launch_traj_to_aos_f32(hipStream_t s, const float *a, float *b, long long n, long long ld, long long r, int nc) { return launch_t2a(s, a, b, n, ld, r, nc); }

// ---- one RHS evaluation per point (yaman_model.py:10-52), terms kept separate ---------------------
// Written term by term in the reference's own grouping (linear + kerr) + fwm so the three partial
// outputs can be checked one by one (golden G5); AoS in and out (NumPy complex128 layout).
__global__ void __launch_bounds__(256) yaman_rhs_kernel(long long n, const double *__restrict__ z,
                                                        const double *__restrict__ a, const double *__restrict__ gamma,
                                                        const double *__restrict__ alpha, const double *__restrict__ dbeta,
                                                        double *__restrict__ out, double *__restrict__ lin,
                                                        double *__restrict__ kerr, double *__restrict__ fwm) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double x[4], y[4], p[4];
    for (int j = 0; j < 4; ++j) {
        x[j] = a[i * 8 + 2 * j];
        y[j] = a[i * 8 + 2 * j + 1];
        p[j] = x[j] * x[j] + y[j] * y[j];
    }
    const double g = gamma[i], al = alpha[i];
    double s, c;
    sincos(dbeta[i] * z[i], &s, &c);
    const double tot = (p[0] + p[1]) + (p[2] + p[3]);
    // FWM: pumps  2i*gamma*e+ * conj(partner pump) * A3*A4 ; sidebands 2i*gamma*e- * conj(partner) * A1*A2
    const double q12r = x[0] * x[1] - y[0] * y[1], q12i = x[0] * y[1] + y[0] * x[1];
    const double q34r = x[2] * x[3] - y[2] * y[3], q34i = x[2] * y[3] + y[2] * x[3];
    const double Fpr = c * q34r - s * q34i, Fpi = c * q34i + s * q34r;   // e+ * q34
    const double Fsr = c * q12r + s * q12i, Fsi = c * q12i - s * q12r;   // e- * q12
    const int partner[4] = {1, 0, 3, 2};
    for (int j = 0; j < 4; ++j) {
        const double f = p[j] + 2.0 * (tot - p[j]);
        const double lr = -0.5 * al * x[j], li = -0.5 * al * y[j];
        const double kr = -g * f * y[j], ki = g * f * x[j];
        const int q = partner[j];
        const double Fr = (j < 2) ? Fpr : Fsr, Fi = (j < 2) ? Fpi : Fsi;
        // conj(A_q) * F
        const double tr = x[q] * Fr + y[q] * Fi, ti = x[q] * Fi - y[q] * Fr;
        const double fr = -2.0 * g * ti, fi = 2.0 * g * tr;  // * 2i*gamma
        out[i * 8 + 2 * j] = (lr + kr) + fr;
        out[i * 8 + 2 * j + 1] = (li + ki) + fi;
        if (lin) { lin[i * 8 + 2 * j] = lr; lin[i * 8 + 2 * j + 1] = li; }
        if (kerr) { kerr[i * 8 + 2 * j] = kr; kerr[i * 8 + 2 * j + 1] = ki; }
        if (fwm) { fwm[i * 8 + 2 * j] = fr; fwm[i * 8 + 2 * j + 1] = fi; }
    }
}

hipError_t launch_yaman_rhs_f64(hipStream_t s, long long n, const double *z, const double *a, const double *gamma,
                                const double *alpha, const double *dbeta, double *out, double *lin, double *kerr,
                                double *fwm) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(yaman_rhs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, z, a, gamma, alpha,
                       dbeta, out, lin, kerr, fwm);
    return hipGetLastError();
}

// ---- gain summary: per-point gain + (max, argmax, count) over the sweep -------------------------------
// The only cross-lane step of the whole path: a 64-lane butterfly per wave (ds_swizzle / DPP via __shfl_xor),
// one LDS hop across the 4 waves of a block, then a second single-block pass over the per-block partials.
struct Best {
    double g;
    long long i;
    long long n;
};
__device__ __forceinline__ Best best_merge(const Best a, const Best b) {
    Best r;
    r.n = a.n + b.n;
    // larger gain wins; ties -> lower index (first maximum, like np.argmax); i < 0 means "none"
    const bool take_b = (b.i >= 0) && (a.i < 0 || b.g > a.g || (b.g == a.g && b.i < a.i));
    r.g = take_b ? b.g : a.g;
    r.i = take_b ? b.i : a.i;
    return r;
}
__device__ __forceinline__ Best wave_reduce(Best v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Best o;
        o.g = __shfl_xor(v.g, m, 64);
        o.i = __shfl_xor(v.i, m, 64);
        o.n = __shfl_xor(v.n, m, 64);
        v = best_merge(v, o);
    }
    return v;
}
__device__ __forceinline__ Best block_reduce(Best v, Best *sm) {
    v = wave_reduce(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    if (w == 0) {
        Best t = (lane < (int)(blockDim.x >> 6)) ? sm[lane] : Best{0.0, -1, 0};
        v = wave_reduce(t);
    }
    return v;
}

// T = the sweep's arithmetic type: p_metric and the per-point gain are T, the ratio and log10 are formed in float64
template <typename T>
__global__ void __launch_bounds__(256) gain_pass1(long long n, const T *__restrict__ p_metric,
                                                  const long long *__restrict__ first_bad, double p0, int gain_db,
                                                  T *__restrict__ gain_out, Best *__restrict__ partial) {
    __shared__ Best sm[4];
    Best acc{0.0, -1, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double p = (double)p_metric[i];
        double g = p / p0;
        // scan_mismtach.py:377-384: non-finite P3, non-finite or <= 0 gain -> NaN; :391 any exception -> NaN
        const bool ok = (p - p == 0.0) && (g - g == 0.0) && (g > 0.0) && (first_bad == nullptr || first_bad[i] < 0);
        if (ok && gain_db) g = 10.0 * log10(g);
        g = ok ? g : __builtin_nan("");
        if (gain_out) gain_out[i] = (T)g;
        if (ok) acc = best_merge(acc, Best{g, i, 1});
    }
    acc = block_reduce(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) gain_pass2(int n_partial, const Best *__restrict__ partial,
                                                  long long *__restrict__ best_index, double *__restrict__ best_gain,
                                                  long long *__restrict__ n_finite) {
    __shared__ Best sm[4];
    Best acc{0.0, -1, 0};
    for (int i = threadIdx.x; i < n_partial; i += 256) acc = best_merge(acc, partial[i]);
    acc = block_reduce(acc, sm);
    if (threadIdx.x == 0) {
        *best_index = acc.i;
        *best_gain = acc.i >= 0 ? acc.g : __builtin_nan("");
        *n_finite = acc.n;
    }
}

static int gain_blocks(long long n) {
    long long b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;  // 256 CUs x 8; grid-stride beyond that
    return (int)b;
}
long long gain_summary_workspace_bytes(long long n) { return (long long)gain_blocks(n) * (long long)sizeof(Best); }

template <typename T>
static hipError_t launch_gain_summary_t(hipStream_t s, long long n, const T *p_metric, const long long *first_bad,
                                        double p0_sig, int gain_db, T *gain_out, long long *best_index,
                                        double *best_gain, long long *n_finite, void *workspace) {
    const int nb = gain_blocks(n);
    hipLaunchKernelGGL(gain_pass1<T>, dim3(nb), dim3(256), 0, s, n, p_metric, first_bad, p0_sig, gain_db, gain_out,
                       (Best *)workspace);
    hipLaunchKernelGGL(gain_pass2, dim3(1), dim3(256), 0, s, nb, (const Best *)workspace, best_index, best_gain,
                       n_finite);
    return hipGetLastError();
}
hipError_t launch_gain_summary_f64(hipStream_t s, long long n, const double *p_metric, const long long *first_bad,
                                   double p0_sig, int gain_db, double *gain_out, long long *best_index,
                                   double *best_gain, long long *n_finite, void *workspace) {
    return launch_gain_summary_t<double>(s, n, p_metric, first_bad, p0_sig, gain_db, gain_out, best_index, best_gain, n_finite, workspace);
}
hipError_t launch_gain_summary_f32(hipStream_t s, long long n, const float *p_metric, const long long *first_bad,
                                   double p0_sig, int gain_db, float *gain_out, long long *best_index,
                                   double *best_gain, long long *n_finite, void *workspace) {
    return launch_gain_summary_t<float>(s, n, p_metric, first_bad, p0_sig, gain_db, gain_out, best_index, best_gain, n_finite, workspace);
}

}  // namespace psa
