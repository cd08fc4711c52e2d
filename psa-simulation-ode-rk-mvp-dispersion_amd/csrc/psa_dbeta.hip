// psa_dbeta.hip -- the phase mismatch of every sweep point, generated ON the device from the grid definition, so a
// multi-GPU shard needs no per-point input at all ("each rank generates its dbeta slice locally").
//
// Restates, per point of the flattened grid lambda_p2[n2] x lambda_signal[n3] (row-major, the order of
// scan_gain_grid / np.meshgrid(..., indexing="ij").ravel()):
//   frequency_plan.plan_from_wavelengths   (frequency_plan.py:291-327)  w_j = (2*pi*c)/lambda_j, w4 = (w1 + w2) - w3
//   frequency_plan.infer_symmetry_from_omegas (:215-255)                omega_c, omega_d, Omega + its consistency checks
//   dispersion.delta_beta_symmetric        (dispersion.py:321-372)      sum over even n of beta_n (Omega^n - omega_d^n) 2/n!
//   dispersion.delta_beta_from_omegas      (:282-318)                   (beta(w3) + beta(w4)) - (beta(w1) + beta(w2))
// with every failure the scalar functions raise mapped to an invalid point (dbeta = NaN), as the sweep drivers do
// (scan_mismtach.py:391-392, :736-738).
//
// Arithmetic follows the reference operation by operation in float64 -- this TU is compiled with -ffp-contract=off so
// that no multiply-add pair is fused -- because dbeta feeds exp(i*dbeta*z) over ~1e5 steps.  x**n for n >= 3 is libm's
// pow in the reference (correct to ~0.5 ulp); here it is formed in double-double and rounded once.  Result: bit-equal to
// the reference on every golden vector; against NumPy's array path bit-equal on 99.8 % of a random 10^6-point grid and
// within a few ulp (<= 2e-15 relative) where the two orders cancel (tests/test_gpu_dbeta.py, tools/dbeta_fuzz.py).
#include <hip/hip_runtime.h>

#include "psa_internal.h"

#pragma clang fp contract(off)

namespace psa {

__device__ __forceinline__ bool finite_(double x) { return x - x == 0.0; }
// np.isclose(lhs, rhs, atol, rtol) as frequency_plan._conserves states it
__device__ __forceinline__ bool conserves(double lhs, double rhs, double atol, double rtol) {
    return fabs(lhs - rhs) <= (atol + rtol * fabs(rhs));
}

// x**n for a small non-negative integer n with ONE rounding: n = 0, 1, 2 as NumPy's fast paths (1, x, x*x); n >= 3 by
// double-double multiplication (error ~2^-100 before the final rounding).
__device__ __forceinline__ double pow_int(double x, int n) {
    if (n == 0) return 1.0;
    if (n == 1) return x;
    if (n == 2) return x * x;
    double h = x, l = 0.0;
    for (int k = 2; k <= n; ++k) {
        const double p = h * x;
        double e = __builtin_fma(h, x, -p);   // exact low part of h*x (an explicit fma is not a contraction)
        e = e + l * x;
        const double s = p + e;
        l = e - (s - p);
        h = s;
    }
    return h;
}

__device__ __forceinline__ double factorial_(int n) {
    double f = 1.0;
    for (int k = 2; k <= n; ++k) f *= (double)k;
    return f;
}

// dispersion.delta_beta_symmetric: accumulate in the order given, ((beta_n * (Omega^n - omega_d^n)) * 2) / n!
__device__ __forceinline__ double dbeta_symmetric(const DbetaModel &m, double od, double Om) {
    double acc = 0.0;
    for (int k = 0; k < m.n_orders; ++k) {
        const int n = m.orders[k];
        const double bn = m.beta[n];
        if (bn != 0.0) acc = acc + bn * (pow_int(Om, n) - pow_int(od, n)) * 2.0 / factorial_(n);
    }
    return acc;
}

// dispersion._taylor_sum: sum_{n <= max_order} (beta_n * dw^n) / n!, skipping zero coefficients
__device__ __forceinline__ double beta_taylor(const DbetaModel &m, double w) {
    const double dw = w - m.omega_ref;
    double acc = 0.0;
    for (int n = 0; n <= m.max_order; ++n) {
        const double bn = m.beta[n];
        if (bn != 0.0) acc = acc + bn * pow_int(dw, n) / factorial_(n);
    }
    return acc;
}

template <typename T>
__global__ void __launch_bounds__(256) dbeta_grid_kernel(const DbetaModel m, const double lambda1,
                                                         const double *__restrict__ ax2, const long long n2,
                                                         const double *__restrict__ ax3, const long long n3,
                                                         const long long first, const long long n,
                                                         T *__restrict__ out, unsigned char *__restrict__ valid) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const long long i = first + t;
    const double l1 = lambda1, l2 = ax2[i / n3], l3 = ax3[i % n3];
    (void)n2;
    // plan_from_wavelengths_batch (its own tolerances: atol 0, rtol 1e-12)
    bool ok = finite_(l1) && finite_(l2) && finite_(l3) && l1 > 0.0 && l2 > 0.0 && l3 > 0.0;
    const double w1 = m.two_pi_c / l1, w2 = m.two_pi_c / l2, w3 = m.two_pi_c / l3;
    const double w4 = w1 + w2 - w3;
    ok = ok && finite_(w4) && w4 > 0.0;
    ok = ok && conserves(w1 + w2, w3 + w4, 0.0, 1e-12);
    // compute_phase_mismatch_batch
    ok = ok && finite_(w1) && finite_(w2) && finite_(w3) && w1 > 0.0 && w2 > 0.0 && w3 > 0.0;
    double db;
    if (m.method == 1) {  // GENERAL_TAYLOR
        ok = ok && conserves(w1 + w2, w3 + w4, m.atol, m.rtol);
        const double b1 = beta_taylor(m, w1), b2 = beta_taylor(m, w2), b3 = beta_taylor(m, w3), b4 = beta_taylor(m, w4);
        db = (b3 + b4) - (b1 + b2);
    } else {              // SYMMETRIC_EVEN: frequency_plan.symmetry_arrays, then the closed form
        ok = ok && conserves(w1 + w2, w3 + w4, m.atol, m.rtol);
        const double oc = 0.5 * (w1 + w2), od = 0.5 * (w1 - w2), Om = w3 - oc;
        ok = ok && finite_(oc) && finite_(od) && finite_(Om) && oc > 0.0 && fabs(od) < oc;
        const double r1 = oc + od, r2 = oc - od, r3 = oc + Om, r4 = oc - Om;
        ok = ok && r1 > 0.0 && r2 > 0.0 && r3 > 0.0 && r4 > 0.0;
        ok = ok && conserves(r1 + r2, r3 + r4, 0.0, 1e-12);
        ok = ok && conserves(r4, w4, m.atol, m.rtol);
        db = dbeta_symmetric(m, od, Om);
    }
    ok = ok && finite_(db);
    out[t] = ok ? (T)db : (T)__builtin_nan("");
    if (valid) valid[t] = ok ? 1 : 0;
}

// Six-wave grid (scan_six_wave_grid): pair k at omega_c +- Omega_k has dbeta_k = delta_beta_symmetric(omega_d, Omega_k);
// point i of the flattened Omega1[n1] x Omega2[n2] grid gets (dbeta_1[i / n2], dbeta_2[i % n2]).
template <typename T>
__global__ void __launch_bounds__(256) dbeta_pairs_kernel(const DbetaModel m, const double omega_d,
                                                          const double *__restrict__ ax1, const long long n1,
                                                          const double *__restrict__ ax2, const long long n2,
                                                          const long long first, const long long n,
                                                          T *__restrict__ out1, T *__restrict__ out2) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const long long i = first + t;
    (void)n1;
    out1[t] = (T)dbeta_symmetric(m, omega_d, ax1[i / n2]);
    out2[t] = (T)dbeta_symmetric(m, omega_d, ax2[i % n2]);
}

template <typename T>
static hipError_t launch_grid_t(hipStream_t s, const DbetaModel &m, double lambda1, const double *ax2, long long n2,
                                const double *ax3, long long n3, long long first, long long n, T *out,
                                unsigned char *valid) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((dbeta_grid_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m, lambda1, ax2, n2,
                       ax3, n3, first, n, out, valid);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_pairs_t(hipStream_t s, const DbetaModel &m, double omega_d, const double *ax1, long long n1,
                                 const double *ax2, long long n2, long long first, long long n, T *out1, T *out2) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((dbeta_pairs_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m, omega_d, ax1, n1,
                       ax2, n2, first, n, out1, out2);
    return hipGetLastError();
}

hipError_t launch_dbeta_grid_f64(hipStream_t s, const DbetaModel &m, double lambda1, const double *ax2, long long n2,
                                 const double *ax3, long long n3, long long first, long long n, double *out,
                                 unsigned char *valid) {
    return launch_grid_t<double>(s, m, lambda1, ax2, n2, ax3, n3, first, n, out, valid);
}
hipError_t launch_dbeta_grid_f32(hipStream_t s, const DbetaModel &m, double lambda1, const double *ax2, long long n2,
                                 const double *ax3, long long n3, long long first, long long n, float *out,
                                 unsigned char *valid) {
    return launch_grid_t<float>(s, m, lambda1, ax2, n2, ax3, n3, first, n, out, valid);
}
hipError_t launch_dbeta_pairs_f64(hipStream_t s, const DbetaModel &m, double omega_d, const double *ax1, long long n1,
                                  const double *ax2, long long n2, long long first, long long n, double *out1,
                                  double *out2) {
    return launch_pairs_t<double>(s, m, omega_d, ax1, n1, ax2, n2, first, n, out1, out2);
}
hipError_t launch_dbeta_pairs_f32(hipStream_t s, const DbetaModel &m, double omega_d, const double *ax1, long long n1,
                                  const double *ax2, long long n2, long long first, long long n, float *out1,
                                  float *out2) {
    return launch_pairs_t<float>(s, m, omega_d, ax1, n1, ax2, n2, first, n, out1, out2);
}

}  // namespace psa
