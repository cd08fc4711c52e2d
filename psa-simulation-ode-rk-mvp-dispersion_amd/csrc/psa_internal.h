// psa_internal.h -- shared between the kernel TUs and the C-ABI TU (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psa {

// Everything one sweep launch needs; passed by value as the kernel argument.
// All pointers are device pointers.  Strides are in elements: 1 = per point, 0 = broadcast.
template <typename T>
struct SweepArgs {
    const T *dbeta;      // [N]
    const T *dbeta2;     // [N] (6-wave) or nullptr
    const T *gamma;      // [N] | [1]
    const T *alpha;      // [N] | [1]
    const T *a0;         // SoA [2*NW][a0_ld]; a0_ld = N (per point) or 1 (broadcast, stride 0)
    T *a_end;            // SoA [2*NW][N]
    T *p_end;            // [N]
    T *p_max;            // [N]
    long long *first_bad;  // [N]
    T *traj;             // [n_saved][NW][traj_ld][2] ((re, im) pairs) or nullptr
    long long traj_ld;   // points per (row, wave) region of traj: n_points, or padded (psa_traj_ld)
    long long n_points;
    double z_max;
    int n_steps;
    int save_every;
    int gamma_stride, alpha_stride, a0_stride;  // 0 | 1
    long long a0_ld;
};

enum CheckMode : int { CHECK_NONE = 0, CHECK_BLOCK = 1, CHECK_EXACT = 2 };

// launchers (defined in psa_rk4_f64.hip / psa_rk4_f32.hip)
// lossless: the caller promises alpha == 0 for every point -> instantiation without the loss links
// split: 1 two lanes per point, 2 four lanes per point (4 waves) -- float64 only --, 0 one lane per point, -1 auto (psa_rk4_f64.hip)
hipError_t launch_sweep_f64(hipStream_t s, int n_waves, int check, bool lds, int block, bool lossless, int split,
                            const SweepArgs<double> &a);
hipError_t launch_sweep_f32(hipStream_t s, int n_waves, int check, bool lds, int block, int pack, bool lossless,
                            const SweepArgs<float> &a);  // pack: 1 two points/lane, 0 one, -1 auto

// aux kernels (psa_aux.hip)
hipError_t launch_aos_to_soa_f64(hipStream_t s, const double *aos, double *soa, long long n, int nc);
hipError_t launch_soa_to_aos_f64(hipStream_t s, const double *soa, double *aos, long long n, int nc);
hipError_t launch_aos_to_soa_f32(hipStream_t s, const float *aos, float *soa, long long n, int nc);
hipError_t launch_soa_to_aos_f32(hipStream_t s, const float *soa, float *aos, long long n, int nc);
// traj: device [rows][nw][n][2] -> NumPy [n][rows][nw][2]   (nc = 2*nw)
// `soa` points at the first point of the chunk; ld = points per row of the full device buffer (>= n)
hipError_t launch_traj_to_aos_f64(hipStream_t s, const double *soa, double *aos, long long n, long long ld, long long rows, int nc);
hipError_t launch_traj_to_aos_f32(hipStream_t s, const float *soa, float *aos, long long n, long long ld, long long rows, int nc);
hipError_t launch_yaman_rhs_f64(hipStream_t s, long long n, const double *z, const double *a, const double *gamma,
                                const double *alpha, const double *dbeta, double *out, double *lin, double *kerr,
                                double *fwm);
hipError_t launch_gain_summary_f64(hipStream_t s, long long n, const double *p_metric, const long long *first_bad,
                                   double p0_sig, int gain_db, double *gain_out, long long *best_index,
                                   double *best_gain, long long *n_finite, void *workspace);
hipError_t launch_gain_summary_f32(hipStream_t s, long long n, const float *p_metric, const long long *first_bad,
                                   double p0_sig, int gain_db, float *gain_out, long long *best_index,
                                   double *best_gain, long long *n_finite, void *workspace);
long long gain_summary_workspace_bytes(long long n);

// ---- device-side dbeta producer (psa_dbeta.hip) ------------------------------------------------------------
constexpr int DBETA_MAX_ORDER = 8;
struct DbetaModel {
    double beta[DBETA_MAX_ORDER + 1];  // beta_n per length unit, n = 0..8 (DispersionParams.get_beta_n)
    double omega_ref, two_pi_c, atol, rtol;
    int method;        // 0 SYMMETRIC_EVEN, 1 GENERAL_TAYLOR
    int n_orders;      // SYMMETRIC_EVEN: even orders, summed in this order
    int orders[4];
    int max_order;     // GENERAL_TAYLOR
};
hipError_t launch_dbeta_grid_f64(hipStream_t s, const DbetaModel &m, double lambda1, const double *ax2, long long n2,
                                 const double *ax3, long long n3, long long first, long long n, double *out,
                                 unsigned char *valid);
hipError_t launch_dbeta_grid_f32(hipStream_t s, const DbetaModel &m, double lambda1, const double *ax2, long long n2,
                                 const double *ax3, long long n3, long long first, long long n, float *out,
                                 unsigned char *valid);
hipError_t launch_dbeta_pairs_f64(hipStream_t s, const DbetaModel &m, double omega_d, const double *ax1, long long n1,
                                  const double *ax2, long long n2, long long first, long long n, double *out1,
                                  double *out2);
hipError_t launch_dbeta_pairs_f32(hipStream_t s, const DbetaModel &m, double omega_d, const double *ax1, long long n1,
                                  const double *ax2, long long n2, long long first, long long n, float *out1,
                                  float *out2);

}  // namespace psa
