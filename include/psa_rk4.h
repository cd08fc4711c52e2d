/*
 * psa_rk4.h -- C-ABI of libpsa_hip.so: the MI355X (gfx950) drop-in for the
 * reference's RK4 / Agrawal-Yaman hot path.
 *
 * The reference (Alxkov/PSA-simulation-ODE-RK-MVP-Dispersion) is pure Python and
 * has no FFI; its de-facto operator API is three nested call surfaces
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it
 * replaces (file:line into the upstream tree).  The binding a maintainer adds on
 * the reference side is a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer;
 *   - return value: 0 ok, < 0 argument error (PSA_E_*), > 0 a hipError_t;
 *     psa_last_error() gives the message for the calling thread;
 *   - a per-point NUMERICAL failure (NaN/Inf) is never a return code: it is
 *     reported in first_bad_step[N] (the reference raises FloatingPointError
 *     per run, integrators.py:132-135, and its sweep drivers turn that into a
 *     NaN gain, scan_mismtach.py:391-392);
 *   - "host" functions are blocking and take host pointers in NumPy layout
 *     (complex128 = interleaved re,im); "_dev" functions take device (HBM)
 *     pointers in SoA layout and are asynchronous on the given hipStream_t;
 *   - thread-safe, also for concurrent calls on one device; no global mutable state
 *     except the per-thread error string and a mutex-protected pool of idle
 *     per-device call contexts (stream, events, <= 64 MB of device scratch, 1 MB of
 *     page-locked memory each) that the host-buffer entry points reuse between
 *     calls; psa_release_cache() destroys them.
 *
 * Wave order everywhere: [pump1, pump2, signal, idler] (n_waves = 4) or
 * [pump1, pump2, signal1, idler1, signal2, idler2] (n_waves = 6, build-defined
 * extension; the reference has no 6-wave model).  The gain summary is taken on
 * wave index 2 (the signal), as scan_mismtach.py:376 does.
 */
#ifndef PSA_RK4_H
#define PSA_RK4_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- argument-error codes (negative) ------------------------------------- */
#define PSA_OK            0
#define PSA_E_NWAVES     -1   /* n_waves not 4 or 6                                  */
#define PSA_E_NPOINTS    -2   /* n_points < 0                                        */
#define PSA_E_NSTEPS     -3   /* n_steps <= 0                                        */
#define PSA_E_ZMAX       -4   /* z_max <= 0 or not finite  (integrators.py:188-189)  */
#define PSA_E_SAVE_EVERY -5   /* save_every <= 0           (integrators.py:108-109)  */
#define PSA_E_NULLPTR    -6   /* a required pointer is NULL                          */
#define PSA_E_DEVICE     -7   /* device index out of range / no gfx950 device        */
#define PSA_E_DBETA2     -8   /* n_waves == 6 needs dbeta2; n_waves == 4 forbids it  */
#define PSA_E_TOO_LARGE  -9   /* trajectory does not fit (int64 / device memory), or n_points exceeds the launch grid */
#define PSA_E_DBETA_MODEL -10 /* dbeta producer: unknown method, bad even_orders / max_order / beta count  */
#define PSA_E_FLAGS      -11  /* options that exclude each other (two of SPLIT_POINT / ONE_LANE / QUAD_POINT, F32_SCALAR + F32_PACKED, QUAD with 6 waves) */

/* The most points one launch takes: a launch has at most 2^32 - 1 threads in x and the two-lane float64 layout uses two
 * per point.  (2^31 - 256 float64 records are 189 GB: a 288 GB MI355X holds them, so the limit is stated, not theoretical.) */
#define PSA_MAX_POINTS   2147483392LL

/* ---- flags ----------------------------------------------------------------- */
/* broadcast: the array has ONE entry used for every sweep point */
#define PSA_BCAST_GAMMA      (1u << 0)
#define PSA_BCAST_ALPHA      (1u << 1)
#define PSA_BCAST_A0         (1u << 2)
/* options */
#define PSA_OPT_CHECK_NAN    (1u << 8)   /* SimulationConfig.check_nan (config.py:29): track first_bad_step over ALL
                                            n_steps (also the tail after the last saved row).  Without it
                                            first_bad_step is -1 everywhere and NaNs propagate silently.         */
#define PSA_OPT_EXACT_STEP   (1u << 9)   /* with CHECK_NAN: first_bad_step is the EXACT step index, as the reference's
                                            per-step test reports it (integrators.py:132-135).  float64: free -- the
                                            forward pass tests once per saved row and a wave with a newly failing point
                                            replays the steps since the previous test with a per-step test; float32:
                                            tested in the loop.  Without the flag first_bad_step is the LAST step of
                                            the first non-finite save block (the sweep drivers only need "did it
                                            fail").                                                                */
#define PSA_OPT_LDS_STAGING  (1u << 10)  /* keep y / y_stage / k-accumulator in LDS instead of VGPRs (the layout
                                            the north-star sketches; slower -- kept for the A/B in DESIGN.md)    */
#define PSA_OPT_BLOCK64      (1u << 11)  /* 64-thread workgroups (one wave) instead of 256                       */
#define PSA_OPT_LOSSLESS     (1u << 14)  /* the caller promises alpha == 0 for EVERY point: use the instantiation without
                                            the -alpha/2 terms (the reference's own `alpha == 0.0` branch,
                                            yaman_model.py:130-131; -10 % instructions).  The host-buffer entry points
                                            set it themselves when alpha is a broadcast 0; `_dev` callers pass it.    */
#define PSA_OPT_SPLIT_POINT  (1u << 15)  /* float64 only: force TWO LANES per sweep point (a point's waves divided between
                                            neighbouring lanes, partial sums exchanged by DPP): halves the sequential
                                            instruction stream per lane.  Default: chosen automatically when the sweep is
                                            smaller than the chip (2*N lanes still get one SIMD per wave: N <= 32 768).   */
#define PSA_OPT_QUAD_POINT   (1u << 18)  /* float64, n_waves == 4 only: force FOUR lanes per sweep point (one wave of the model
                                            per lane, sums and products exchanged by DPP within the quad): ~155
                                            instructions per step and lane.  Default: chosen automatically while 4*N
                                            lanes still get one SIMD per wave (N <= 16 384) -- the reference's own
                                            scenarios (1, 30, 100 points).                                          */
#define PSA_OPT_ONE_LANE     (1u << 16)  /* float64 only: never split a point over two lanes                              */
#define PSA_OPT_TRAJ_LD      (1u << 17)  /* `_dev` entry points: the trajectory buffer is [n_saved][n_waves][ld][2] with the
                                            leading dimension ld = psa_traj_ld(n_points, sizeof(element)) >= n_points
                                            instead of n_points: sizes whose wave regions would lie a multiple of 2 MiB
                                            apart are padded by 4 352 B, which lifts the store rate of every-step
                                            trajectories by 8-35 %.  The host-buffer entry points use
                                            it internally; the caller's array stays dense.                          */
#define PSA_OPT_F32_SCALAR   (1u << 12)  /* float32 only: force one sweep point per lane                          */
#define PSA_OPT_F32_PACKED   (1u << 13)  /* float32 only: force two points per lane (v_pk_fma_f32 packed math);
                                            this is also the default whenever n_points >= 2                      */

/* ---- environment ----------------------------------------------------------- */
int         psa_device_count(void);          /* number of visible HIP devices (0 if none / no driver) */
const char *psa_last_error(void);            /* message of the last failure on this thread            */
const char *psa_version(void);               /* "psa-hip <semver> gfx950"                              */
int64_t     psa_n_saved(int64_t n_steps, int32_t save_every);   /* n_steps / save_every + 1, integrators.py:115 */
int64_t     psa_traj_ld(int64_t n_points, int32_t elem_size);   /* leading dimension for PSA_OPT_TRAJ_LD (elem_size 4 | 8) */
int         psa_release_cache(void);         /* destroy the idle call contexts of every device; returns how many       */

/* ---- B3/B2: the sweep (host buffers, blocking) ----------------------------------
 * Replaces the body of the per-point loops scan_mismtach.py:357-392 and :694-738, i.e.
 * N x { simulation.run_single_simulation (simulation.py:349-357) -> integrators.integrate_interval
 * (integrators.py:150-204) -> integrate_fixed_step (:68-142) -> rk4_step (:25-61) ->
 * yaman_model.rhs_yaman_simplified (yaman_model.py:10-52) } plus the reduction
 * P3 = |A[:,2]|^2, max / last over saved rows (scan_mismtach.py:376-381, :27-40).
 *
 *   n_steps    = int(round(z_max/dz)) computed by the caller (integrators.py:194);
 *                the kernel steps on z_i = i * (z_max / n_steps)  (np.linspace, :195)
 *   dbeta      [N]   phase mismatch per point, 1/length          (parameters.py:236 CacheParams.delta_beta_1_m)
 *   dbeta2     [N]   second pair's mismatch (n_waves == 6) or NULL
 *   gamma      [N] | [1]    fiber.gamma_W_m   (parameters.py:166)
 *   alpha      [N] | [1]    fiber.alpha_1_m
 *   a0_re_im   [N][n_waves][2] | [1][n_waves][2]   initial amplitudes (simulation.py:103-123)
 *   a_end_re_im[N][n_waves][2]   state at the LAST SAVED row = step (n_steps/save_every)*save_every
 *   p_sig_end  [N]   |A_sig|^2 at that row            (gain_mode "end", scan_mismtach.py:36-37)
 *   p_sig_max  [N]   max over saved rows incl. z = 0  (gain_mode "max", :38-39; NaN-propagating like np.max)
 *   first_bad_step [N]  -1, or the 0-based step index after which the state was non-finite
 *   traj_or_null   [N][n_saved][n_waves][2]  every saved row (integrators.py:137-140), or NULL.  A launch with a
 *                  trajectory takes at most 2^27 - 1 points in float64 (2^28 - 1 in float32: rows are addressed with a
 *                  32-bit lane offset kept below 2^31; with PSA_OPT_SPLIT_POINT 2^32 / (n_waves * 16) - 1) and must fit
 *                  the device's free memory, else PSA_E_TOO_LARGE; the host-buffer
 *                  variant moves it to the host in bounded chunks (two 256 MB staging buffers)
 *   elapsed_ms_or_null  kernel time from hipEvents on the launch stream, or NULL
 */
int psa_rk4_sweep_f64(int device, int n_waves, int64_t n_points, int64_t n_steps, double z_max,
                      int32_t save_every, const double *dbeta, const double *dbeta2, const double *gamma,
                      const double *alpha, const double *a0_re_im, uint32_t flags, double *a_end_re_im,
                      double *p_sig_end, double *p_sig_max, int64_t *first_bad_step, double *traj_or_null,
                      double *elapsed_ms_or_null);

/* float32 state/arithmetic variant (BASELINE config 4; build-defined, the reference forces complex128,
 * yaman_model.py:41-44).  The phase dbeta*z is still formed in float64.  z_max stays double. */
int psa_rk4_sweep_f32(int device, int n_waves, int64_t n_points, int64_t n_steps, double z_max,
                      int32_t save_every, const float *dbeta, const float *dbeta2, const float *gamma,
                      const float *alpha, const float *a0_re_im, uint32_t flags, float *a_end_re_im,
                      float *p_sig_end, float *p_sig_max, int64_t *first_bad_step, float *traj_or_null,
                      double *elapsed_ms_or_null);

/* ---- the same sweep on buffers already resident in HBM (async on `stream`) -------
 * Device layout is SoA so that every wave instruction is a contiguous 512-B (f64) access:
 *   d_a0_soa    [2*n_waves][N] (or [2*n_waves][1] with PSA_BCAST_A0): row 2j = Re A_j, 2j+1 = Im A_j
 *   d_a_end_soa [2*n_waves][N]
 *   d_traj_soa  [n_saved][n_waves][ld][2] ((re, im) pairs: 16-B stores, 1 KiB per wave instruction) or NULL;
 *               ld = N, or psa_traj_ld(N, sizeof(element)) with PSA_OPT_TRAJ_LD (rows padded off a 2 MiB stride)
 * `stream` is a hipStream_t (NULL = default stream).  No allocation, no synchronisation: safe to capture
 * into a hipGraph.  This is what bench.py times and what the multi-GPU path calls per rank.
 */
int psa_rk4_sweep_f64_dev(void *stream, int n_waves, int64_t n_points, int64_t n_steps, double z_max,
                          int32_t save_every, const double *d_dbeta, const double *d_dbeta2,
                          const double *d_gamma, const double *d_alpha, const double *d_a0_soa, uint32_t flags,
                          double *d_a_end_soa, double *d_p_sig_end, double *d_p_sig_max,
                          int64_t *d_first_bad_step, double *d_traj_soa);

int psa_rk4_sweep_f32_dev(void *stream, int n_waves, int64_t n_points, int64_t n_steps, double z_max,
                          int32_t save_every, const float *d_dbeta, const float *d_dbeta2, const float *d_gamma,
                          const float *d_alpha, const float *d_a0_soa, uint32_t flags, float *d_a_end_soa,
                          float *d_p_sig_end, float *d_p_sig_max, int64_t *d_first_bad_step, float *d_traj_soa);

/* ---- B1': one RHS evaluation per point (host buffers, blocking) --------------------
 * Replaces yaman_model.rhs_yaman_simplified (yaman_model.py:10-52) for a batch:
 *   z [N], a_re_im [N][4][2], gamma/alpha/dbeta [N]  ->  out_re_im [N][4][2]
 * and, when non-NULL, the three terms of yaman_model.py:123-132 / :135-156 / :159-186.
 */
int psa_yaman_rhs_f64(int device, int64_t n_points, const double *z, const double *a_re_im,
                      const double *gamma, const double *alpha, const double *dbeta, double *out_re_im,
                      double *out_linear, double *out_kerr, double *out_fwm);

/* ---- gain summary over a finished sweep (host buffers, blocking) -------------------
 * Per-point reduction of scan_mismtach.py:376-389 / :723-734 and the argmax-over-sweep summary of the
 * (dead) scan_mismatch_seeded_signal (scan_mismtach.py:183-186), as one wavefront/block reduction:
 *   gain[i] = p_metric[i] / p0_sig  (linear) or 10*log10 of it (gain_db != 0);
 *             NaN if p_metric is not finite, gain <= 0, or first_bad_step[i] >= 0
 *   *best_index = argmax over finite gains (-1 if none), *best_gain its value, *n_finite their count.
 */
int psa_gain_summary_f64(int device, int64_t n_points, const double *p_metric, const int64_t *first_bad_step,
                         double p0_sig, int gain_db, double *gain_out, int64_t *best_index, double *best_gain,
                         int64_t *n_finite);

int psa_gain_summary_f64_dev(void *stream, int64_t n_points, const double *d_p_metric,
                             const int64_t *d_first_bad_step, double p0_sig, int gain_db, double *d_gain_out,
                             int64_t *d_best_index, double *d_best_gain, int64_t *d_n_finite,
                             void *d_workspace /* >= psa_gain_summary_workspace_bytes(n_points) */);
int64_t psa_gain_summary_workspace_bytes(int64_t n_points);

/* float32 sweeps (psa_rk4_sweep_f32): p_metric and the per-point gain are float, the ratio p/p0 and log10 are formed in
 * float64; best_gain stays double. */
int psa_gain_summary_f32(int device, int64_t n_points, const float *p_metric, const int64_t *first_bad_step,
                         double p0_sig, int gain_db, float *gain_out, int64_t *best_index, double *best_gain,
                         int64_t *n_finite);
int psa_gain_summary_f32_dev(void *stream, int64_t n_points, const float *d_p_metric, const int64_t *d_first_bad_step,
                             double p0_sig, int gain_db, float *d_gain_out, int64_t *d_best_index, double *d_best_gain,
                             int64_t *d_n_finite, void *d_workspace);

/* ---- the phase mismatch of a whole grid, produced on the device ------------------------------------------------
 * A multi-GPU shard generates its own dbeta slice from the grid definition instead of receiving it:
 * point i of the flattened grid lambda_p2[n2] x lambda_signal[n3] (row-major: i2 = i / n3, i3 = i % n3) gets
 *   w_j = two_pi_c / lambda_j,  w4 = (w1 + w2) - w3                     frequency_plan.plan_from_wavelengths  :291-327
 *   method PSA_DBETA_SYMMETRIC_EVEN: omega_c, omega_d, Omega as frequency_plan.infer_symmetry_from_omegas :215-255, then
 *       dbeta = sum over even_orders of beta_n (Omega^n - omega_d^n) 2/n!   dispersion.delta_beta_symmetric  :321-372
 *   method PSA_DBETA_GENERAL_TAYLOR: (beta(w3) + beta(w4)) - (beta(w1) + beta(w2)), beta(w) = sum_{n <= max_order}
 *       beta_n (w - omega_ref)^n / n!                                        dispersion.delta_beta_from_omegas :282-318
 * and dbeta = NaN (valid = 0) wherever the reference would raise for that point (wavelength <= 0, w4 <= 0, energy
 * conservation beyond atol/rtol, inconsistent symmetric plan, non-finite result) -- what its sweep drivers turn into a
 * NaN gain (scan_mismtach.py:391-392).  float64 arithmetic in the reference's operation order; the _f32 variants round
 * the float64 result to float.
 *   beta[n_beta]   beta_0 .. beta_{n_beta-1} per length unit (host pointer, n_beta <= 9; DispersionParams.get_beta_n)
 *   two_pi_c       the caller's 2*pi*c (constants.c), passed so host and device divide the same double
 *   lambda axes    device pointers (_dev) or host pointers (blocking variant); [first_index, first_index + n_points)
 *                  is the caller's block of the flattened grid.
 */
#define PSA_DBETA_SYMMETRIC_EVEN 0
#define PSA_DBETA_GENERAL_TAYLOR 1
int psa_dbeta_grid_f64_dev(void *stream, int method, const int32_t *even_orders, int n_even_orders, int max_order,
                           const double *beta, int n_beta, double omega_ref, double two_pi_c, double atol, double rtol,
                           double lambda1_m, const double *d_lambda2_axis, int64_t n2, const double *d_lambda3_axis,
                           int64_t n3, int64_t first_index, int64_t n_points, double *d_dbeta, uint8_t *d_valid_or_null);
int psa_dbeta_grid_f32_dev(void *stream, int method, const int32_t *even_orders, int n_even_orders, int max_order,
                           const double *beta, int n_beta, double omega_ref, double two_pi_c, double atol, double rtol,
                           double lambda1_m, const double *d_lambda2_axis, int64_t n2, const double *d_lambda3_axis,
                           int64_t n3, int64_t first_index, int64_t n_points, float *d_dbeta, uint8_t *d_valid_or_null);
int psa_dbeta_grid_f64(int device, int method, const int32_t *even_orders, int n_even_orders, int max_order,
                       const double *beta, int n_beta, double omega_ref, double two_pi_c, double atol, double rtol,
                       double lambda1_m, const double *lambda2_axis, int64_t n2, const double *lambda3_axis, int64_t n3,
                       int64_t first_index, int64_t n_points, double *dbeta, uint8_t *valid_or_null);

/* Six-wave grid (build-defined, scan_six_wave_grid): pair k sits at omega_c +- Omega_k and has
 * dbeta_k = delta_beta_symmetric(omega_d, Omega_k); point i of the flattened Omega1[n1] x Omega2[n2] grid gets
 * (dbeta1, dbeta2) = (dbeta(Omega1[i / n2]), dbeta(Omega2[i % n2])). */
int psa_dbeta_pairs_f64_dev(void *stream, const int32_t *even_orders, int n_even_orders, const double *beta, int n_beta,
                            double omega_d, const double *d_Omega1_axis, int64_t n1, const double *d_Omega2_axis,
                            int64_t n2, int64_t first_index, int64_t n_points, double *d_dbeta1, double *d_dbeta2);
int psa_dbeta_pairs_f32_dev(void *stream, const int32_t *even_orders, int n_even_orders, const double *beta, int n_beta,
                            double omega_d, const double *d_Omega1_axis, int64_t n1, const double *d_Omega2_axis,
                            int64_t n2, int64_t first_index, int64_t n_points, float *d_dbeta1, float *d_dbeta2);
int psa_dbeta_pairs_f64(int device, const int32_t *even_orders, int n_even_orders, const double *beta, int n_beta,
                        double omega_d, const double *Omega1_axis, int64_t n1, const double *Omega2_axis, int64_t n2,
                        int64_t first_index, int64_t n_points, double *dbeta1, double *dbeta2);

#ifdef __cplusplus
}
#endif
#endif /* PSA_RK4_H */
