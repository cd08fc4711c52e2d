/*
 * psa_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, one sweep point at a time, complex double) of
 * the reference's RK4 / Agrawal-Yaman hot path.  It is the CHECKER for the HIP
 * kernels: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  Nothing under psa-simulation-ode-rk-mvp-dispersion_amd/
 * imports, links or calls it.
 *
 * Parity status: PINNED.  Checked against golden vectors G1-G9, G11 produced
 * by importing the reference itself (tests/golden/gen_golden.py) -- see
 * tests/test_oracle_golden.py.  Agreement is at the level of libm-vs-NumPy
 * ulp differences (<= 1e-13 relative on A_end at 45 dB gain).
 *
 * Every function cites the reference file:line it restates (paths are into
 * the upstream repo Alxkov/PSA-simulation-ODE-RK-MVP-Dispersion).  Evaluation
 * ORDER of the floating-point operations follows the reference expression by
 * expression; compile with -ffp-contract=off so nothing is fused.
 *
 * The 6-wave functions at the end are a BUILD-DEFINED extension (the reference
 * has no 6-wave model): "parity unpinned" beyond the reduction property that
 * with the second sideband pair at zero amplitude they reproduce the 4-wave
 * system exactly.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cplx;

/* np.abs(x) ** 2 : npy_cabs -> hypot, then squared (yaman_model.py:143-146) */
static inline double abs2_np(cplx a) {
    double m = hypot(creal(a), cimag(a));
    return m * m;
}

/* complex * complex exactly as NumPy's loop does it (no Annex-G recovery):
 * (ar*br - ai*bi) + i (ar*bi + ai*br)                                      */
static inline cplx cmul(cplx a, cplx b) {
    double ar = creal(a), ai = cimag(a), br = creal(b), bi = cimag(b);
    return CMPLX(ar * br - ai * bi, ar * bi + ai * br);
}

/* real-scalar * complex: NumPy promotes the scalar to (s + 0j) and multiplies */
static inline cplx rmul(double s, cplx a) { return cmul(CMPLX(s, 0.0), a); }

/* np.exp(1j * dbeta * z): argument (0 + i*dbeta*z); npy_cexp -> exp(0)*(cos + i sin) */
static inline cplx exp_i(double phase) { return CMPLX(cos(phase), sin(phase)); }

/* ---- yaman_model.py:123-132  _linear_loss_terms --------------------------- */
static void linear_loss_terms(const cplx a[4], double alpha, cplx out[4]) {
    if (alpha == 0.0) {
        for (int j = 0; j < 4; ++j) out[j] = 0.0;
        return;
    }
    double c = -0.5 * alpha;
    for (int j = 0; j < 4; ++j) out[j] = rmul(c, a[j]);
}

/* ---- yaman_model.py:135-156  _kerr_terms ---------------------------------- */
static void kerr_terms(const cplx a[4], double gamma, cplx out[4]) {
    double p1 = abs2_np(a[0]), p2 = abs2_np(a[1]), ps = abs2_np(a[2]), pi = abs2_np(a[3]);
    double f1 = p1 + 2.0 * (p2 + ps + pi);
    double f2 = p2 + 2.0 * (p1 + ps + pi);
    double fs = ps + 2.0 * (p1 + p2 + pi);
    double fi = pi + 2.0 * (p1 + p2 + ps);
    cplx ig = CMPLX(0.0, gamma); /* 1j * gamma */
    out[0] = cmul(ig, rmul(f1, a[0]));
    out[1] = cmul(ig, rmul(f2, a[1]));
    out[2] = cmul(ig, rmul(fs, a[2]));
    out[3] = cmul(ig, rmul(fi, a[3]));
}

/* ---- yaman_model.py:159-186  _fwm_terms ----------------------------------- */
static void fwm_terms(double z, const cplx a[4], double gamma, double dbeta, cplx out[4]) {
    cplx ph_p = exp_i(dbeta * z);    /* np.exp(+1j*dbeta*z) */
    cplx ph_s = exp_i(-dbeta * z);   /* np.exp(-1j*dbeta*z): (-1j*dbeta)*z */
    cplx t1 = cmul(ph_p, cmul(cmul(conj(a[1]), a[2]), a[3]));
    cplx t2 = cmul(ph_p, cmul(cmul(conj(a[0]), a[2]), a[3]));
    cplx t3 = cmul(ph_s, cmul(cmul(conj(a[3]), a[0]), a[1]));
    cplx t4 = cmul(ph_s, cmul(cmul(conj(a[2]), a[0]), a[1]));
    cplx c = CMPLX(0.0, gamma * 2.0); /* 1j * gamma * 2.0 */
    out[0] = cmul(c, t1);
    out[1] = cmul(c, t2);
    out[2] = cmul(c, t3);
    out[3] = cmul(c, t4);
}

/* ---- yaman_model.py:10-52  rhs_yaman_simplified --------------------------- */
static void rhs4(double z, const cplx a[4], double gamma, double alpha, double dbeta, cplx out[4]) {
    cplx lin[4], ker[4], fwm[4];
    linear_loss_terms(a, alpha, lin);
    kerr_terms(a, gamma, ker);
    fwm_terms(z, a, gamma, dbeta, fwm);
    for (int j = 0; j < 4; ++j) out[j] = (lin[j] + ker[j]) + fwm[j];
}

/* Exposed for the G5 direct-RHS fixture: total and the three terms.
 * a, out*: [4][2] doubles (numpy complex128 layout).                         */
void psa_oracle_rhs4(double z, const double *a, double gamma, double alpha, double dbeta,
                     double *out, double *out_lin, double *out_kerr, double *out_fwm) {
    cplx A[4], r[4], l[4], k[4], f[4];
    for (int j = 0; j < 4; ++j) A[j] = CMPLX(a[2 * j], a[2 * j + 1]);
    rhs4(z, A, gamma, alpha, dbeta, r);
    linear_loss_terms(A, alpha, l);
    kerr_terms(A, gamma, k);
    fwm_terms(z, A, gamma, dbeta, f);
    for (int j = 0; j < 4; ++j) {
        out[2 * j] = creal(r[j]); out[2 * j + 1] = cimag(r[j]);
        if (out_lin)  { out_lin[2 * j] = creal(l[j]);  out_lin[2 * j + 1] = cimag(l[j]); }
        if (out_kerr) { out_kerr[2 * j] = creal(k[j]); out_kerr[2 * j + 1] = cimag(k[j]); }
        if (out_fwm)  { out_fwm[2 * j] = creal(f[j]);  out_fwm[2 * j + 1] = cimag(f[j]); }
    }
}

/* ---- build-defined 6-wave RHS (no reference counterpart; see header) ------
 * Waves [p1, p2, s1, i1, s2, i2]; both sideband pairs satisfy
 * w_s + w_i = w_p1 + w_p2 with their own mismatch dbeta1 / dbeta2.
 * Same conventions as the 4-wave model: SPM weight 1, XPM weight 2,
 * FWM prefactor 2*i*gamma, pumps see exp(+i dbeta z), sidebands exp(-i dbeta z).
 */
static void rhs6(double z, const cplx a[6], double gamma, double alpha, double dbeta1, double dbeta2,
                 cplx out[6]) {
    double p[6], tot = 0.0;
    for (int j = 0; j < 6; ++j) { p[j] = abs2_np(a[j]); }
    for (int j = 0; j < 6; ++j) tot += p[j];
    cplx e1p = exp_i(dbeta1 * z), e1s = exp_i(-dbeta1 * z);
    cplx e2p = exp_i(dbeta2 * z), e2s = exp_i(-dbeta2 * z);
    cplx q12 = cmul(a[0], a[1]);
    cplx q34 = cmul(a[2], a[3]);
    cplx q56 = cmul(a[4], a[5]);
    cplx fw[6];
    fw[0] = cmul(conj(a[1]), cmul(e1p, q34) + cmul(e2p, q56));
    fw[1] = cmul(conj(a[0]), cmul(e1p, q34) + cmul(e2p, q56));
    fw[2] = cmul(conj(a[3]), cmul(e1s, q12));
    fw[3] = cmul(conj(a[2]), cmul(e1s, q12));
    fw[4] = cmul(conj(a[5]), cmul(e2s, q12));
    fw[5] = cmul(conj(a[4]), cmul(e2s, q12));
    cplx ig = CMPLX(0.0, gamma), ig2 = CMPLX(0.0, gamma * 2.0);
    for (int j = 0; j < 6; ++j) {
        double f = p[j] + 2.0 * (tot - p[j]);
        cplx lin = (alpha == 0.0) ? 0.0 : rmul(-0.5 * alpha, a[j]);
        out[j] = (lin + cmul(ig, rmul(f, a[j]))) + cmul(ig2, fw[j]);
    }
}

typedef struct {
    int n_waves;
    double gamma, alpha, dbeta, dbeta2;
} rhs_par;

static inline void rhs_any(double z, const cplx *a, const rhs_par *p, cplx *out) {
    if (p->n_waves == 4) rhs4(z, a, p->gamma, p->alpha, p->dbeta, out);
    else rhs6(z, a, p->gamma, p->alpha, p->dbeta, p->dbeta2, out);
}

/* ---- integrators.py:25-61  rk4_step --------------------------------------- */
static void rk4_step(double z, const cplx *y, double dz, const rhs_par *p, cplx *y_next) {
    const int n = p->n_waves;
    cplx k1[6], k2[6], k3[6], k4[6], t[6];
    double hdz = 0.5 * dz;
    rhs_any(z, y, p, k1);
    for (int j = 0; j < n; ++j) t[j] = y[j] + rmul(hdz, k1[j]);
    rhs_any(z + 0.5 * dz, t, p, k2);
    for (int j = 0; j < n; ++j) t[j] = y[j] + rmul(hdz, k2[j]);
    rhs_any(z + 0.5 * dz, t, p, k3);
    for (int j = 0; j < n; ++j) t[j] = y[j] + rmul(dz, k3[j]);
    rhs_any(z + dz, t, p, k4);
    double c6 = dz / 6.0;
    for (int j = 0; j < n; ++j) {
        cplx s = ((k1[j] + rmul(2.0, k2[j])) + rmul(2.0, k3[j])) + k4[j];
        y_next[j] = y[j] + rmul(c6, s);
    }
}

/* ---- integrators.py:194  n_steps = int(round(z_max / dz)) -----------------
 * Python round() is round-half-to-even == rint() in the default FP mode.     */
int64_t psa_oracle_n_steps(double z_max, double dz) { return (int64_t)rint(z_max / dz); }

/* np.linspace(0, z_max, n+1)[i] (integrators.py:195): i*step, last forced to stop */
static inline double zgrid(int64_t i, int64_t n, double z_max, double step) {
    return (i == n) ? z_max : (double)i * step;
}

/* ---- integrators.py:68-142 + 150-204: integrate_interval ------------------
 * One point.  y_out is [n_saved][n_waves][2] or NULL (summary only).
 * Returns first_bad_step (-1 if finite or check_nan == 0).  When check_nan
 * fires the reference raises, so integration stops there and *n_rows_written
 * is the number of rows saved before the raise.
 * Summary outputs (computed from the SAVED rows only, as the sweep drivers do,
 * scan_mismtach.py:376-381): p_end = |A[-1, sig]|^2, p_max = max_rows |A[:, sig]|^2.
 */
static int64_t integrate_point(const rhs_par *par, double z_max, int64_t n_steps, int64_t save_every,
                               int check_nan, const cplx *a0, double *z_out, double *y_out,
                               int64_t *n_rows_written, cplx *a_end, double *p_end, double *p_max) {
    const int nw = par->n_waves;
    const int sig = 2;
    double step = z_max / (double)n_steps; /* linspace step = (stop-start)/div */
    cplx y[6], yn[6];
    memcpy(y, a0, sizeof(cplx) * nw);
    int64_t rows = 0;
    if (z_out) z_out[0] = 0.0;
    if (y_out) for (int j = 0; j < nw; ++j) { y_out[2 * j] = creal(y[j]); y_out[2 * j + 1] = cimag(y[j]); }
    rows = 1;
    memcpy(a_end, y, sizeof(cplx) * nw);
    double pm = abs2_np(y[sig]);
    double pe = pm;
    int64_t bad = -1;
    for (int64_t i = 0; i < n_steps; ++i) {
        double z = zgrid(i, n_steps, z_max, step);
        double dz = zgrid(i + 1, n_steps, z_max, step) - z; /* integrators.py:128 */
        rk4_step(z, y, dz, par, yn);
        memcpy(y, yn, sizeof(cplx) * nw);
        if (check_nan) {
            int fin = 1;
            for (int j = 0; j < nw; ++j) fin &= isfinite(creal(y[j])) && isfinite(cimag(y[j]));
            if (!fin) { bad = i; break; } /* FloatingPointError, integrators.py:132-135 */
        }
        if ((i + 1) % save_every == 0) {
            if (z_out) z_out[rows] = zgrid(i + 1, n_steps, z_max, step);
            if (y_out) for (int j = 0; j < nw; ++j) {
                y_out[(rows * nw + j) * 2] = creal(y[j]);
                y_out[(rows * nw + j) * 2 + 1] = cimag(y[j]);
            }
            rows++;
            memcpy(a_end, y, sizeof(cplx) * nw);
            pe = abs2_np(y[sig]);
            /* np.max propagates NaN */
            if (isnan(pe) || isnan(pm)) pm = NAN; else if (pe > pm) pm = pe;
        }
    }
    if (n_rows_written) *n_rows_written = rows;
    *p_end = pe;
    *p_max = pm;
    return bad;
}

/* Single point with trajectory: a0 [nw][2]; z_out [n_saved]; y_out [n_saved][nw][2],
 * n_saved = n_steps / save_every + 1 (integrators.py:115).                   */
int64_t psa_oracle_integrate(int n_waves, double z_max, int64_t n_steps, int64_t save_every, int check_nan,
                             double gamma, double alpha, double dbeta, double dbeta2, const double *a0,
                             double *z_out, double *y_out, int64_t *n_rows_written) {
    rhs_par par = {n_waves, gamma, alpha, dbeta, dbeta2};
    cplx A0[6], a_end[6];
    double pe, pm;
    for (int j = 0; j < n_waves; ++j) A0[j] = CMPLX(a0[2 * j], a0[2 * j + 1]);
    return integrate_point(&par, z_max, n_steps, save_every, check_nan, A0, z_out, y_out, n_rows_written,
                           a_end, &pe, &pm);
}

/* Sweep (scan_mismtach.py:357-392 / 694-738 loop body minus plotting): N independent
 * points.  Per-point arrays have stride 1 (per point) or 0 (broadcast scalar).
 * a0: [N or 1][nw][2].  a_end: [N][nw][2].  OpenMP over points when n_threads > 1. */
int psa_oracle_sweep(int n_waves, int64_t n_points, double z_max, int64_t n_steps, int64_t save_every,
                     int check_nan, const double *dbeta, const double *dbeta2, const double *gamma,
                     int gamma_stride, const double *alpha, int alpha_stride, const double *a0, int a0_stride,
                     double *a_end, double *p_end, double *p_max, int64_t *first_bad_step, int n_threads) {
    if ((n_waves != 4 && n_waves != 6) || n_points < 0 || n_steps <= 0 || save_every <= 0) return -1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int64_t p = 0; p < n_points; ++p) {
        rhs_par par = {n_waves, gamma[p * gamma_stride], alpha[p * alpha_stride], dbeta[p],
                       dbeta2 ? dbeta2[p] : 0.0};
        cplx A0[6], ae[6];
        const double *src = a0 + (size_t)p * a0_stride * n_waves * 2;
        for (int j = 0; j < n_waves; ++j) A0[j] = CMPLX(src[2 * j], src[2 * j + 1]);
        double pe, pm;
        int64_t bad = integrate_point(&par, z_max, n_steps, save_every, check_nan, A0, NULL, NULL, NULL, ae,
                                      &pe, &pm);
        for (int j = 0; j < n_waves; ++j) {
            a_end[((size_t)p * n_waves + j) * 2] = creal(ae[j]);
            a_end[((size_t)p * n_waves + j) * 2 + 1] = cimag(ae[j]);
        }
        p_end[p] = pe;
        p_max[p] = pm;
        first_bad_step[p] = bad;
    }
    (void)n_threads;
    return 0;
}

int psa_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
