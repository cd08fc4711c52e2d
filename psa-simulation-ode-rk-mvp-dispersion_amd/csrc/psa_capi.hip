// psa_capi.hip -- the extern "C" surface of libpsa_hip.so (see include/psa_rk4.h for the contract and the
// reference file:line each entry point replaces).  Host code only: argument validation, HBM staging for the
// host-buffer variants, launches.  No CPU compute path exists here on purpose: without a gfx950 device the
// host-buffer calls fail with PSA_E_DEVICE / a hipError_t -- they never fall back.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "psa_internal.h"
#include "psa_rk4.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    return (int)e;
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) { rc = hip_fail(_e, #expr); goto done; } \
    } while (0)

int validate_common(int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
                    const void *dbeta, const void *dbeta2, const void *gamma, const void *alpha, const void *a0,
                    const void *a_end, const void *p_end, const void *p_max, const void *first_bad) {
    if (n_waves != 4 && n_waves != 6) return fail(PSA_E_NWAVES, "n_waves must be 4 or 6, got %d", n_waves);
    if (n_points < 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0, got %lld", (long long)n_points);
    if (n_steps <= 0 || n_steps > 2147483647LL)
        return fail(PSA_E_NSTEPS, "n_steps must be in [1, 2^31), got %lld", (long long)n_steps);
    if (!(z_max > 0.0) || !std::isfinite(z_max)) return fail(PSA_E_ZMAX, "z_max must be positive");
    if (save_every <= 0) return fail(PSA_E_SAVE_EVERY, "save_every must be a positive integer");
    if (n_waves == 6 && !dbeta2 && n_points > 0) return fail(PSA_E_DBETA2, "n_waves == 6 requires dbeta2");
    if (n_waves == 4 && dbeta2) return fail(PSA_E_DBETA2, "dbeta2 must be NULL for n_waves == 4");
    if (n_points > 0 && (!dbeta || !gamma || !alpha || !a0 || !a_end || !p_end || !p_max || !first_bad))
        return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    return PSA_OK;
}

int check_mode(uint32_t flags) {
    if (!(flags & PSA_OPT_CHECK_NAN)) return psa::CHECK_NONE;
    return (flags & PSA_OPT_EXACT_STEP) ? psa::CHECK_EXACT : psa::CHECK_BLOCK;
}

template <typename T>
psa::SweepArgs<T> make_args(int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
                            const T *dbeta, const T *dbeta2, const T *gamma, const T *alpha, const T *a0_soa,
                            uint32_t flags, T *a_end, T *p_end, T *p_max, int64_t *first_bad, T *traj) {
    psa::SweepArgs<T> a;
    a.dbeta = dbeta;
    a.dbeta2 = dbeta2;
    a.gamma = gamma;
    a.alpha = alpha;
    a.a0 = a0_soa;
    a.a_end = a_end;
    a.p_end = p_end;
    a.p_max = p_max;
    a.first_bad = (long long *)first_bad;
    a.traj = traj;
    a.n_points = n_points;
    a.z_max = z_max;
    a.n_steps = (int)n_steps;
    a.save_every = save_every;
    a.gamma_stride = (flags & PSA_BCAST_GAMMA) ? 0 : 1;
    a.alpha_stride = (flags & PSA_BCAST_ALPHA) ? 0 : 1;
    a.a0_stride = (flags & PSA_BCAST_A0) ? 0 : 1;
    a.a0_ld = (flags & PSA_BCAST_A0) ? 1 : n_points;
    (void)n_waves;
    return a;
}

template <typename T> struct Launch;
template <> struct Launch<double> {
    static hipError_t sweep(hipStream_t s, int nw, int chk, bool lds, int blk, uint32_t flags,
                            const psa::SweepArgs<double> &a) {
        return psa::launch_sweep_f64(s, nw, chk, lds, blk, (flags & PSA_OPT_LOSSLESS) != 0, a);
    }
    static hipError_t a2s(hipStream_t s, const double *a, double *b, long long n, int nc) { return psa::launch_aos_to_soa_f64(s, a, b, n, nc); }
    static hipError_t s2a(hipStream_t s, const double *a, double *b, long long n, int nc) { return psa::launch_soa_to_aos_f64(s, a, b, n, nc); }
    static hipError_t t2a(hipStream_t s, const double *a, double *b, long long n, long long r, int nc) { return psa::launch_traj_to_aos_f64(s, a, b, n, r, nc); }
};
template <> struct Launch<float> {
    static hipError_t sweep(hipStream_t s, int nw, int chk, bool lds, int blk, uint32_t flags,
                            const psa::SweepArgs<float> &a) {
        const int pack = (flags & PSA_OPT_F32_PACKED) ? 1 : ((flags & PSA_OPT_F32_SCALAR) ? 0 : -1);
        // the packed kernel has no lossless form: with the promise given, prefer it only when packing was forced
        const bool lossless = (flags & PSA_OPT_LOSSLESS) != 0;
        return psa::launch_sweep_f32(s, nw, chk, lds, blk, pack, lossless && pack == 0, a);
    }
    static hipError_t a2s(hipStream_t s, const float *a, float *b, long long n, int nc) { return psa::launch_aos_to_soa_f32(s, a, b, n, nc); }
    static hipError_t s2a(hipStream_t s, const float *a, float *b, long long n, int nc) { return psa::launch_soa_to_aos_f32(s, a, b, n, nc); }
    static hipError_t t2a(hipStream_t s, const float *a, float *b, long long n, long long r, int nc) { return psa::launch_traj_to_aos_f32(s, a, b, n, r, nc); }
};

template <typename T>
int sweep_dev(void *stream, int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
              const T *d_dbeta, const T *d_dbeta2, const T *d_gamma, const T *d_alpha, const T *d_a0_soa,
              uint32_t flags, T *d_a_end_soa, T *d_p_end, T *d_p_max, int64_t *d_first_bad, T *d_traj_soa) {
    int rc = validate_common(n_waves, n_points, n_steps, z_max, save_every, d_dbeta, d_dbeta2, d_gamma, d_alpha,
                             d_a0_soa, d_a_end_soa, d_p_end, d_p_max, d_first_bad);
    if (rc != PSA_OK) return rc;
    if (n_points == 0) return PSA_OK;
    auto a = make_args<T>(n_waves, n_points, n_steps, z_max, save_every, d_dbeta, d_dbeta2, d_gamma, d_alpha,
                          d_a0_soa, flags, d_a_end_soa, d_p_end, d_p_max, d_first_bad, d_traj_soa);
    hipError_t e = Launch<T>::sweep((hipStream_t)stream, n_waves, check_mode(flags), (flags & PSA_OPT_LDS_STAGING) != 0,
                                    (flags & PSA_OPT_BLOCK64) ? 64 : 256, flags, a);
    if (e != hipSuccess) return hip_fail(e, "rk4_sweep launch");
    return PSA_OK;
}

// Device scratch that frees itself on every exit path of the host-buffer entry points.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <typename U> U *as() { return (U *)p; }
};

template <typename T>
int sweep_host(int device, int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
               const T *dbeta, const T *dbeta2, const T *gamma, const T *alpha, const T *a0, uint32_t flags,
               T *a_end, T *p_end, T *p_max, int64_t *first_bad, T *traj, double *elapsed_ms) {
    int rc = validate_common(n_waves, n_points, n_steps, z_max, save_every, dbeta, dbeta2, gamma, alpha, a0, a_end,
                             p_end, p_max, first_bad);
    if (rc != PSA_OK) return rc;
    if (elapsed_ms) *elapsed_ms = 0.0;
    if (n_points == 0) return PSA_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PSA_E_DEVICE, "no HIP device visible: the RK4 sweep has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(PSA_E_DEVICE, "device %d out of range [0, %d)", device, ndev);

    const int nc = 2 * n_waves;
    const size_t N = (size_t)n_points;
    const int64_t n_saved = n_steps / save_every + 1;
    size_t traj_elems = 0;
    if (traj) {
        // N * n_saved * nc must fit comfortably in int64 / size_t
        const long double te = (long double)N * (long double)n_saved * (long double)nc;
        if (te > 4.0e18L) return fail(PSA_E_TOO_LARGE, "trajectory buffer too large");
        traj_elems = N * (size_t)n_saved * (size_t)nc;
    }
    if ((flags & PSA_BCAST_ALPHA) && alpha[0] == T(0)) flags |= PSA_OPT_LOSSLESS;   // the reference's alpha == 0.0 branch
    const size_t n_gamma = (flags & PSA_BCAST_GAMMA) ? 1 : N;
    const size_t n_alpha = (flags & PSA_BCAST_ALPHA) ? 1 : N;
    const size_t n_a0 = (flags & PSA_BCAST_A0) ? 1 : N;

    DevBuf b_dbeta, b_dbeta2, b_gamma, b_alpha, b_a0_aos, b_a0_soa, b_aend_soa, b_aend_aos, b_pend, b_pmax, b_bad,
        b_traj_soa, b_traj_aos;
    hipStream_t st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreate(&st));
    HIP_TRY(b_dbeta.alloc(N * sizeof(T)));
    if (dbeta2) HIP_TRY(b_dbeta2.alloc(N * sizeof(T)));
    HIP_TRY(b_gamma.alloc(n_gamma * sizeof(T)));
    HIP_TRY(b_alpha.alloc(n_alpha * sizeof(T)));
    HIP_TRY(b_a0_aos.alloc(n_a0 * nc * sizeof(T)));
    HIP_TRY(b_a0_soa.alloc(n_a0 * nc * sizeof(T)));
    HIP_TRY(b_aend_soa.alloc(N * nc * sizeof(T)));
    HIP_TRY(b_aend_aos.alloc(N * nc * sizeof(T)));
    HIP_TRY(b_pend.alloc(N * sizeof(T)));
    HIP_TRY(b_pmax.alloc(N * sizeof(T)));
    HIP_TRY(b_bad.alloc(N * sizeof(int64_t)));
    if (traj) {
        HIP_TRY(b_traj_soa.alloc(traj_elems * sizeof(T)));
        if (N > 1) HIP_TRY(b_traj_aos.alloc(traj_elems * sizeof(T)));
    }
    HIP_TRY(hipMemcpyAsync(b_dbeta.p, dbeta, N * sizeof(T), hipMemcpyHostToDevice, st));
    if (dbeta2) HIP_TRY(hipMemcpyAsync(b_dbeta2.p, dbeta2, N * sizeof(T), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_gamma.p, gamma, n_gamma * sizeof(T), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_alpha.p, alpha, n_alpha * sizeof(T), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b_a0_aos.p, a0, n_a0 * nc * sizeof(T), hipMemcpyHostToDevice, st));
    HIP_TRY(Launch<T>::a2s(st, b_a0_aos.as<T>(), b_a0_soa.as<T>(), (long long)n_a0, nc));
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, st));
    rc = sweep_dev<T>(st, n_waves, n_points, n_steps, z_max, save_every, b_dbeta.as<T>(),
                      dbeta2 ? b_dbeta2.as<T>() : nullptr, b_gamma.as<T>(), b_alpha.as<T>(), b_a0_soa.as<T>(), flags,
                      b_aend_soa.as<T>(), b_pend.as<T>(), b_pmax.as<T>(), b_bad.as<int64_t>(),
                      traj ? b_traj_soa.as<T>() : nullptr);
    if (rc != PSA_OK) goto done;
    HIP_TRY(hipEventRecord(ev1, st));
    HIP_TRY(Launch<T>::s2a(st, b_aend_soa.as<T>(), b_aend_aos.as<T>(), (long long)N, nc));
    HIP_TRY(hipMemcpyAsync(a_end, b_aend_aos.p, N * nc * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(p_end, b_pend.p, N * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(p_max, b_pmax.p, N * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(first_bad, b_bad.p, N * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (traj) {
        if (N > 1) {  // [rows][nc][N] -> [N][rows][nc]; for N == 1 the two layouts coincide
            HIP_TRY(Launch<T>::t2a(st, b_traj_soa.as<T>(), b_traj_aos.as<T>(), (long long)N, (long long)n_saved, nc));
            HIP_TRY(hipMemcpyAsync(traj, b_traj_aos.p, traj_elems * sizeof(T), hipMemcpyDeviceToHost, st));
        } else {
            HIP_TRY(hipMemcpyAsync(traj, b_traj_soa.p, traj_elems * sizeof(T), hipMemcpyDeviceToHost, st));
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (elapsed_ms) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        *elapsed_ms = (double)ms;
    }
done:
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (st) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    return rc;
}

}  // namespace

extern "C" {

int psa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char *psa_last_error(void) { return g_err; }
const char *psa_version(void) { return "psa-hip 0.1.0 gfx950"; }
int64_t psa_n_saved(int64_t n_steps, int32_t save_every) {
    if (n_steps < 0 || save_every <= 0) return -1;
    return n_steps / save_every + 1;
}

int psa_rk4_sweep_f64(int device, int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
                      const double *dbeta, const double *dbeta2, const double *gamma, const double *alpha,
                      const double *a0, uint32_t flags, double *a_end, double *p_end, double *p_max,
                      int64_t *first_bad, double *traj, double *elapsed_ms) {
    return sweep_host<double>(device, n_waves, n_points, n_steps, z_max, save_every, dbeta, dbeta2, gamma, alpha, a0,
                              flags, a_end, p_end, p_max, first_bad, traj, elapsed_ms);
}
int psa_rk4_sweep_f32(int device, int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
                      const float *dbeta, const float *dbeta2, const float *gamma, const float *alpha, const float *a0,
                      uint32_t flags, float *a_end, float *p_end, float *p_max, int64_t *first_bad, float *traj,
                      double *elapsed_ms) {
    return sweep_host<float>(device, n_waves, n_points, n_steps, z_max, save_every, dbeta, dbeta2, gamma, alpha, a0,
                             flags, a_end, p_end, p_max, first_bad, traj, elapsed_ms);
}
int psa_rk4_sweep_f64_dev(void *stream, int n_waves, int64_t n_points, int64_t n_steps, double z_max,
                          int32_t save_every, const double *d_dbeta, const double *d_dbeta2, const double *d_gamma,
                          const double *d_alpha, const double *d_a0_soa, uint32_t flags, double *d_a_end_soa,
                          double *d_p_end, double *d_p_max, int64_t *d_first_bad, double *d_traj_soa) {
    return sweep_dev<double>(stream, n_waves, n_points, n_steps, z_max, save_every, d_dbeta, d_dbeta2, d_gamma,
                             d_alpha, d_a0_soa, flags, d_a_end_soa, d_p_end, d_p_max, d_first_bad, d_traj_soa);
}
int psa_rk4_sweep_f32_dev(void *stream, int n_waves, int64_t n_points, int64_t n_steps, double z_max,
                          int32_t save_every, const float *d_dbeta, const float *d_dbeta2, const float *d_gamma,
                          const float *d_alpha, const float *d_a0_soa, uint32_t flags, float *d_a_end_soa,
                          float *d_p_end, float *d_p_max, int64_t *d_first_bad, float *d_traj_soa) {
    return sweep_dev<float>(stream, n_waves, n_points, n_steps, z_max, save_every, d_dbeta, d_dbeta2, d_gamma, d_alpha,
                            d_a0_soa, flags, d_a_end_soa, d_p_end, d_p_max, d_first_bad, d_traj_soa);
}

int psa_yaman_rhs_f64(int device, int64_t n, const double *z, const double *a, const double *gamma,
                      const double *alpha, const double *dbeta, double *out, double *out_lin, double *out_kerr,
                      double *out_fwm) {
    if (n < 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0");
    if (n == 0) return PSA_OK;
    if (!z || !a || !gamma || !alpha || !dbeta || !out) return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PSA_E_DEVICE, "no HIP device visible: the RHS kernel has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(PSA_E_DEVICE, "device %d out of range [0, %d)", device, ndev);
    int rc = PSA_OK;
    const size_t N = (size_t)n;
    DevBuf bz, ba, bg, bal, bd, bo, bl, bk, bf;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(bz.alloc(N * 8)); HIP_TRY(ba.alloc(N * 64)); HIP_TRY(bg.alloc(N * 8)); HIP_TRY(bal.alloc(N * 8));
    HIP_TRY(bd.alloc(N * 8)); HIP_TRY(bo.alloc(N * 64));
    if (out_lin) HIP_TRY(bl.alloc(N * 64));
    if (out_kerr) HIP_TRY(bk.alloc(N * 64));
    if (out_fwm) HIP_TRY(bf.alloc(N * 64));
    HIP_TRY(hipMemcpy(bz.p, z, N * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ba.p, a, N * 64, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(bg.p, gamma, N * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(bal.p, alpha, N * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(bd.p, dbeta, N * 8, hipMemcpyHostToDevice));
    HIP_TRY(psa::launch_yaman_rhs_f64(nullptr, n, bz.as<double>(), ba.as<double>(), bg.as<double>(), bal.as<double>(),
                                      bd.as<double>(), bo.as<double>(), out_lin ? bl.as<double>() : nullptr,
                                      out_kerr ? bk.as<double>() : nullptr, out_fwm ? bf.as<double>() : nullptr));
    HIP_TRY(hipMemcpy(out, bo.p, N * 64, hipMemcpyDeviceToHost));
    if (out_lin) HIP_TRY(hipMemcpy(out_lin, bl.p, N * 64, hipMemcpyDeviceToHost));
    if (out_kerr) HIP_TRY(hipMemcpy(out_kerr, bk.p, N * 64, hipMemcpyDeviceToHost));
    if (out_fwm) HIP_TRY(hipMemcpy(out_fwm, bf.p, N * 64, hipMemcpyDeviceToHost));
done:
    return rc;
}

int64_t psa_gain_summary_workspace_bytes(int64_t n) { return psa::gain_summary_workspace_bytes(n); }

int psa_gain_summary_f64_dev(void *stream, int64_t n, const double *d_p, const int64_t *d_bad, double p0_sig,
                             int gain_db, double *d_gain, int64_t *d_best_i, double *d_best_g, int64_t *d_nfin,
                             void *d_ws) {
    if (n < 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0");
    if (!d_best_i || !d_best_g || !d_nfin || !d_ws || (n > 0 && !d_p))
        return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    hipError_t e = psa::launch_gain_summary_f64((hipStream_t)stream, n, d_p, (const long long *)d_bad, p0_sig, gain_db,
                                                d_gain, (long long *)d_best_i, d_best_g, (long long *)d_nfin, d_ws);
    if (e != hipSuccess) return hip_fail(e, "gain_summary launch");
    return PSA_OK;
}

int psa_gain_summary_f64(int device, int64_t n, const double *p_metric, const int64_t *first_bad, double p0_sig,
                         int gain_db, double *gain_out, int64_t *best_index, double *best_gain, int64_t *n_finite) {
    if (n < 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0");
    if (!best_index || !best_gain || !n_finite || (n > 0 && !p_metric))
        return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PSA_E_DEVICE, "no HIP device visible: the gain summary has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(PSA_E_DEVICE, "device %d out of range [0, %d)", device, ndev);
    int rc = PSA_OK;
    const size_t N = (size_t)n;
    DevBuf bp, bb, bg, bi, bbg, bn, bw;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(bp.alloc(N * 8));
    if (first_bad) HIP_TRY(bb.alloc(N * 8));
    if (gain_out) HIP_TRY(bg.alloc(N * 8));
    HIP_TRY(bi.alloc(8)); HIP_TRY(bbg.alloc(8)); HIP_TRY(bn.alloc(8));
    HIP_TRY(bw.alloc((size_t)psa::gain_summary_workspace_bytes(n)));
    if (N) HIP_TRY(hipMemcpy(bp.p, p_metric, N * 8, hipMemcpyHostToDevice));
    if (first_bad && N) HIP_TRY(hipMemcpy(bb.p, first_bad, N * 8, hipMemcpyHostToDevice));
    rc = psa_gain_summary_f64_dev(nullptr, n, bp.as<double>(), first_bad ? bb.as<int64_t>() : nullptr, p0_sig, gain_db,
                                  gain_out ? bg.as<double>() : nullptr, bi.as<int64_t>(), bbg.as<double>(),
                                  bn.as<int64_t>(), bw.p);
    if (rc != PSA_OK) goto done;
    if (gain_out && N) HIP_TRY(hipMemcpy(gain_out, bg.p, N * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(best_index, bi.p, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(best_gain, bbg.p, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(n_finite, bn.p, 8, hipMemcpyDeviceToHost));
done:
    return rc;
}

}  // extern "C"
