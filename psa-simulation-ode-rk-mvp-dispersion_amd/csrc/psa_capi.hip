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
    // one 64-thread workgroup per wave at the finest launch shape; the grid's x extent is a 32-bit count
    if (n_points > 64LL * 2147483647LL) return fail(PSA_E_TOO_LARGE, "n_points %lld exceeds the launch grid limit", (long long)n_points);
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
        const int split = (flags & PSA_OPT_SPLIT_POINT) ? 1 : ((flags & PSA_OPT_ONE_LANE) ? 0 : -1);
        return psa::launch_sweep_f64(s, nw, chk, lds, blk, (flags & PSA_OPT_LOSSLESS) != 0, split, a);
    }
    static hipError_t a2s(hipStream_t s, const double *a, double *b, long long n, int nc) { return psa::launch_aos_to_soa_f64(s, a, b, n, nc); }
    static hipError_t s2a(hipStream_t s, const double *a, double *b, long long n, int nc) { return psa::launch_soa_to_aos_f64(s, a, b, n, nc); }
    static hipError_t t2a(hipStream_t s, const double *a, double *b, long long n, long long ld, long long r, int nc) { return psa::launch_traj_to_aos_f64(s, a, b, n, ld, r, nc); }
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
    static hipError_t t2a(hipStream_t s, const float *a, float *b, long long n, long long ld, long long r, int nc) { return psa::launch_traj_to_aos_f32(s, a, b, n, ld, r, nc); }
};

template <typename T>
int sweep_dev(void *stream, int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
              const T *d_dbeta, const T *d_dbeta2, const T *d_gamma, const T *d_alpha, const T *d_a0_soa,
              uint32_t flags, T *d_a_end_soa, T *d_p_end, T *d_p_max, int64_t *d_first_bad, T *d_traj_soa) {
    int rc = validate_common(n_waves, n_points, n_steps, z_max, save_every, d_dbeta, d_dbeta2, d_gamma, d_alpha,
                             d_a0_soa, d_a_end_soa, d_p_end, d_p_max, d_first_bad);
    if (rc != PSA_OK) return rc;
    if (n_points == 0) return PSA_OK;
    // trajectory rows are addressed as (row, wave) base + a 32-bit byte offset per lane
    if (d_traj_soa && (unsigned long long)n_points * (2 * sizeof(T)) >= (1ull << 32))
        return fail(PSA_E_TOO_LARGE, "a trajectory launch takes at most %llu points", (1ull << 32) / (2 * sizeof(T)) - 1);
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

// The dozen small per-sweep buffers of the host-buffer API come out of ONE allocation (256-B aligned slices): a single-point
// run (BASELINE config 1) otherwise spends as long in hipMalloc / hipFree as a tenth of its kernel.
struct DevArena {
    DevBuf buf;
    size_t used = 0, cap = 0;
    static size_t aligned(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    hipError_t reserve(size_t bytes) { cap = bytes; return buf.alloc(bytes); }
    template <typename U> U *take(size_t count) {
        U *p = (U *)((char *)buf.p + used);
        used += aligned(count * sizeof(U));
        return p;
    }
};

// Makes `device` current for the scope of a host-buffer entry point and puts the caller's device back afterwards
// (a host-API call must not leave a side effect on a thread that also drives torch or another HIP library).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int device) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev == device) return hipSuccess;
        e = hipSetDevice(device);
        switched = (e == hipSuccess);
        return e;
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

int check_device(int device, const char *what) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PSA_E_DEVICE, "no HIP device visible: %s has no CPU fallback", what);
    if (device < 0 || device >= ndev) return fail(PSA_E_DEVICE, "device %d out of range [0, %d)", device, ndev);
    return PSA_OK;
}

// staging for the trajectory transpose of the host-buffer API: two buffers of at most this many bytes each
constexpr size_t TRAJ_STAGE_BYTES = 256u << 20;

template <typename T>
int sweep_host(int device, int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
               const T *dbeta, const T *dbeta2, const T *gamma, const T *alpha, const T *a0, uint32_t flags,
               T *a_end, T *p_end, T *p_max, int64_t *first_bad, T *traj, double *elapsed_ms) {
    int rc = validate_common(n_waves, n_points, n_steps, z_max, save_every, dbeta, dbeta2, gamma, alpha, a0, a_end,
                             p_end, p_max, first_bad);
    if (rc != PSA_OK) return rc;
    if (elapsed_ms) *elapsed_ms = 0.0;
    if (n_points == 0) return PSA_OK;
    rc = check_device(device, "the RK4 sweep");
    if (rc != PSA_OK) return rc;

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

    // trajectory rows leave the device in chunks of points: [rows][nw][N] -> chunk [pts][rows][nw] in a bounded staging
    // buffer (two of them, so a chunk's device-to-host copy overlaps the next chunk's transpose)
    const size_t point_bytes = (size_t)n_saved * nc * sizeof(T);          // one point's whole trajectory
    size_t chunk_pts = 0;
    if (traj && N > 1) {
        chunk_pts = (TRAJ_STAGE_BYTES / point_bytes) / 32 * 32;            // whole 32-point transpose tiles
        if (chunk_pts < 32) chunk_pts = 32;                                // (very long single runs: one tile per chunk)
        if (chunk_pts > N) chunk_pts = N;
    }

    DeviceScope scope;
    DevArena arena;
    DevBuf b_traj_soa, b_stage[2];
    T *d_dbeta = nullptr, *d_dbeta2 = nullptr, *d_gamma = nullptr, *d_alpha = nullptr, *d_a0_aos = nullptr, *d_a0_soa = nullptr,
      *d_aend_soa = nullptr, *d_aend_aos = nullptr, *d_pend = nullptr, *d_pmax = nullptr;
    int64_t *d_bad = nullptr;
    hipStream_t st = nullptr, st_copy[2] = {nullptr, nullptr};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_kernel = nullptr;

    HIP_TRY(scope.enter(device));
    if (traj) {   // say "too large" before hipMalloc says "out of memory"
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const long double need = (long double)traj_elems * sizeof(T) + 2.0L * (long double)chunk_pts * point_bytes
                                 + (long double)N * (3 * nc + 4) * sizeof(T);
        if (need > (long double)free_b) {
            rc = fail(PSA_E_TOO_LARGE, "trajectory of %.3g GB does not fit the %.3g GB free on device %d",
                      (double)(need / 1e9L), (double)free_b / 1e9, device);
            goto done;
        }
    }
    HIP_TRY(hipStreamCreate(&st));
    {
        using A = DevArena;
        const size_t total = A::aligned(N * sizeof(T)) * (dbeta2 ? 2 : 1) + A::aligned(n_gamma * sizeof(T)) +
                             A::aligned(n_alpha * sizeof(T)) + 2 * A::aligned(n_a0 * nc * sizeof(T)) +
                             2 * A::aligned(N * nc * sizeof(T)) + 2 * A::aligned(N * sizeof(T)) + A::aligned(N * sizeof(int64_t));
        HIP_TRY(arena.reserve(total));
        d_dbeta = arena.take<T>(N);
        if (dbeta2) d_dbeta2 = arena.take<T>(N);
        d_gamma = arena.take<T>(n_gamma);
        d_alpha = arena.take<T>(n_alpha);
        d_a0_aos = arena.take<T>(n_a0 * nc);
        d_a0_soa = arena.take<T>(n_a0 * nc);
        d_aend_soa = arena.take<T>(N * nc);
        d_aend_aos = arena.take<T>(N * nc);
        d_pend = arena.take<T>(N);
        d_pmax = arena.take<T>(N);
        d_bad = arena.take<int64_t>(N);
    }
    if (traj) HIP_TRY(b_traj_soa.alloc(traj_elems * sizeof(T)));
    HIP_TRY(hipMemcpyAsync(d_dbeta, dbeta, N * sizeof(T), hipMemcpyHostToDevice, st));
    if (dbeta2) HIP_TRY(hipMemcpyAsync(d_dbeta2, dbeta2, N * sizeof(T), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_gamma, gamma, n_gamma * sizeof(T), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_alpha, alpha, n_alpha * sizeof(T), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_a0_aos, a0, n_a0 * nc * sizeof(T), hipMemcpyHostToDevice, st));
    HIP_TRY(Launch<T>::a2s(st, d_a0_aos, d_a0_soa, (long long)n_a0, nc));
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, st));
    rc = sweep_dev<T>(st, n_waves, n_points, n_steps, z_max, save_every, d_dbeta,
                      d_dbeta2, d_gamma, d_alpha, d_a0_soa, flags,
                      d_aend_soa, d_pend, d_pmax, d_bad,
                      traj ? b_traj_soa.as<T>() : nullptr);
    if (rc != PSA_OK) goto done;
    HIP_TRY(hipEventRecord(ev1, st));
    HIP_TRY(Launch<T>::s2a(st, d_aend_soa, d_aend_aos, (long long)N, nc));
    HIP_TRY(hipMemcpyAsync(a_end, d_aend_aos, N * nc * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(p_end, d_pend, N * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(p_max, d_pmax, N * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(first_bad, d_bad, N * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (traj && N == 1) {   // [rows][nw][1] and [1][rows][nw] coincide
        HIP_TRY(hipMemcpyAsync(traj, b_traj_soa.p, traj_elems * sizeof(T), hipMemcpyDeviceToHost, st));
    } else if (traj) {
        HIP_TRY(hipEventCreateWithFlags(&ev_kernel, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_kernel, st));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(b_stage[b].alloc(chunk_pts * point_bytes));
            HIP_TRY(hipStreamCreate(&st_copy[b]));
            HIP_TRY(hipStreamWaitEvent(st_copy[b], ev_kernel, 0));
        }
        int b = 0;
        for (size_t p0 = 0; p0 < N; p0 += chunk_pts, b ^= 1) {
            const size_t pts = (N - p0 < chunk_pts) ? N - p0 : chunk_pts;
            // stream order on st_copy[b] keeps the staging buffer busy until its previous copy has finished
            HIP_TRY(Launch<T>::t2a(st_copy[b], b_traj_soa.as<T>() + 2 * p0, b_stage[b].as<T>(), (long long)pts,
                                   (long long)N, (long long)n_saved, nc));
            HIP_TRY(hipMemcpyAsync(traj + p0 * (size_t)n_saved * nc, b_stage[b].p, pts * point_bytes,
                                   hipMemcpyDeviceToHost, st_copy[b]));
        }
        HIP_TRY(hipStreamSynchronize(st_copy[0]));
        HIP_TRY(hipStreamSynchronize(st_copy[1]));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (elapsed_ms) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        *elapsed_ms = (double)ms;
    }
done:
    for (int b = 0; b < 2; ++b) {
        if (st_copy[b]) {
            (void)hipStreamSynchronize(st_copy[b]);
            (void)hipStreamDestroy(st_copy[b]);
        }
    }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_kernel) (void)hipEventDestroy(ev_kernel);
    if (st) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    return rc;
}

template <typename T> struct GainLaunch;
template <> struct GainLaunch<double> { static constexpr auto fn = psa::launch_gain_summary_f64; };
template <> struct GainLaunch<float> { static constexpr auto fn = psa::launch_gain_summary_f32; };

template <typename T>
int gain_summary_dev(void *stream, int64_t n, const T *d_p, const int64_t *d_bad, double p0_sig, int gain_db, T *d_gain,
                     int64_t *d_best_i, double *d_best_g, int64_t *d_nfin, void *d_ws) {
    if (n < 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0");
    if (!d_best_i || !d_best_g || !d_nfin || !d_ws || (n > 0 && !d_p))
        return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    hipError_t e = GainLaunch<T>::fn((hipStream_t)stream, n, d_p, (const long long *)d_bad, p0_sig, gain_db, d_gain,
                                     (long long *)d_best_i, d_best_g, (long long *)d_nfin, d_ws);
    if (e != hipSuccess) return hip_fail(e, "gain_summary launch");
    return PSA_OK;
}

template <typename T>
int gain_summary_host(int device, int64_t n, const T *p_metric, const int64_t *first_bad, double p0_sig, int gain_db,
                      T *gain_out, int64_t *best_index, double *best_gain, int64_t *n_finite) {
    if (n < 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0");
    if (!best_index || !best_gain || !n_finite || (n > 0 && !p_metric))
        return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    int rc = check_device(device, "the gain summary");
    if (rc != PSA_OK) return rc;
    const size_t N = (size_t)n;
    DeviceScope scope;
    DevBuf bp, bb, bg, bi, bbg, bn, bw;
    HIP_TRY(scope.enter(device));
    HIP_TRY(bp.alloc(N * sizeof(T)));
    if (first_bad) HIP_TRY(bb.alloc(N * 8));
    if (gain_out) HIP_TRY(bg.alloc(N * sizeof(T)));
    HIP_TRY(bi.alloc(8)); HIP_TRY(bbg.alloc(8)); HIP_TRY(bn.alloc(8));
    HIP_TRY(bw.alloc((size_t)psa::gain_summary_workspace_bytes(n)));
    if (N) HIP_TRY(hipMemcpy(bp.p, p_metric, N * sizeof(T), hipMemcpyHostToDevice));
    if (first_bad && N) HIP_TRY(hipMemcpy(bb.p, first_bad, N * 8, hipMemcpyHostToDevice));
    rc = gain_summary_dev<T>(nullptr, n, bp.as<T>(), first_bad ? bb.as<int64_t>() : nullptr, p0_sig, gain_db,
                             gain_out ? bg.as<T>() : nullptr, bi.as<int64_t>(), bbg.as<double>(), bn.as<int64_t>(), bw.p);
    if (rc != PSA_OK) goto done;
    if (gain_out && N) HIP_TRY(hipMemcpy(gain_out, bg.p, N * sizeof(T), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(best_index, bi.p, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(best_gain, bbg.p, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(n_finite, bn.p, 8, hipMemcpyDeviceToHost));
done:
    return rc;
}

// dbeta producer: argument checks + the model struct shared by both kernels
int make_dbeta_model(psa::DbetaModel &m, int method, const int32_t *orders, int n_orders, int max_order,
                     const double *beta, int n_beta, double omega_ref, double two_pi_c, double atol, double rtol) {
    if (method != PSA_DBETA_SYMMETRIC_EVEN && method != PSA_DBETA_GENERAL_TAYLOR)
        return fail(PSA_E_DBETA_MODEL, "method must be PSA_DBETA_SYMMETRIC_EVEN or PSA_DBETA_GENERAL_TAYLOR");
    if (!beta || n_beta < 1 || n_beta > psa::DBETA_MAX_ORDER + 1)
        return fail(PSA_E_DBETA_MODEL, "beta must hold 1..%d coefficients", psa::DBETA_MAX_ORDER + 1);
    for (int n = 0; n <= psa::DBETA_MAX_ORDER; ++n) m.beta[n] = n < n_beta ? beta[n] : 0.0;
    m.omega_ref = omega_ref; m.two_pi_c = two_pi_c; m.atol = atol; m.rtol = rtol;
    m.method = method; m.n_orders = 0; m.max_order = 0;
    for (int k = 0; k < 4; ++k) m.orders[k] = 0;
    if (method == PSA_DBETA_SYMMETRIC_EVEN) {
        if (!orders || n_orders < 1 || n_orders > 4) return fail(PSA_E_DBETA_MODEL, "1..4 even orders expected");
        for (int k = 0; k < n_orders; ++k) {
            if (orders[k] < 2 || orders[k] % 2 || orders[k] > psa::DBETA_MAX_ORDER)
                return fail(PSA_E_DBETA_MODEL, "even_orders must be even ints in [2, %d], got %d", psa::DBETA_MAX_ORDER, orders[k]);
            m.orders[k] = orders[k];
        }
        m.n_orders = n_orders;
    } else {
        if (max_order < 0 || max_order > psa::DBETA_MAX_ORDER)
            return fail(PSA_E_DBETA_MODEL, "max_order must be in [0, %d]", psa::DBETA_MAX_ORDER);
        m.max_order = max_order;
    }
    return PSA_OK;
}

template <typename T> struct DbetaLaunch;
template <> struct DbetaLaunch<double> {
    static constexpr auto grid = psa::launch_dbeta_grid_f64;
    static constexpr auto pairs = psa::launch_dbeta_pairs_f64;
};
template <> struct DbetaLaunch<float> {
    static constexpr auto grid = psa::launch_dbeta_grid_f32;
    static constexpr auto pairs = psa::launch_dbeta_pairs_f32;
};

template <typename T>
int dbeta_grid_dev(void *stream, int method, const int32_t *orders, int n_orders, int max_order, const double *beta,
                   int n_beta, double omega_ref, double two_pi_c, double atol, double rtol, double lambda1_m,
                   const double *d_ax2, int64_t n2, const double *d_ax3, int64_t n3, int64_t first, int64_t n, T *d_out,
                   uint8_t *d_valid) {
    psa::DbetaModel m;
    int rc = make_dbeta_model(m, method, orders, n_orders, max_order, beta, n_beta, omega_ref, two_pi_c, atol, rtol);
    if (rc != PSA_OK) return rc;
    if (n < 0 || n2 <= 0 || n3 <= 0 || first < 0) return fail(PSA_E_NPOINTS, "n_points / axes / first_index out of range");
    if ((long double)first + (long double)n > (long double)n2 * (long double)n3)
        return fail(PSA_E_NPOINTS, "[first_index, first_index + n_points) leaves the %lld x %lld grid", (long long)n2, (long long)n3);
    if (n == 0) return PSA_OK;
    if (!d_ax2 || !d_ax3 || !d_out) return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    hipError_t e = DbetaLaunch<T>::grid((hipStream_t)stream, m, lambda1_m, d_ax2, n2, d_ax3, n3, first, n, d_out, d_valid);
    if (e != hipSuccess) return hip_fail(e, "dbeta_grid launch");
    return PSA_OK;
}

template <typename T>
int dbeta_pairs_dev(void *stream, const int32_t *orders, int n_orders, const double *beta, int n_beta, double omega_d,
                    const double *d_ax1, int64_t n1, const double *d_ax2, int64_t n2, int64_t first, int64_t n,
                    T *d_out1, T *d_out2) {
    psa::DbetaModel m;
    int rc = make_dbeta_model(m, PSA_DBETA_SYMMETRIC_EVEN, orders, n_orders, 0, beta, n_beta, 1.0, 0.0, 0.0, 0.0);
    if (rc != PSA_OK) return rc;
    if (n < 0 || n1 <= 0 || n2 <= 0 || first < 0) return fail(PSA_E_NPOINTS, "n_points / axes / first_index out of range");
    if ((long double)first + (long double)n > (long double)n1 * (long double)n2)
        return fail(PSA_E_NPOINTS, "[first_index, first_index + n_points) leaves the %lld x %lld grid", (long long)n1, (long long)n2);
    if (n == 0) return PSA_OK;
    if (!d_ax1 || !d_ax2 || !d_out1 || !d_out2) return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    hipError_t e = DbetaLaunch<T>::pairs((hipStream_t)stream, m, omega_d, d_ax1, n1, d_ax2, n2, first, n, d_out1, d_out2);
    if (e != hipSuccess) return hip_fail(e, "dbeta_pairs launch");
    return PSA_OK;
}

}  // namespace

extern "C" {

int psa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char *psa_last_error(void) { return g_err; }
const char *psa_version(void) { return "psa-hip 0.2.0 gfx950"; }
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
    int rc = check_device(device, "the RHS kernel");
    if (rc != PSA_OK) return rc;
    const size_t N = (size_t)n;
    DeviceScope scope;
    DevBuf bz, ba, bg, bal, bd, bo, bl, bk, bf;
    HIP_TRY(scope.enter(device));
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
    return gain_summary_dev<double>(stream, n, d_p, d_bad, p0_sig, gain_db, d_gain, d_best_i, d_best_g, d_nfin, d_ws);
}
int psa_gain_summary_f32_dev(void *stream, int64_t n, const float *d_p, const int64_t *d_bad, double p0_sig,
                             int gain_db, float *d_gain, int64_t *d_best_i, double *d_best_g, int64_t *d_nfin,
                             void *d_ws) {
    return gain_summary_dev<float>(stream, n, d_p, d_bad, p0_sig, gain_db, d_gain, d_best_i, d_best_g, d_nfin, d_ws);
}
int psa_gain_summary_f64(int device, int64_t n, const double *p_metric, const int64_t *first_bad, double p0_sig,
                         int gain_db, double *gain_out, int64_t *best_index, double *best_gain, int64_t *n_finite) {
    return gain_summary_host<double>(device, n, p_metric, first_bad, p0_sig, gain_db, gain_out, best_index, best_gain, n_finite);
}
int psa_gain_summary_f32(int device, int64_t n, const float *p_metric, const int64_t *first_bad, double p0_sig,
                         int gain_db, float *gain_out, int64_t *best_index, double *best_gain, int64_t *n_finite) {
    return gain_summary_host<float>(device, n, p_metric, first_bad, p0_sig, gain_db, gain_out, best_index, best_gain, n_finite);
}

/* ---- device-side dbeta producer ---------------------------------------------------------------------------- */
int psa_dbeta_grid_f64_dev(void *stream, int method, const int32_t *orders, int n_orders, int max_order,
                           const double *beta, int n_beta, double omega_ref, double two_pi_c, double atol, double rtol,
                           double lambda1_m, const double *d_lambda2_axis, int64_t n2, const double *d_lambda3_axis,
                           int64_t n3, int64_t first_index, int64_t n_points, double *d_dbeta, uint8_t *d_valid) {
    return dbeta_grid_dev<double>(stream, method, orders, n_orders, max_order, beta, n_beta, omega_ref, two_pi_c, atol,
                                  rtol, lambda1_m, d_lambda2_axis, n2, d_lambda3_axis, n3, first_index, n_points, d_dbeta,
                                  d_valid);
}
int psa_dbeta_grid_f32_dev(void *stream, int method, const int32_t *orders, int n_orders, int max_order,
                           const double *beta, int n_beta, double omega_ref, double two_pi_c, double atol, double rtol,
                           double lambda1_m, const double *d_lambda2_axis, int64_t n2, const double *d_lambda3_axis,
                           int64_t n3, int64_t first_index, int64_t n_points, float *d_dbeta, uint8_t *d_valid) {
    return dbeta_grid_dev<float>(stream, method, orders, n_orders, max_order, beta, n_beta, omega_ref, two_pi_c, atol,
                                 rtol, lambda1_m, d_lambda2_axis, n2, d_lambda3_axis, n3, first_index, n_points, d_dbeta,
                                 d_valid);
}
int psa_dbeta_pairs_f64_dev(void *stream, const int32_t *orders, int n_orders, const double *beta, int n_beta,
                            double omega_d, const double *d_Omega1_axis, int64_t n1, const double *d_Omega2_axis,
                            int64_t n2, int64_t first_index, int64_t n_points, double *d_dbeta1, double *d_dbeta2) {
    return dbeta_pairs_dev<double>(stream, orders, n_orders, beta, n_beta, omega_d, d_Omega1_axis, n1, d_Omega2_axis, n2,
                                   first_index, n_points, d_dbeta1, d_dbeta2);
}
int psa_dbeta_pairs_f32_dev(void *stream, const int32_t *orders, int n_orders, const double *beta, int n_beta,
                            double omega_d, const double *d_Omega1_axis, int64_t n1, const double *d_Omega2_axis,
                            int64_t n2, int64_t first_index, int64_t n_points, float *d_dbeta1, float *d_dbeta2) {
    return dbeta_pairs_dev<float>(stream, orders, n_orders, beta, n_beta, omega_d, d_Omega1_axis, n1, d_Omega2_axis, n2,
                                  first_index, n_points, d_dbeta1, d_dbeta2);
}

int psa_dbeta_grid_f64(int device, int method, const int32_t *orders, int n_orders, int max_order, const double *beta,
                       int n_beta, double omega_ref, double two_pi_c, double atol, double rtol, double lambda1_m,
                       const double *lambda2_axis, int64_t n2, const double *lambda3_axis, int64_t n3,
                       int64_t first_index, int64_t n_points, double *dbeta, uint8_t *valid) {
    if (n_points < 0 || n2 <= 0 || n3 <= 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0 and the axes non-empty");
    if (!lambda2_axis || !lambda3_axis || (n_points > 0 && !dbeta)) return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    int rc = check_device(device, "the dbeta producer");
    if (rc != PSA_OK) return rc;
    if (n_points == 0) return PSA_OK;
    DeviceScope scope;
    DevBuf b2, b3, bo, bv;
    HIP_TRY(scope.enter(device));
    HIP_TRY(b2.alloc((size_t)n2 * 8)); HIP_TRY(b3.alloc((size_t)n3 * 8)); HIP_TRY(bo.alloc((size_t)n_points * 8));
    if (valid) HIP_TRY(bv.alloc((size_t)n_points));
    HIP_TRY(hipMemcpy(b2.p, lambda2_axis, (size_t)n2 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b3.p, lambda3_axis, (size_t)n3 * 8, hipMemcpyHostToDevice));
    rc = psa_dbeta_grid_f64_dev(nullptr, method, orders, n_orders, max_order, beta, n_beta, omega_ref, two_pi_c, atol, rtol,
                                lambda1_m, b2.as<double>(), n2, b3.as<double>(), n3, first_index, n_points,
                                bo.as<double>(), valid ? bv.as<uint8_t>() : nullptr);
    if (rc != PSA_OK) goto done;
    HIP_TRY(hipMemcpy(dbeta, bo.p, (size_t)n_points * 8, hipMemcpyDeviceToHost));
    if (valid) HIP_TRY(hipMemcpy(valid, bv.p, (size_t)n_points, hipMemcpyDeviceToHost));
done:
    return rc;
}

int psa_dbeta_pairs_f64(int device, const int32_t *orders, int n_orders, const double *beta, int n_beta, double omega_d,
                        const double *Omega1_axis, int64_t n1, const double *Omega2_axis, int64_t n2,
                        int64_t first_index, int64_t n_points, double *dbeta1, double *dbeta2) {
    if (n_points < 0 || n1 <= 0 || n2 <= 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0 and the axes non-empty");
    if (!Omega1_axis || !Omega2_axis || (n_points > 0 && (!dbeta1 || !dbeta2))) return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    int rc = check_device(device, "the dbeta producer");
    if (rc != PSA_OK) return rc;
    if (n_points == 0) return PSA_OK;
    DeviceScope scope;
    DevBuf b1, b2, o1, o2;
    HIP_TRY(scope.enter(device));
    HIP_TRY(b1.alloc((size_t)n1 * 8)); HIP_TRY(b2.alloc((size_t)n2 * 8));
    HIP_TRY(o1.alloc((size_t)n_points * 8)); HIP_TRY(o2.alloc((size_t)n_points * 8));
    HIP_TRY(hipMemcpy(b1.p, Omega1_axis, (size_t)n1 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b2.p, Omega2_axis, (size_t)n2 * 8, hipMemcpyHostToDevice));
    rc = psa_dbeta_pairs_f64_dev(nullptr, orders, n_orders, beta, n_beta, omega_d, b1.as<double>(), n1, b2.as<double>(), n2,
                                 first_index, n_points, o1.as<double>(), o2.as<double>());
    if (rc != PSA_OK) goto done;
    HIP_TRY(hipMemcpy(dbeta1, o1.p, (size_t)n_points * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dbeta2, o2.p, (size_t)n_points * 8, hipMemcpyDeviceToHost));
done:
    return rc;
}

}  // extern "C"
