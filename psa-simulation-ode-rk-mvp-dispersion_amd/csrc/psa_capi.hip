// psa_capi.hip -- the extern "C" surface of libpsa_hip.so (see include/psa_rk4.h for the contract and the
// reference file:line each entry point replaces).  Host code only: argument validation, HBM staging for the
// host-buffer variants, launches.  No CPU compute path exists here on purpose: without a gfx950 device the
// host-buffer calls fail with PSA_E_DEVICE / a hipError_t -- they never fall back.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
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

// Leading dimension of the trajectory buffer [n_saved][n_waves][ld][2].  The four (six) wave regions of a row, and
// consecutive rows, are ld * pair bytes apart; when that is a multiple of 2 MiB the streams of the resident waves collide in
// the memory system's address hash and the store rate drops (store-only probe, tools/hbm_write_peak: 5.7 / 5.6 / 4.9 TB/s at
// 262 144 / 524 288 / 1 048 576 points against 6.3 / 6.5 / 6.6 with the regions 4 352 B further apart; sizes that are not
// such multiples, and strides of 1 MiB or less, are best left alone: profiles/r03_store_layout_probe.log).
int64_t traj_ld_of(int64_t n_points, size_t elem_size) {
    const int64_t pair = 2 * (int64_t)elem_size;
    const int64_t bytes = n_points * pair;
    if (n_points <= 0 || bytes % (2ll << 20) != 0) return n_points;
    return n_points + 4352 / pair;          // 17 x 256 B: 272 float64 points, 544 float32 points
}

int validate_common(int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
                    const void *dbeta, const void *dbeta2, const void *gamma, const void *alpha, const void *a0,
                    const void *a_end, const void *p_end, const void *p_max, const void *first_bad, uint32_t flags,
                    bool has_traj, size_t elem_size) {
    if (n_waves != 4 && n_waves != 6) return fail(PSA_E_NWAVES, "n_waves must be 4 or 6, got %d", n_waves);
    if (n_points < 0) return fail(PSA_E_NPOINTS, "n_points must be >= 0, got %lld", (long long)n_points);
    // a launch is at most 2^32 - 1 threads in x, and the two-lane float64 layout spends two of them per point
    if (n_points > PSA_MAX_POINTS)
        return fail(PSA_E_TOO_LARGE, "n_points %lld exceeds the launch limit of %lld points", (long long)n_points,
                    (long long)PSA_MAX_POINTS);
    if (n_steps <= 0 || n_steps > 2147483647LL)
        return fail(PSA_E_NSTEPS, "n_steps must be in [1, 2^31), got %lld", (long long)n_steps);
    if (!(z_max > 0.0) || !std::isfinite(z_max)) return fail(PSA_E_ZMAX, "z_max must be positive");
    if (save_every <= 0) return fail(PSA_E_SAVE_EVERY, "save_every must be a positive integer");
    if (n_waves == 6 && !dbeta2 && n_points > 0) return fail(PSA_E_DBETA2, "n_waves == 6 requires dbeta2");
    if (n_waves == 4 && dbeta2) return fail(PSA_E_DBETA2, "dbeta2 must be NULL for n_waves == 4");
    {
        const int layouts = !!(flags & PSA_OPT_SPLIT_POINT) + !!(flags & PSA_OPT_ONE_LANE) + !!(flags & PSA_OPT_QUAD_POINT);
        if (layouts > 1)
            return fail(PSA_E_FLAGS, "PSA_OPT_SPLIT_POINT, PSA_OPT_ONE_LANE and PSA_OPT_QUAD_POINT exclude each other");
        if ((flags & PSA_OPT_QUAD_POINT) && n_waves != 4)
            return fail(PSA_E_FLAGS, "PSA_OPT_QUAD_POINT (four lanes per point) exists for the 4-wave model only");
    }
    if ((flags & PSA_OPT_F32_SCALAR) && (flags & PSA_OPT_F32_PACKED))
        return fail(PSA_E_FLAGS, "PSA_OPT_F32_SCALAR and PSA_OPT_F32_PACKED exclude each other");
    if (n_points > 0 && (!dbeta || !gamma || !alpha || !a0 || !a_end || !p_end || !p_max || !first_bad))
        return fail(PSA_E_NULLPTR, "a required buffer pointer is NULL");
    if (has_traj) {
        // trajectory rows are addressed as a wave-uniform (row, wave) base + a 32-bit byte offset per lane, kept below 2^31
        const unsigned long long pair = 2ull * elem_size;
        if ((unsigned long long)n_points * pair >= (1ull << 31))
            return fail(PSA_E_TOO_LARGE, "a trajectory launch takes at most %llu points", (1ull << 31) / pair - 1);
        // the two-lane layout folds the lane's wave offset into that 32-bit offset
        const unsigned long long ld = (flags & PSA_OPT_TRAJ_LD) ? (unsigned long long)traj_ld_of(n_points, elem_size) : (unsigned long long)n_points;
        if ((flags & (PSA_OPT_SPLIT_POINT | PSA_OPT_QUAD_POINT)) && ld * n_waves * pair >= (1ull << 32))
            return fail(PSA_E_TOO_LARGE, "a two-lane trajectory launch takes at most %llu points",
                        (1ull << 32) / (n_waves * pair) - 1);
    }
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
    a.traj_ld = (flags & PSA_OPT_TRAJ_LD) ? traj_ld_of(n_points, sizeof(T)) : n_points;
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
        const int split = (flags & PSA_OPT_QUAD_POINT) ? 2 : ((flags & PSA_OPT_SPLIT_POINT) ? 1 : ((flags & PSA_OPT_ONE_LANE) ? 0 : -1));
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
                             d_a0_soa, d_a_end_soa, d_p_end, d_p_max, d_first_bad, flags, d_traj_soa != nullptr, sizeof(T));
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

// ---- host-call contexts ------------------------------------------------------------------------------------------
// What a host-buffer sweep needs besides its buffers -- a stream, two timing events, (for trajectories of more than one
// point) two copy streams and an event, device scratch and a page-locked mirror for small transfers -- is created once per
// concurrent caller and device and kept in a pool: the reference's own scenarios are sweeps of 1, 30 and 100 points
// (main.py), where creating and destroying these per call cost as much as a tenth of the kernel (0.5-0.75 ms of a 4-5 ms
// single run).  A context is leased to one call at a time (concurrent callers on one device each get their own), its
// streams are drained before it goes back, and psa_release_cache() destroys the idle ones.
constexpr size_t CTX_ARENA_KEEP = 64u << 20;     // device scratch larger than this is allocated per call, not kept
constexpr size_t CTX_PINNED_IN = 256u << 10;     // page-locked mirror: inputs ...
constexpr size_t CTX_PINNED_OUT = 768u << 10;    // ... and outputs of a small sweep travel in ONE copy each way

struct HostCtx {
    int device = -1;
    hipStream_t st = nullptr, st_copy[2] = {nullptr, nullptr};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_kernel = nullptr;
    void *arena = nullptr;
    size_t arena_cap = 0;
    char *pinned = nullptr;

    hipError_t init(int dev) {
        device = dev;
        hipError_t e;
        if ((e = hipStreamCreate(&st)) != hipSuccess) return e;
        if ((e = hipEventCreate(&ev0)) != hipSuccess) return e;
        if ((e = hipEventCreate(&ev1)) != hipSuccess) return e;
        return hipHostMalloc((void **)&pinned, CTX_PINNED_IN + CTX_PINNED_OUT, hipHostMallocDefault);
    }
    hipError_t need_copy_streams() {
        hipError_t e;
        for (int b = 0; b < 2; ++b)
            if (!st_copy[b] && (e = hipStreamCreate(&st_copy[b])) != hipSuccess) return e;
        if (!ev_kernel && (e = hipEventCreateWithFlags(&ev_kernel, hipEventDisableTiming)) != hipSuccess) return e;
        return hipSuccess;
    }
    hipError_t need_arena(size_t bytes) {   // grow-only; the caller has checked bytes <= CTX_ARENA_KEEP
        if (bytes <= arena_cap) return hipSuccess;
        if (arena) { (void)hipFree(arena); arena = nullptr; arena_cap = 0; }
        size_t cap = 1u << 20;
        while (cap < bytes) cap <<= 1;
        hipError_t e = hipMalloc(&arena, cap);
        if (e == hipSuccess) arena_cap = cap;
        return e;
    }
    void drain() {
        for (int b = 0; b < 2; ++b) if (st_copy[b]) (void)hipStreamSynchronize(st_copy[b]);
        if (st) (void)hipStreamSynchronize(st);
    }
    void destroy() {   // its device must be current
        drain();
        for (int b = 0; b < 2; ++b) if (st_copy[b]) (void)hipStreamDestroy(st_copy[b]);
        if (ev_kernel) (void)hipEventDestroy(ev_kernel);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (st) (void)hipStreamDestroy(st);
        if (arena) (void)hipFree(arena);
        if (pinned) (void)hipHostFree(pinned);
    }
};

std::mutex g_ctx_mutex;
std::vector<HostCtx *> g_ctx_idle;

// Lease of one context for the scope of a call; the device must already be current (DeviceScope) and stays so until the
// lease has ended (declare the lease AFTER the scope and after any per-call DevBuf: it drains the streams first).
struct CtxLease {
    HostCtx *c = nullptr;
    hipError_t acquire(int device) {
        {
            std::lock_guard<std::mutex> lock(g_ctx_mutex);
            for (size_t i = 0; i < g_ctx_idle.size(); ++i)
                if (g_ctx_idle[i]->device == device) {
                    c = g_ctx_idle[i];
                    g_ctx_idle.erase(g_ctx_idle.begin() + (long)i);
                    return hipSuccess;
                }
        }
        c = new HostCtx();
        hipError_t e = c->init(device);
        if (e != hipSuccess) {
            c->destroy();
            delete c;
            c = nullptr;
        }
        return e;
    }
    ~CtxLease() {
        if (!c) return;
        c->drain();   // nothing of this call may still be in flight when the buffers are reused or freed
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_ctx_idle.push_back(c);
    }
};

// 256-B aligned slices of one allocation
struct Carver {
    char *base = nullptr;
    size_t used = 0;
    static size_t aligned(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    template <typename U> U *take(size_t count) {
        U *p = (U *)(base + used);
        used += aligned(count * sizeof(U));
        return p;
    }
};

// staging for the trajectory transpose of the host-buffer API: two buffers of at most this many bytes each
constexpr size_t TRAJ_STAGE_BYTES = 256u << 20;

#ifdef PSA_FAULT_INJECTION
// Sanitizer builds only (tools/host_sanitize.sh): PSA_FAIL_CHUNK=k makes the k-th staged trajectory chunk of every call
// fail, so the error path OUT of the staging loop (streams drained, per-call buffers freed, context returned) runs under ASan.
static bool injected_chunk_failure(size_t chunk_index) {
    const char *e = getenv("PSA_FAIL_CHUNK");
    return e && (size_t)atoll(e) == chunk_index;
}
#endif

#define HIP_RET(expr)                                                \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return hip_fail(_e, #expr);            \
    } while (0)

template <typename T>
int sweep_host(int device, int n_waves, int64_t n_points, int64_t n_steps, double z_max, int32_t save_every,
               const T *dbeta, const T *dbeta2, const T *gamma, const T *alpha, const T *a0, uint32_t flags,
               T *a_end, T *p_end, T *p_max, int64_t *first_bad, T *traj, double *elapsed_ms) {
    int rc = validate_common(n_waves, n_points, n_steps, z_max, save_every, dbeta, dbeta2, gamma, alpha, a0, a_end,
                             p_end, p_max, first_bad, flags, traj != nullptr, sizeof(T));
    if (rc != PSA_OK) return rc;
    if (elapsed_ms) *elapsed_ms = 0.0;
    if (n_points == 0) return PSA_OK;
    rc = check_device(device, "the RK4 sweep");
    if (rc != PSA_OK) return rc;

    const int nc = 2 * n_waves;
    const size_t N = (size_t)n_points;
    const int64_t n_saved = n_steps / save_every + 1;
    // the device-side trajectory buffer has its own leading dimension (traj_ld_of); the caller's array is dense
    const size_t ld = (size_t)traj_ld_of(n_points, sizeof(T));
    size_t traj_elems = 0;
    if (traj) {
        // ld * n_saved * nc must fit comfortably in int64 / size_t
        const long double te = (long double)ld * (long double)n_saved * (long double)nc;
        if (te > 4.0e18L) return fail(PSA_E_TOO_LARGE, "trajectory buffer too large");
        traj_elems = ld * (size_t)n_saved * (size_t)nc;
        flags |= PSA_OPT_TRAJ_LD;
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
    const size_t traj_bytes = traj_elems * sizeof(T);
    const size_t stage_bytes = chunk_pts * point_bytes;

    // -- layout of the call's device scratch: [inputs] [device-only] [outputs (+ the one-point trajectory)] [trajectory] [staging]
    using C = Carver;
    const size_t in_bytes = C::aligned(N * sizeof(T)) * (dbeta2 ? 2 : 1) + C::aligned(n_gamma * sizeof(T)) +
                            C::aligned(n_alpha * sizeof(T)) + C::aligned(n_a0 * nc * sizeof(T));
    const size_t mid_bytes = C::aligned(n_a0 * nc * sizeof(T)) + C::aligned(N * nc * sizeof(T));
    const bool traj_in_out = traj && N == 1;     // [rows][nw][1] and [1][rows][nw] coincide: it leaves with the outputs
    const size_t out_bytes = C::aligned(N * nc * sizeof(T)) + 2 * C::aligned(N * sizeof(T)) + C::aligned(N * sizeof(int64_t)) +
                             (traj_in_out ? C::aligned(traj_bytes) : 0);
    const size_t small_total = in_bytes + mid_bytes + out_bytes;
    const size_t traj_total = (traj && !traj_in_out) ? C::aligned(traj_bytes) + 2 * C::aligned(stage_bytes) : 0;

    DeviceScope scope;
    DevBuf b_small, b_traj;                      // per-call allocations when the cached scratch is not used
    CtxLease lease;
    HIP_RET(scope.enter(device));
    if (traj) {   // say "too large" before hipMalloc says "out of memory"
        size_t free_b = 0, total_b = 0;
        HIP_RET(hipMemGetInfo(&free_b, &total_b));
        const long double need = (long double)small_total + (long double)traj_total;
        if (need > (long double)CTX_ARENA_KEEP && need > (long double)free_b)
            return fail(PSA_E_TOO_LARGE, "trajectory of %.3g GB does not fit the %.3g GB free on device %d",
                        (double)(need / 1e9L), (double)free_b / 1e9, device);
    }
    HIP_RET(lease.acquire(device));
    HostCtx &cx = *lease.c;
    hipStream_t st = cx.st;

    Carver small, big;
    const bool keep_all = small_total + traj_total <= CTX_ARENA_KEEP;
    if (keep_all) {
        HIP_RET(cx.need_arena(small_total + traj_total));
        small.base = (char *)cx.arena;
        big.base = (char *)cx.arena + small_total;
    } else {
        if (small_total <= CTX_ARENA_KEEP) {
            HIP_RET(cx.need_arena(small_total));
            small.base = (char *)cx.arena;
        } else {
            HIP_RET(b_small.alloc(small_total));
            small.base = (char *)b_small.p;
        }
        if (traj_total) {
            HIP_RET(b_traj.alloc(traj_total));
            big.base = (char *)b_traj.p;
        }
    }
    T *d_dbeta = small.take<T>(N);
    T *d_dbeta2 = dbeta2 ? small.take<T>(N) : nullptr;
    T *d_gamma = small.take<T>(n_gamma);
    T *d_alpha = small.take<T>(n_alpha);
    T *d_a0_aos = small.take<T>(n_a0 * nc);
    T *d_a0_soa = small.take<T>(n_a0 * nc);
    T *d_aend_soa = small.take<T>(N * nc);
    const size_t out_off = small.used;
    T *d_aend_aos = small.take<T>(N * nc);
    T *d_pend = small.take<T>(N);
    T *d_pmax = small.take<T>(N);
    int64_t *d_bad = small.take<int64_t>(N);
    T *d_traj = nullptr, *d_stage[2] = {nullptr, nullptr};
    if (traj_in_out) {
        d_traj = small.take<T>(traj_elems);
    } else if (traj) {
        d_traj = big.take<T>(traj_elems);
        d_stage[0] = (T *)big.take<char>(stage_bytes);
        d_stage[1] = (T *)big.take<char>(stage_bytes);
    }

    // -- inputs: a small sweep's arrive in ONE copy from the page-locked mirror (same slice offsets as on the device)
    const bool mirror = in_bytes <= CTX_PINNED_IN && out_bytes <= CTX_PINNED_OUT;
    if (mirror) {
        char *m = cx.pinned;
        auto put = [&](const T *dev, const T *src, size_t count) {
            std::memcpy(m + ((const char *)dev - small.base), src, count * sizeof(T));
        };
        put(d_dbeta, dbeta, N);
        if (dbeta2) put(d_dbeta2, dbeta2, N);
        put(d_gamma, gamma, n_gamma);
        put(d_alpha, alpha, n_alpha);
        put(d_a0_aos, a0, n_a0 * nc);
        HIP_RET(hipMemcpyAsync(small.base, m, in_bytes, hipMemcpyHostToDevice, st));
    } else {
        HIP_RET(hipMemcpyAsync(d_dbeta, dbeta, N * sizeof(T), hipMemcpyHostToDevice, st));
        if (dbeta2) HIP_RET(hipMemcpyAsync(d_dbeta2, dbeta2, N * sizeof(T), hipMemcpyHostToDevice, st));
        HIP_RET(hipMemcpyAsync(d_gamma, gamma, n_gamma * sizeof(T), hipMemcpyHostToDevice, st));
        HIP_RET(hipMemcpyAsync(d_alpha, alpha, n_alpha * sizeof(T), hipMemcpyHostToDevice, st));
        HIP_RET(hipMemcpyAsync(d_a0_aos, a0, n_a0 * nc * sizeof(T), hipMemcpyHostToDevice, st));
    }
    HIP_RET(Launch<T>::a2s(st, d_a0_aos, d_a0_soa, (long long)n_a0, nc));
    HIP_RET(hipEventRecord(cx.ev0, st));
    rc = sweep_dev<T>(st, n_waves, n_points, n_steps, z_max, save_every, d_dbeta, d_dbeta2, d_gamma, d_alpha, d_a0_soa,
                      flags, d_aend_soa, d_pend, d_pmax, d_bad, d_traj);
    if (rc != PSA_OK) return rc;
    HIP_RET(hipEventRecord(cx.ev1, st));
    HIP_RET(Launch<T>::s2a(st, d_aend_soa, d_aend_aos, (long long)N, nc));
    if (mirror) {
        HIP_RET(hipMemcpyAsync(cx.pinned + CTX_PINNED_IN, small.base + out_off, out_bytes, hipMemcpyDeviceToHost, st));
    } else {
        HIP_RET(hipMemcpyAsync(a_end, d_aend_aos, N * nc * sizeof(T), hipMemcpyDeviceToHost, st));
        HIP_RET(hipMemcpyAsync(p_end, d_pend, N * sizeof(T), hipMemcpyDeviceToHost, st));
        HIP_RET(hipMemcpyAsync(p_max, d_pmax, N * sizeof(T), hipMemcpyDeviceToHost, st));
        HIP_RET(hipMemcpyAsync(first_bad, d_bad, N * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        if (traj_in_out) HIP_RET(hipMemcpyAsync(traj, d_traj, traj_bytes, hipMemcpyDeviceToHost, st));
    }
    if (traj && !traj_in_out) {
        HIP_RET(cx.need_copy_streams());
        HIP_RET(hipEventRecord(cx.ev_kernel, st));
        for (int b = 0; b < 2; ++b) HIP_RET(hipStreamWaitEvent(cx.st_copy[b], cx.ev_kernel, 0));
        int b = 0;
        for (size_t p0 = 0; p0 < N; p0 += chunk_pts, b ^= 1) {
            const size_t pts = (N - p0 < chunk_pts) ? N - p0 : chunk_pts;
#ifdef PSA_FAULT_INJECTION
            if (injected_chunk_failure(p0 / chunk_pts)) return hip_fail(hipErrorUnknown, "injected failure in the staging loop");
#endif
            // stream order on st_copy[b] keeps the staging buffer busy until its previous copy has finished
            HIP_RET(Launch<T>::t2a(cx.st_copy[b], d_traj + 2 * p0, d_stage[b], (long long)pts, (long long)ld,
                                   (long long)n_saved, nc));
            HIP_RET(hipMemcpyAsync(traj + p0 * (size_t)n_saved * nc, d_stage[b], pts * point_bytes, hipMemcpyDeviceToHost,
                                   cx.st_copy[b]));
        }
        HIP_RET(hipStreamSynchronize(cx.st_copy[0]));
        HIP_RET(hipStreamSynchronize(cx.st_copy[1]));
    }
    HIP_RET(hipStreamSynchronize(st));
    if (mirror) {
        const char *m = cx.pinned + CTX_PINNED_IN;
        auto get = [&](void *dst, const void *dev, size_t bytes) {
            std::memcpy(dst, m + ((const char *)dev - (small.base + out_off)), bytes);
        };
        get(a_end, d_aend_aos, N * nc * sizeof(T));
        get(p_end, d_pend, N * sizeof(T));
        get(p_max, d_pmax, N * sizeof(T));
        get(first_bad, d_bad, N * sizeof(int64_t));
        if (traj_in_out) get(traj, d_traj, traj_bytes);
    }
    if (elapsed_ms) {
        float ms = 0.f;
        HIP_RET(hipEventElapsedTime(&ms, cx.ev0, cx.ev1));
        *elapsed_ms = (double)ms;
    }
    return PSA_OK;
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
    using C = Carver;
    // [inputs: p, first_bad] [workspace] [outputs: gain, (best_index, best_gain, n_finite)]
    const size_t in_bytes = C::aligned(N * sizeof(T)) + (first_bad ? C::aligned(N * 8) : 0);
    const size_t ws_bytes = C::aligned((size_t)psa::gain_summary_workspace_bytes(n));
    const size_t out_bytes = (gain_out ? C::aligned(N * sizeof(T)) : 0) + C::aligned(3 * 8);
    const size_t total = in_bytes + ws_bytes + out_bytes;
    DeviceScope scope;
    DevBuf b_all;
    CtxLease lease;
    HIP_RET(scope.enter(device));
    HIP_RET(lease.acquire(device));
    HostCtx &cx = *lease.c;
    Carver cv;
    if (total <= CTX_ARENA_KEEP) {
        HIP_RET(cx.need_arena(total));
        cv.base = (char *)cx.arena;
    } else {
        HIP_RET(b_all.alloc(total));
        cv.base = (char *)b_all.p;
    }
    T *d_p = cv.take<T>(N);
    int64_t *d_bad = first_bad ? cv.take<int64_t>(N) : nullptr;
    void *d_ws = cv.take<char>(ws_bytes);
    const size_t out_off = cv.used;
    T *d_gain = gain_out ? cv.take<T>(N) : nullptr;
    int64_t *d_scal = cv.take<int64_t>(3);          // best_index | best_gain (double) | n_finite
    const bool mirror = in_bytes <= CTX_PINNED_IN && out_bytes <= CTX_PINNED_OUT;
    if (mirror) {
        if (N) std::memcpy(cx.pinned, p_metric, N * sizeof(T));
        if (first_bad && N) std::memcpy(cx.pinned + ((char *)d_bad - cv.base), first_bad, N * 8);
        if (N) HIP_RET(hipMemcpyAsync(cv.base, cx.pinned, in_bytes, hipMemcpyHostToDevice, cx.st));
    } else {
        HIP_RET(hipMemcpyAsync(d_p, p_metric, N * sizeof(T), hipMemcpyHostToDevice, cx.st));
        if (first_bad) HIP_RET(hipMemcpyAsync(d_bad, first_bad, N * 8, hipMemcpyHostToDevice, cx.st));
    }
    rc = gain_summary_dev<T>(cx.st, n, d_p, d_bad, p0_sig, gain_db, d_gain, d_scal, (double *)(d_scal + 1), d_scal + 2, d_ws);
    if (rc != PSA_OK) return rc;
    int64_t scal[3];
    if (mirror) {
        char *m = cx.pinned + CTX_PINNED_IN;
        HIP_RET(hipMemcpyAsync(m, cv.base + out_off, out_bytes, hipMemcpyDeviceToHost, cx.st));
        HIP_RET(hipStreamSynchronize(cx.st));
        if (gain_out && N) std::memcpy(gain_out, m, N * sizeof(T));
        std::memcpy(scal, m + ((char *)d_scal - (cv.base + out_off)), sizeof(scal));
    } else {
        if (gain_out && N) HIP_RET(hipMemcpyAsync(gain_out, d_gain, N * sizeof(T), hipMemcpyDeviceToHost, cx.st));
        HIP_RET(hipMemcpyAsync(scal, d_scal, sizeof(scal), hipMemcpyDeviceToHost, cx.st));
        HIP_RET(hipStreamSynchronize(cx.st));
    }
    *best_index = scal[0];
    std::memcpy(best_gain, &scal[1], 8);
    *n_finite = scal[2];
    return PSA_OK;
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
const char *psa_version(void) { return "psa-hip 0.3.0 gfx950"; }
int64_t psa_n_saved(int64_t n_steps, int32_t save_every) {
    if (n_steps < 0 || save_every <= 0) return -1;
    return n_steps / save_every + 1;
}
int64_t psa_traj_ld(int64_t n_points, int32_t elem_size) {
    if (n_points < 0 || (elem_size != 4 && elem_size != 8)) return -1;
    return traj_ld_of(n_points, (size_t)elem_size);
}
int psa_release_cache(void) {
    std::vector<HostCtx *> idle;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        idle.swap(g_ctx_idle);
    }
    int prev = -1;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (HostCtx *c : idle) {
        if (hipSetDevice(c->device) == hipSuccess) c->destroy();
        delete c;
    }
    if (have_prev) (void)hipSetDevice(prev);
    return (int)idle.size();
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
