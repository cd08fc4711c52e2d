// psa_rk4_f64.hip -- float64 instantiations of the RK4 sweep kernels (gfx950): one point per lane, and one point per
// lane PAIR (psa_rk4_split_kernel.inc.h) for sweeps smaller than the chip.
#include <atomic>

#include "psa_rk4_quad_kernel.inc.h"

namespace psa {

// SIMDs (4 per CU) of the device the launch goes to -- the stream's device, which need not be the thread's current one
// (the _dev entry points take the caller's stream); cached per device ordinal.
static int simd_count(hipStream_t s) {
    static std::atomic<int> cache[64];
    int dev = -1;
    if (s == nullptr || hipStreamGetDevice(s, &dev) != hipSuccess) {
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    }
    if (dev < 0 || dev >= 64) return 1024;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        v = 4 * cus;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// ---- which lane layout?  A cost model from measured instruction counts (profiles/kernels.json, SQ_INSTS_VALU per wave
// per z-step): one lane per point issues I1 = 300.7 (4 waves) / 471.9 (6 waves) instructions per step, two lanes per point
// I2 = 183.6 / 296.8 per lane, four lanes per point (4 waves only, psa_rk4_quad_kernel.inc.h) I4 ~ 157.  The z-loop is issue-bound and sequential, so a launch takes as long as its busiest SIMD:
//     T(layout) ~ I(layout) * k / eff(k),      k = ceil(waves(layout) / SIMDs)  waves sharing a SIMD,
// eff(k) = sustained DP issue rate with k resident waves (tools/dp_peak.hip: 0.896, 0.94, 0.96 of nominal for 1, 2, >= 3).
// Two lanes per point win whenever the doubled wave count still rounds to the same k (N <= 32 768 on MI355X: k = 1 either
// way, 0.61-0.65 of the time) and again in windows like 65 536 < N <= 98 304 (three half-length waves per SIMD beat two
// full-length ones); in between (32 768 < N <= 65 536) every SIMD that holds 64 points needs I1 whatever the layout, so one
// lane per point is the floor there -- a hybrid launch cannot beat its slowest wave (profiles/r03_split_cliff.log).
// -> lanes per point: 1, 2 or (4 waves only) 4
static int best_lanes_per_point(int n_waves, long long n_points, int simds) {
    const double i1 = (n_waves == 4) ? 300.7 : 471.9, i2 = (n_waves == 4) ? 183.6 : 296.8, i4 = 157.0;
    auto eff = [](long long k) { return k <= 1 ? 0.896 : (k == 2 ? 0.94 : 0.96); };
    auto cost = [&](double instr, long long lanes) {
        const long long k = ((lanes + 63) / 64 + simds - 1) / simds;
        return instr * (double)k / eff(k);
    };
    const double t1 = cost(i1, n_points), t2 = cost(i2, 2 * n_points);
    int best = t2 < t1 ? 2 : 1;
    if (n_waves == 4 && cost(i4, 4 * n_points) < (best == 2 ? t2 : t1)) best = 4;
    return best;
}

// split: 1 = two lanes per point, 2 = four lanes per point (4 waves only), 0 = one, -1 = choose by the cost model above.
hipError_t launch_sweep_f64(hipStream_t s, int n_waves, int check, bool lds, int block, bool lossless, int split,
                            const SweepArgs<double> &a) {
    const int simds = simd_count(s);
    // a multi-lane trajectory launch folds the lane's wave offset into the 32-bit store offset: NW * ld * 16 B < 2^32
    const bool multi_ok = !lds && (a.traj == nullptr || (unsigned long long)a.traj_ld * n_waves * 16ull < (1ull << 32));
    int lanes = 1;
    if (multi_ok) lanes = split == 1 ? 2 : (split == 2 && n_waves == 4 ? 4 : (split < 0 ? best_lanes_per_point(n_waves, a.n_points, simds) : 1));
    if (lanes == 1) return launch_sweep_t<double>(s, n_waves, check, lds, block, lossless, a);
    // 64-thread workgroups while the sweep's waves fit half the SIMDs (spread over as many CUs as it has waves), 256 beyond
    const long long waves = ((long long)lanes * a.n_points + 63) / 64;
    const int sb = (block == 64 || 2 * waves <= (long long)simds) ? 64 : 256;
    return lanes == 2 ? launch_sweep_split(s, n_waves, check, lossless, sb, a) : launch_sweep_quad(s, check, lossless, sb, a);
}
}  // namespace psa
