// psa_rk4_f64.hip -- float64 instantiations of the RK4 sweep kernels (gfx950): one point per lane, and one point per
// lane PAIR (psa_rk4_split_kernel.inc.h) for sweeps smaller than the chip.
#include <atomic>

#include "psa_rk4_split_kernel.inc.h"

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
// I2 = 183.6 / 296.8 per lane.  The z-loop is issue-bound and sequential, so a launch takes as long as its busiest SIMD:
//     T(layout) ~ I(layout) * k / eff(k),      k = ceil(waves(layout) / SIMDs)  waves sharing a SIMD,
// eff(k) = sustained DP issue rate with k resident waves (tools/dp_peak.hip: 0.896, 0.94, 0.96 of nominal for 1, 2, >= 3).
// Two lanes per point win whenever the doubled wave count still rounds to the same k (N <= 32 768 on MI355X: k = 1 either
// way, 0.61-0.65 of the time) and again in windows like 65 536 < N <= 98 304 (three half-length waves per SIMD beat two
// full-length ones); in between (32 768 < N <= 65 536) every SIMD that holds 64 points needs I1 whatever the layout, so one
// lane per point is the floor there -- a hybrid launch cannot beat its slowest wave (profiles/r03_split_cliff.log).
static bool split_is_faster(int n_waves, long long n_points, int simds) {
    const double i1 = (n_waves == 4) ? 300.7 : 471.9, i2 = (n_waves == 4) ? 183.6 : 296.8;
    auto eff = [](long long k) { return k <= 1 ? 0.896 : (k == 2 ? 0.94 : 0.96); };
    const long long w1 = (n_points + 63) / 64, w2 = (2 * n_points + 63) / 64;
    const long long k1 = (w1 + simds - 1) / simds, k2 = (w2 + simds - 1) / simds;
    const double t1 = i1 * (double)k1 / eff(k1), t2 = i2 * (double)k2 / eff(k2);
    return t2 < t1;
}

// split: 1 = two lanes per point, 0 = one, -1 = choose by the cost model above.
hipError_t launch_sweep_f64(hipStream_t s, int n_waves, int check, bool lds, int block, bool lossless, int split,
                            const SweepArgs<double> &a) {
    const int simds = simd_count(s);
    const long long split_waves = (2 * a.n_points + 63) / 64;
    // a two-lane trajectory launch folds the lane's wave offset into the 32-bit store offset: NW * N * 16 B < 2^32
    const bool split_ok = !lds && (a.traj == nullptr || (unsigned long long)a.traj_ld * n_waves * 16ull < (1ull << 32));
    const bool use_split = split_ok && (split == 1 || (split < 0 && split_is_faster(n_waves, a.n_points, simds)));
    if (use_split) {
        const int sb = (block == 64 || 2 * split_waves <= (long long)simds) ? 64 : 256;   // see launch_sweep_split
        return launch_sweep_split(s, n_waves, check, lossless, sb, a);
    }
    return launch_sweep_t<double>(s, n_waves, check, lds, block, lossless, a);
}
}  // namespace psa
