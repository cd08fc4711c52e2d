// psa_rk4_f64.hip -- float64 instantiations of the RK4 sweep kernels (gfx950): one point per lane, and one point per
// lane PAIR (psa_rk4_split_kernel.inc.h) for sweeps smaller than the chip.
#include <atomic>

#include "psa_rk4_split_kernel.inc.h"

namespace psa {

// SIMDs of the current device (4 per CU), cached per device ordinal.
static int simd_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        v = 4 * cus;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// split: 1 = two lanes per point, 0 = one, -1 = choose: two lanes per point exactly when the doubled wave count still
// gives every wave a SIMD of its own (N <= 32 768 on MI355X) -- beyond that the extra waves only queue behind each other.
hipError_t launch_sweep_f64(hipStream_t s, int n_waves, int check, bool lds, int block, bool lossless, int split,
                            const SweepArgs<double> &a) {
    const long long split_waves = (2 * a.n_points + 63) / 64;
    const bool use_split = !lds && (split == 1 || (split < 0 && split_waves <= (long long)simd_count()));
    if (use_split) {
        const int sb = (block == 64 || 2 * split_waves <= (long long)simd_count()) ? 64 : 256;   // see launch_sweep_split
        return launch_sweep_split(s, n_waves, check, lossless, sb, a);
    }
    return launch_sweep_t<double>(s, n_waves, check, lds, block, lossless, a);
}
}  // namespace psa
