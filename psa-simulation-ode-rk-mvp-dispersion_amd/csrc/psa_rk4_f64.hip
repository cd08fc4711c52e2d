// psa_rk4_f64.hip -- float64 instantiations of the RK4 sweep kernel (gfx950).
#include "psa_rk4_kernel.inc.h"

namespace psa {
hipError_t launch_sweep_f64(hipStream_t s, int n_waves, int check, bool lds, int block, bool lossless,
                            const SweepArgs<double> &a) {
    return launch_sweep_t<double>(s, n_waves, check, lds, block, lossless, a);
}
}  // namespace psa
