// psa_rk4_f32.hip -- float32 instantiations of the RK4 sweep kernel (gfx950).
#include "psa_rk4_kernel.inc.h"

namespace psa {
hipError_t launch_sweep_f32(hipStream_t s, int n_waves, int check, bool lds, int block, const SweepArgs<float> &a) {
    return launch_sweep_t<float>(s, n_waves, check, lds, block, a);
}
}  // namespace psa
