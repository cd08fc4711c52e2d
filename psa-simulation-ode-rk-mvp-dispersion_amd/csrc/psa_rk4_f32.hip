// psa_rk4_f32.hip -- float32 instantiations of the RK4 sweep (gfx950): one point per lane, and two points per lane
// with packed math.  `pack`: 1 = packed, 0 = one point per lane, -1 = choose.  Measured on MI355X the packed form is
// never slower (1.14x at 65 536 points, 1.85x at 131 072, 1.94x at 2^20: non-packed and packed float32 VALU ops both
// occupy a SIMD for 4 cycles per wave64 -- tools/sp_peak.hip, profiles/r02_sp_peak.log -- so packing is the only way to
// the float32 vector peak), hence the default is packed for every sweep of at least two points.
#include "psa_rk4_pk_kernel.inc.h"

namespace psa {
hipError_t launch_sweep_f32(hipStream_t s, int n_waves, int check, bool lds, int block, int pack, bool lossless,
                            const SweepArgs<float> &a) {
    const bool use_pack = !lds && (pack == 1 || (pack < 0 && a.n_points >= 2));
    if (use_pack) return launch_sweep_pk(s, n_waves, check, block, a);
    return launch_sweep_t<float>(s, n_waves, check, lds, block, lossless, a);
}
}  // namespace psa
