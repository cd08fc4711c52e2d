/* A plain-C consumer of include/psa_rk4.h: proves the ABI is C (no C++/torch types), links against
 * libpsa_hip.so and exercises the documented argument-error codes.  No GPU needed: every call below must be
 * rejected in validation or is a pure host helper.  Exit code 0 = all as documented. */
#include <stdio.h>
#include <string.h>
#include "psa_rk4.h"

#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "FAILED: %s (line %d): %s\n", #cond, __LINE__, psa_last_error()); return 1; } } while (0)

int main(void) {
    double buf[64];
    int64_t bad[8];
    memset(buf, 0, sizeof buf);
    EXPECT(strstr(psa_version(), "gfx950") != NULL);
    EXPECT(psa_n_saved(1005, 10) == 101);
    EXPECT(psa_device_count() >= 0);
    EXPECT(psa_rk4_sweep_f64_dev(NULL, 5, 8, 10, 1.0, 1, buf, NULL, buf, buf, buf, 0u, buf, buf, buf, bad, NULL) == PSA_E_NWAVES);
    EXPECT(psa_rk4_sweep_f64_dev(NULL, 4, 8, 0, 1.0, 1, buf, NULL, buf, buf, buf, 0u, buf, buf, buf, bad, NULL) == PSA_E_NSTEPS);
    EXPECT(psa_rk4_sweep_f64_dev(NULL, 4, 8, 10, -1.0, 1, buf, NULL, buf, buf, buf, 0u, buf, buf, buf, bad, NULL) == PSA_E_ZMAX);
    EXPECT(psa_rk4_sweep_f64_dev(NULL, 4, 8, 10, 1.0, 0, buf, NULL, buf, buf, buf, 0u, buf, buf, buf, bad, NULL) == PSA_E_SAVE_EVERY);
    EXPECT(psa_rk4_sweep_f64_dev(NULL, 6, 8, 10, 1.0, 1, buf, NULL, buf, buf, buf, 0u, buf, buf, buf, bad, NULL) == PSA_E_DBETA2);
    EXPECT(psa_rk4_sweep_f64_dev(NULL, 4, 8, 10, 1.0, 1, NULL, NULL, buf, buf, buf, 0u, buf, buf, buf, bad, NULL) == PSA_E_NULLPTR);
    EXPECT(strlen(psa_last_error()) > 0);
    /* an empty sweep is a valid no-op on both faces */
    EXPECT(psa_rk4_sweep_f64(0, 4, 0, 10, 1.0, 1, NULL, NULL, NULL, NULL, NULL, PSA_OPT_CHECK_NAN, NULL, NULL, NULL, NULL, NULL, NULL) == PSA_OK);
    EXPECT(psa_gain_summary_workspace_bytes(65536) > 0);
    printf("abi_client ok: %s, %d device(s)\n", psa_version(), psa_device_count());
    return 0;
}
