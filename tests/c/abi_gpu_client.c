/* GPU-side plain-C client used for host-sanitizer runs (tools/host_sanitize.sh): a small sweep through the
 * host-buffer entry points, trajectory included, plus the RHS and gain-summary calls. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "psa_rk4.h"
#define N 257
#define NS 1005
#define SE 10
int main(void) {
    static double dbeta[N], a_end[N * 8], p_end[N], p_max[N], gain[N];
    static int64_t bad[N];
    const int64_t rows = psa_n_saved(NS, SE);
    double *traj = (double *)malloc(sizeof(double) * N * rows * 8);
    double gamma = 0.0115, alpha = 1.15e-4, a0[8] = {0.7071067811865476, 0, 0.7071067811865476, 0, 0.0031622776601683794, 0, 0.0031622776601683794, 0};
    double ms = 0, best_gain = 0;
    int64_t best = -1, nfin = 0;
    if (!traj || psa_device_count() < 1) { fprintf(stderr, "no device\n"); return 2; }
    for (int i = 0; i < N; ++i) dbeta[i] = -0.05 + 0.1 * i / (N - 1);
    int rc = psa_rk4_sweep_f64(0, 4, N, NS, 100.5, SE, dbeta, NULL, &gamma, &alpha, a0,
                               PSA_BCAST_GAMMA | PSA_BCAST_ALPHA | PSA_BCAST_A0 | PSA_OPT_CHECK_NAN | PSA_OPT_EXACT_STEP,
                               a_end, p_end, p_max, bad, traj, &ms);
    if (rc) { fprintf(stderr, "sweep rc=%d %s\n", rc, psa_last_error()); return 1; }
    for (int i = 0; i < N; ++i) {
        if (bad[i] != -1 || !(p_max[i] >= p_end[i])) return 3;
        const double *last = traj + ((size_t)i * rows + (rows - 1)) * 8;
        for (int c = 0; c < 8; ++c) if (last[c] != a_end[i * 8 + c]) return 4;      /* A[-1] == last saved row */
    }
    rc = psa_gain_summary_f64(0, N, p_max, bad, 1e-5, 1, gain, &best, &best_gain, &nfin);
    if (rc || nfin != N || best < 0 || gain[best] != best_gain) { fprintf(stderr, "summary rc=%d\n", rc); return 5; }
    double z = 3.0, out[8], lin[8];
    rc = psa_yaman_rhs_f64(0, 1, &z, a0, &gamma, &alpha, dbeta, out, lin, NULL, NULL);
    if (rc || fabs(lin[0] + 0.5 * alpha * a0[0]) > 1e-18) return 6;
    printf("abi_gpu_client ok: kernel %.3f ms, best gain %.6f dB at point %lld\n", ms, best_gain, (long long)best);
    free(traj);
    return 0;
}
