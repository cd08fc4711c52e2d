/* GPU-side plain-C client used for host-sanitizer runs (tools/host_sanitize.sh) and by the test-suite: every host-buffer
 * entry point of psa_rk4.h through the ABI --
 *   1. a small sweep with a trajectory, the gain summary (f64), one RHS evaluation;
 *   2. a trajectory that leaves the device in TWO chunks, the second one ragged (4 197 points x 1 001 rows = 269 MB through
 *      the 256 MB staging buffers), checked row by row against the small sweep's arithmetic (A[-1] == last row, row 0 == A0);
 *   3. the dbeta producers psa_dbeta_grid_f64 / psa_dbeta_pairs_f64 and the float32 sweep + psa_gain_summary_f32;
 *   4. (library built with -DPSA_FAULT_INJECTION, PSA_FAIL_CHUNK set) a failure INSIDE the staging loop, then the same call
 *      again without it: the context must come back clean; psa_release_cache() at the end.
 */
#define _POSIX_C_SOURCE 200112L   /* setenv / unsetenv under -std=c99 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psa_rk4.h"
#define N 257
#define NS 1005
#define SE 10
#define NBIG 4197      /* 4 160 points fit one staging buffer at 1 001 rows: one full chunk + 37 points */
#define NSBIG 1000

static const double A0[8] = {0.7071067811865476, 0, 0.7071067811865476, 0, 0.0031622776601683794, 0, 0.0031622776601683794, 0};

static int big_trajectory(int expect_failure) {
    const int64_t rows = psa_n_saved(NSBIG, 1);
    double *dbeta = (double *)malloc(sizeof(double) * NBIG), *a_end = (double *)malloc(sizeof(double) * NBIG * 8);
    double *p_end = (double *)malloc(sizeof(double) * NBIG), *p_max = (double *)malloc(sizeof(double) * NBIG);
    int64_t *bad = (int64_t *)malloc(sizeof(int64_t) * NBIG);
    double *traj = (double *)malloc(sizeof(double) * (size_t)NBIG * rows * 8);
    double gamma = 0.0115, alpha = 0.0, ms = 0;     /* alpha == 0 broadcast: the lossless instantiation */
    int rc, ret = 0;
    if (!dbeta || !a_end || !p_end || !p_max || !bad || !traj) return 20;
    for (int i = 0; i < NBIG; ++i) dbeta[i] = -0.05 + 0.1 * i / (NBIG - 1);
    rc = psa_rk4_sweep_f64(0, 4, NBIG, NSBIG, 100.0, 1, dbeta, NULL, &gamma, &alpha, A0,
                           PSA_BCAST_GAMMA | PSA_BCAST_ALPHA | PSA_BCAST_A0 | PSA_OPT_CHECK_NAN, a_end, p_end, p_max, bad, traj, &ms);
    if (expect_failure) {
        if (rc <= 0 || !strstr(psa_last_error(), "injected")) { fprintf(stderr, "expected the injected failure, got rc=%d\n", rc); ret = 21; }
    } else if (rc) {
        fprintf(stderr, "big trajectory rc=%d %s\n", rc, psa_last_error());
        ret = 22;
    } else {
        for (int i = 0; i < NBIG && !ret; ++i) {
            const double *first = traj + (size_t)i * rows * 8, *last = first + (size_t)(rows - 1) * 8;
            for (int c = 0; c < 8; ++c)
                if (first[c] != A0[c] || last[c] != a_end[i * 8 + c]) ret = 23;
            if (bad[i] != -1 || !(p_max[i] >= p_end[i])) ret = 24;
        }
        /* alpha == 0: total power is conserved along every trajectory (both sides of the chunk boundary included) */
        for (int i = 4150; i < 4170 && !ret; ++i)
            for (int64_t r = 0; r < rows; r += 100) {
                const double *row = traj + ((size_t)i * rows + r) * 8;
                double p = 0;
                for (int c = 0; c < 8; ++c) p += row[c] * row[c];
                if (fabs(p - 1.00002) > 1e-9) ret = 25;
            }
        if (!ret) printf("  two-chunk trajectory ok: %d points x %lld rows, kernel %.3f ms\n", NBIG, (long long)rows, ms);
    }
    free(dbeta); free(a_end); free(p_end); free(p_max); free(bad); free(traj);
    return ret;
}

int main(void) {
    static double dbeta[N], a_end[N * 8], p_end[N], p_max[N], gain[N];
    static int64_t bad[N];
    const int64_t rows = psa_n_saved(NS, SE);
    double *traj = (double *)malloc(sizeof(double) * N * rows * 8);
    double gamma = 0.0115, alpha = 1.15e-4;
    double ms = 0, best_gain = 0;
    int64_t best = -1, nfin = 0;
    if (!traj || psa_device_count() < 1) { fprintf(stderr, "no device\n"); return 2; }
    for (int i = 0; i < N; ++i) dbeta[i] = -0.05 + 0.1 * i / (N - 1);
    int rc = psa_rk4_sweep_f64(0, 4, N, NS, 100.5, SE, dbeta, NULL, &gamma, &alpha, A0,
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
    rc = psa_yaman_rhs_f64(0, 1, &z, A0, &gamma, &alpha, dbeta, out, lin, NULL, NULL);
    if (rc || fabs(lin[0] + 0.5 * alpha * A0[0]) > 1e-18) return 6;
    printf("abi_gpu_client: small sweep ok: kernel %.3f ms, best gain %.6f dB at point %lld\n", ms, best_gain, (long long)best);

    /* a single point with its trajectory: inputs, outputs and rows travel through the page-locked mirror */
    {
        double one_end[8], one_pe, one_pm, one_traj[101 * 8];
        int64_t one_bad;
        rc = psa_rk4_sweep_f64(0, 4, 1, NS, 100.5, SE, dbeta + 5, NULL, &gamma, &alpha, A0,
                               PSA_BCAST_GAMMA | PSA_BCAST_ALPHA | PSA_BCAST_A0 | PSA_OPT_CHECK_NAN | PSA_OPT_EXACT_STEP,
                               one_end, &one_pe, &one_pm, &one_bad, one_traj, NULL);
        if (rc) { fprintf(stderr, "one point rc=%d %s\n", rc, psa_last_error()); return 7; }
        for (int c = 0; c < 8; ++c)
            if (fabs(one_end[c] - a_end[5 * 8 + c]) > 1e-12 || one_traj[100 * 8 + c] != one_end[c]) return 8;
    }

    /* the dbeta producers */
    {
        const double beta[5] = {0, 0, -2.5e-29, 3.3e-41, -1.6e-55};
        const int32_t orders[2] = {2, 4};
        double lam2[3] = {1556e-9, 1558e-9, 1560e-9}, lam3[5] = {1540e-9, 1545e-9, 1550.5e-9, 1562e-9, 0.3e-6};
        double db[15], db1[12], db2[12], O1[3] = {2e12, 5e12, 9e12}, O2[4] = {3e12, 4e12, 6e12, 8e12};
        uint8_t valid[15];
        rc = psa_dbeta_grid_f64(0, PSA_DBETA_SYMMETRIC_EVEN, orders, 2, 0, beta, 5, 1.2e15, 1883651567.3088531, 0.0, 1e-12,
                                1550e-9, lam2, 3, lam3, 5, 0, 15, db, valid);
        if (rc) { fprintf(stderr, "dbeta grid rc=%d %s\n", rc, psa_last_error()); return 9; }
        for (int i = 0; i < 15; ++i)
            if ((i % 5 == 4) ? (valid[i] || db[i] == db[i]) : (!valid[i] || !(fabs(db[i]) < 1.0))) return 10;   /* 0.3 um: no idler */
        rc = psa_dbeta_pairs_f64(0, orders, 2, beta, 5, 2.5e12, O1, 3, O2, 4, 0, 12, db1, db2);
        if (rc || db1[0] != db1[3] || db2[1] != db2[5] || !(fabs(db1[11]) < 1.0)) { fprintf(stderr, "dbeta pairs rc=%d\n", rc); return 11; }
        if (psa_dbeta_grid_f64(0, PSA_DBETA_SYMMETRIC_EVEN, orders, 2, 0, beta, 5, 1.2e15, 1883651567.3088531, 0.0, 1e-12,
                               1550e-9, lam2, 3, lam3, 5, 10, 6, db, valid) != PSA_E_NPOINTS) return 12;    /* leaves the grid */
    }

    /* float32 sweep (packed kernel, odd point count) + float32 gain summary */
    {
        static float db32[N], ae32[N * 8], pe32[N], pm32[N], g32[N];
        float a032[8], gam32 = 0.0115f, al32 = 1.15e-4f;
        for (int c = 0; c < 8; ++c) a032[c] = (float)A0[c];
        for (int i = 0; i < N; ++i) db32[i] = (float)dbeta[i];
        rc = psa_rk4_sweep_f32(0, 4, N, NS, 100.5, SE, db32, NULL, &gam32, &al32, a032,
                               PSA_BCAST_GAMMA | PSA_BCAST_ALPHA | PSA_BCAST_A0 | PSA_OPT_CHECK_NAN, ae32, pe32, pm32, bad, NULL, NULL);
        if (rc) { fprintf(stderr, "f32 sweep rc=%d %s\n", rc, psa_last_error()); return 13; }
        for (int i = 0; i < N; ++i)
            if (bad[i] != -1 || fabs((double)pm32[i] - p_max[i]) > 2e-3 * p_max[i]) return 14;
        rc = psa_gain_summary_f32(0, N, pm32, bad, 1e-5, 1, g32, &best, &best_gain, &nfin);
        if (rc || nfin != N || best < 0 || fabs((double)g32[best] - best_gain) > 1e-4) { fprintf(stderr, "f32 summary rc=%d\n", rc); return 15; }
    }

    /* the chunked trajectory; with PSA_FAIL_CHUNK set (fault-injection build) first the failing call, then a clean one */
    if (getenv("PSA_FAIL_CHUNK")) {
        rc = big_trajectory(1);
        if (rc) return rc;
        unsetenv("PSA_FAIL_CHUNK");
        printf("  injected staging-loop failure reported and cleaned up\n");
    }
    rc = big_trajectory(0);
    if (rc) return rc;
    /* an argument error after real work, then the cache goes */
    if (psa_rk4_sweep_f64(0, 4, N, NS, 100.5, 0, dbeta, NULL, &gamma, &alpha, A0, 0, a_end, p_end, p_max, bad, NULL, NULL) != PSA_E_SAVE_EVERY) return 16;
    if (psa_release_cache() < 1) return 17;
    if (psa_release_cache() != 0) return 18;
    printf("abi_gpu_client ok\n");
    free(traj);
    return 0;
}
