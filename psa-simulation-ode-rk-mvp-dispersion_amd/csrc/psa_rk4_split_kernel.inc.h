// psa_rk4_split_kernel.inc.h -- float64 RK4 sweep with TWO LANES PER SWEEP POINT (gfx950).
//
// Why: the sweep is FP64-issue bound and the z-loop is sequential, so a sweep smaller than the chip
// (N <= 32 768 points = 512 waves for 1 024 SIMDs: BASELINE config 5's per-GPU shard; config 1's single point)
// leaves SIMDs idle that no amount of occupancy can use.  Here a point's waves are divided between an even
// lane and its odd neighbour, which halves the dependent instruction stream per lane:
//
//   4 waves   even lane: (u, v) = (A_p1, A_p2)            odd lane: (u, v) = (A_s, A_i)
//             dA_u/dz = (-alpha/2 + i*gamma*f_u) A_u + i*conj(A_v) * F,    F = E_lane * Q
//             Q = (A_u A_v) of the PARTNER lane,  E_lane = 2*gamma*exp(+i*dbeta*z) (even) | its conjugate (odd)
//             -- the pumps are driven by E*(A_s A_i), the sidebands by conj(E)*(A_p1 A_p2)  (yaman_model.py:174-181),
//             so both lanes run the SAME instruction stream with a lane-dependent sign of dbeta.
//   6 waves   even lane: (w, u, v) = (A_p1, A_s1, A_i1), dbeta_1      odd lane: (A_p2, A_s2, A_i2), dbeta_2
//             dA_w/dz = (...) A_w + i*conj(W) * (t + T),   t = E_own * (A_u A_v),  W, T = the partner's A_w, t
//             dA_u/dz = (...) A_u + i*conj(A_v) * conj(E_own) * (A_w W)            (same for v with u)
//   f_j = 2*S - |A_j|^2 with S = own partial sum + the partner's.
//
// Values cross between the two lanes with DPP quad_perm:[1,0,3,2] moves (two v_mov_b32_dpp per double: gfx950's
// DP ALU accepts no quad_perm, so the exchange cannot be folded into the consuming v_fma_f64).  Per RHS evaluation:
// 4 waves 39 instructions (6 of them moves) instead of 64; 6 waves 65 (10 moves) instead of 100.
// Same RK4 regrouping, phase recurrence, save / NaN semantics as rk4_sweep_kernel -- see that file.
#pragma once
#include "psa_rk4_kernel.inc.h"

namespace psa {

// the value the neighbouring lane (lane ^ 1) holds
__device__ __forceinline__ double from_partner(const double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);  // quad_perm:[1,0,3,2]
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// (A/B, DESIGN.md 5.2: routing the exchanges that have slack -- the other pump, the other pair's driving term -- through
// the LDS crossbar with ds_swizzle_b32 frees 32 VALU slots per step but its latency is exposed with one wave per SIMD:
// 12.50 ms instead of 11.73 ms on the config-5 shard shape.  All exchanges stay DPP moves.)

// out = base + c * dA/dz(a) for the lane's own waves (stage coefficient folded into g, tg, ha, E as in yaman_stage).
// NL = waves per lane (2 | 3); a = [Re, Im] x NL in the lane's order given above.
template <int NL, bool LOSS>
__device__ __forceinline__ void split_stage(const double (&a)[2 * NL], const double (&base)[2 * NL], const double Er,
                                            const double Ei, const double g, const double tg, const double ha,
                                            double (&out)[2 * NL]) {
    double p[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) p[j] = fma_(a[2 * j], a[2 * j], a[2 * j + 1] * a[2 * j + 1]);
    double sl = p[0] + p[1];
    if constexpr (NL == 3) sl += p[2];
    const double s = sl + from_partner(sl);
    const double gs = tg * s;
    double gj[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) gj[j] = fma_(-g, p[j], gs);
    auto link = [&](const double gsig, const double v, const int c) -> double {
        if constexpr (LOSS) return fma_(gsig, v, fma_(ha, a[c], base[c]));
        else return fma_(gsig, v, base[c]);
    };
    // the last two waves of the lane are a pair (u, v) coupled through one driving term F
    constexpr int U = 2 * (NL - 2);
    const double xu = a[U], yu = a[U + 1], xv = a[U + 2], yv = a[U + 3];
    const double qr = fma_(xu, xv, -(yu * yv)), qi = fma_(xu, yv, yu * xv);  // A_u * A_v
    double Fr, Fi;
    if constexpr (NL == 2) {
        const double Qr = from_partner(qr), Qi = from_partner(qi);
        Fr = fma_(Er, Qr, -(Ei * Qi));
        Fi = fma_(Er, Qi, Ei * Qr);
    } else {
        const double xw = a[0], yw = a[1];
        const double Xw = from_partner(xw), Yw = from_partner(yw);  // the other pump
        const double tr = fma_(Er, qr, -(Ei * qi)), ti = fma_(Er, qi, Ei * qr);  // E_own * (A_u A_v)
        const double Fpr = tr + from_partner(tr), Fpi = ti + from_partner(ti);  // sum over both pairs
        const double q12r = fma_(xw, Xw, -(yw * Yw)), q12i = fma_(xw, Yw, yw * Xw);  // A_p1 * A_p2
        Fr = fma_(Er, q12r, Ei * q12i);  // conj(E_own) * (A_p1 A_p2)
        Fi = fma_(Er, q12i, -(Ei * q12r));
        // pump: (ha + i g_w) A_w + i conj(W) Fp
        out[0] = fma_(Yw, Fpr, fma_(-Xw, Fpi, link(-gj[0], yw, 0)));
        out[1] = fma_(Xw, Fpr, fma_(Yw, Fpi, link(gj[0], xw, 1)));
    }
    const double gu = gj[NL - 2], gv = gj[NL - 1];
    out[U] = fma_(yv, Fr, fma_(-xv, Fi, link(-gu, yu, U)));
    out[U + 1] = fma_(xv, Fr, fma_(yv, Fi, link(gu, xu, U + 1)));
    out[U + 2] = fma_(yu, Fr, fma_(-xu, Fi, link(-gv, yv, U + 2)));
    out[U + 3] = fma_(xu, Fr, fma_(yu, Fi, link(gv, xv, U + 3)));
}

template <int NW, int CHECK, bool TRAJ, int BLOCK, bool LOSS>
__global__ void __launch_bounds__(BLOCK) rk4_sweep_split_kernel(const SweepArgs<double> A) {
    constexpr int NL = NW / 2;    // waves per lane
    constexpr int NC = 2 * NL;    // real components per lane
    constexpr int RESYNC = Phase<double>::RESYNC;
    const long long gid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const long long idx = gid >> 1;   // sweep point: lanes 2k and 2k+1 of a wave share one
    const int role = (int)(gid & 1);
    const long long N = A.n_points;
    if (idx >= N) return;             // both lanes of a pair leave together

    // which global wave each of the lane's waves is
    int wave_of[NL];
    if constexpr (NL == 2) {
        wave_of[0] = 2 * role;
        wave_of[1] = 2 * role + 1;
    } else {
        wave_of[0] = role;
        wave_of[1] = 2 + 2 * role;
        wave_of[2] = 3 + 2 * role;
    }
    const bool owns_signal = (NL == 2) ? (role == 1) : (role == 0);   // wave index 2, the gain summary's wave
    constexpr int SIG = (NL == 2) ? 0 : 2;                            // its component offset in the owning lane

    double y[NC];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        y[2 * j] = A.a0[(long long)(2 * wave_of[j]) * A.a0_ld + idx * A.a0_stride];
        y[2 * j + 1] = A.a0[(long long)(2 * wave_of[j] + 1) * A.a0_ld + idx * A.a0_stride];
    }
    const double g = A.gamma[idx * A.gamma_stride];
    const double tg = g + g;
    const double ha = -0.5 * A.alpha[idx * A.alpha_stride];
    // the lane's phase rate: 4 waves +dbeta (pumps) | -dbeta (sidebands: conj(E)); 6 waves the own pair's dbeta_k
    double dbd;
    if constexpr (NL == 2) dbd = role ? -A.dbeta[idx] : A.dbeta[idx];
    else dbd = role ? A.dbeta2[idx] : A.dbeta[idx];

    const double hd = A.z_max / (double)A.n_steps;
    const double hh = 0.5 * hd;
    const double g_d = hh * g, tg_d = hh * tg, ha_d = hh * ha;
    const double g_h = hd * g, tg_h = hd * tg, ha_h = hd * ha;
    const double third = 1.0 / 3.0;
    const double e_amp = tg_d;

    double rc, rs, Er = e_amp, Ei = 0.0;
    Phase<double>::eval(dbd * (0.5 * hd), rc, rs);

    double pe = fma_(y[SIG], y[SIG], y[SIG + 1] * y[SIG + 1]);
    double pm = pe;
    long long bad = -1;
    // any component of the POINT non-finite (own lane's or the partner's)
    auto nonfinite_on = [&](const double (&v)[NC]) -> bool {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) t = fma_(v[c], 0.0, t);
        t += from_partner(t);
        return t != t;
    };
    auto point_nonfinite = [&]() -> bool { return nonfinite_on(y); };

    const int se = A.save_every;
    const int n_rows = A.n_steps / se;
    const int n_run = (CHECK != CHECK_NONE) ? A.n_steps : n_rows * se;

    // trajectory rows [row][wave][N][2]: the lane writes its own waves' (re, im) pairs.  wave_of[j] = U_j + role * R_j, so the
    // address splits into a wave-uniform part (row, U_j: an SGPR pair) and a per-lane constant 32-bit byte offset
    // (role * R_j * N + idx) * 16 -- the global_store saddr form of rk4_sweep_kernel, no vector instruction spent on
    // addressing inside the z-loop.  (The C-ABI keeps NW * N * 16 B < 2^32 for two-lane trajectory launches.)
    using Pair = typename PairOf<double>::type;
    const long long LD = A.traj_ld;   // points per (row, wave) region (psa_traj_ld)
    unsigned lane_off[NL];
    int wave_u[NL];
    if constexpr (NL == 2) {
        wave_u[0] = 0;
        wave_u[1] = 1;
        lane_off[0] = lane_off[1] = (unsigned)((unsigned long long)(role * 2 * LD + idx) * sizeof(Pair));
    } else {
        wave_u[0] = 0;
        wave_u[1] = 2;
        wave_u[2] = 3;
        lane_off[0] = (unsigned)((unsigned long long)(role * LD + idx) * sizeof(Pair));
        lane_off[1] = lane_off[2] = (unsigned)((unsigned long long)(role * 2 * LD + idx) * sizeof(Pair));
    }
    auto store_traj_row = [&](const int r) {
        const char *rowb = reinterpret_cast<const char *>(A.traj) + (long long)r * NW * LD * (long long)sizeof(Pair);
#pragma unroll
        for (int j = 0; j < NL; ++j)
            store_pair_nt(rowb + (long long)wave_u[j] * LD * (long long)sizeof(Pair), lane_off[j], Pair{y[2 * j], y[2 * j + 1]});
    };
    auto store_a_end = [&]() {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            A.a_end[(long long)(2 * wave_of[j]) * N + idx] = y[2 * j];
            A.a_end[(long long)(2 * wave_of[j] + 1) * N + idx] = y[2 * j + 1];
        }
    };
    if constexpr (TRAJ) store_traj_row(0);
    if (n_rows == 0) store_a_end();

    // one RK4 step, regrouped exactly as rk4_step_on of rk4_sweep_kernel (integrators.py:54-59); the per-step finite test of
    // CHECK_EXACT is not in it: the exact index comes from a replay (exact_test below), as in rk4_sweep_kernel
    auto rk4_step_on = [&](double (&y)[NC], double &Er, double &Ei) {
        double Y2[NC], Y3[NC], Y4[NC], t[NC], D[NC];
        split_stage<NL, LOSS>(y, y, Er, Ei, g_d, tg_d, ha_d, Y2);
        rotate(Er, Ei, rc, rs);  // z + h/2
        split_stage<NL, LOSS>(Y2, y, Er, Ei, g_d, tg_d, ha_d, Y3);
        const double E2r = Er + Er, E2i = Ei + Ei;
        split_stage<NL, LOSS>(Y3, y, E2r, E2i, g_h, tg_h, ha_h, Y4);
#pragma unroll
        for (int c = 0; c < NC; ++c) t[c] = fma_(2.0, Y3[c], fma_(-4.0, y[c], Y2[c])) + Y4[c];
        rotate(Er, Ei, rc, rs);  // z + h
        split_stage<NL, LOSS>(Y4, t, Er, Ei, g_d, tg_d, ha_d, D);
#pragma unroll
        for (int c = 0; c < NC; ++c) y[c] = fma_(D[c], third, y[c]);
    };
    auto rk4_step = [&](const int) { rk4_step_on(y, Er, Ei); };
    auto seed_on = [&](const int step, double &er, double &ei) {
        double c, s;
        Phase<double>::eval(dbd * ((double)step * hd), c, s);
        er = e_amp * c;
        ei = e_amp * s;
    };

    // CHECK_EXACT by replay (see rk4_sweep_kernel): state at the last test point; a wave with a newly failing point repeats
    // the steps since then on a copy, testing after each one.  Both lanes of a point take the branch together (the test
    // exchanges their partial sums), and the replay's own exchanges stay within the pair.
    constexpr bool REPLAY = CHECK == CHECK_EXACT;
    double y_chk[REPLAY ? NC : 1], Er_chk = 0.0, Ei_chk = 0.0;
    int i_chk = 0;
    auto checkpoint = [&](const int step) {
        if constexpr (REPLAY) {
#pragma unroll
            for (int c = 0; c < NC; ++c) y_chk[c] = y[c];
            Er_chk = Er;
            Ei_chk = Ei;
            i_chk = step;
        }
    };
    auto exact_test = [&](const int i_now) {
        if constexpr (REPLAY) {
            const bool newly_bad = bad < 0 && point_nonfinite();
            if (__builtin_amdgcn_ballot_w64(newly_bad) != 0) {
                double yy[NC], er = Er_chk, ei = Ei_chk;
#pragma unroll
                for (int c = 0; c < NC; ++c) yy[c] = y_chk[c];
                int ii = i_chk;
                while (ii < i_now) {
                    if (ii % RESYNC == 0) seed_on(ii, er, ei);            // the forward pass seeds at the same steps
                    const int to_seed = RESYNC - ii % RESYNC;
                    const int e = (i_now - ii > to_seed) ? ii + to_seed : i_now;
#pragma nounroll
                    for (int st = ii; st < e; ++st) {
                        rk4_step_on(yy, er, ei);
                        if (bad < 0 && nonfinite_on(yy)) bad = st;
                    }
                    ii = e;
                }
            }
            checkpoint(i_now);
        }
    };

#ifdef PSA_SPLIT_TRAJ_SPREAD
    // A/B hook (tools/ab_traj_stores.sh, profiles/r03_split_traj_ab.log): the same step, with the stores of the PREVIOUS row
    // (y is unchanged until the step's last line) issued one per stage instead of as a burst after the step.  Measured
    // SLOWER (32 768 points, every step saved: 1.71 vs 1.62 ms for 4 waves, 1.62 vs 1.54 for 6), as were default instead of
    // non-temporal stores (1.64 / 1.56): with one wave per SIMD a store that finds the queue full stalls the only wave,
    // wherever in the step it is issued.  Not compiled in.
    auto store_traj_wave = [&](const int r, const int j) {
        const char *rowb = reinterpret_cast<const char *>(A.traj) + (long long)r * NW * LD * (long long)sizeof(Pair);
        store_pair_nt(rowb + (long long)wave_u[j] * LD * (long long)sizeof(Pair), lane_off[j], Pair{y[2 * j], y[2 * j + 1]});
    };
    auto rk4_step_spread = [&](const int step_index, const int r_prev) {
        double Y2[NC], Y3[NC], Y4[NC], t[NC], D[NC];
        split_stage<NL, LOSS>(y, y, Er, Ei, g_d, tg_d, ha_d, Y2);
        store_traj_wave(r_prev, 0);
        rotate(Er, Ei, rc, rs);
        split_stage<NL, LOSS>(Y2, y, Er, Ei, g_d, tg_d, ha_d, Y3);
        store_traj_wave(r_prev, 1);
        const double E2r = Er + Er, E2i = Ei + Ei;
        split_stage<NL, LOSS>(Y3, y, E2r, E2i, g_h, tg_h, ha_h, Y4);
        if constexpr (NL == 3) store_traj_wave(r_prev, 2);
#pragma unroll
        for (int c = 0; c < NC; ++c) t[c] = fma_(2.0, Y3[c], fma_(-4.0, y[c], Y2[c])) + Y4[c];
        rotate(Er, Ei, rc, rs);
        split_stage<NL, LOSS>(Y4, t, Er, Ei, g_d, tg_d, ha_d, D);
#pragma unroll
        for (int c = 0; c < NC; ++c) y[c] = fma_(D[c], third, y[c]);
        (void)step_index;
    };
#endif

    // ---- save_every == 1 with a trajectory: every step is a saved row -- the dedicated loop of rk4_sweep_kernel (per row:
    // |A_sig|^2, running maximum, block-mode finite test, the lane's NL streaming stores; two steps per trip).
    if constexpr (TRAJ) {
        if (se == 1) {
            auto save_row = [&](const int r) {
                pe = fma_(y[SIG], y[SIG], y[SIG + 1] * y[SIG + 1]);
                pm = pe > pm ? pe : pm;               // NaN is made to propagate after the loop (it is sticky in y)
                if constexpr (CHECK != CHECK_NONE) {      // a row is a step here: exact in either mode
                    if (bad < 0 && point_nonfinite()) bad = r - 1;
                }
                store_traj_row(r);
            };
            int i = 0;
            while (i < n_run) {
                seed_on(i, Er, Ei);
                const int end = (n_run - i > RESYNC) ? i + RESYNC : n_run;
#ifdef PSA_SPLIT_TRAJ_SPREAD
                for (; i < end; ++i) {      // row i was "saved" (summary only) after step i-1; its stores ride in step i
                    if (i == 0) rk4_step(0); else rk4_step_spread(i, i);
                    pe = fma_(y[SIG], y[SIG], y[SIG + 1] * y[SIG + 1]);
                    pm = pe > pm ? pe : pm;
                    if constexpr (CHECK != CHECK_NONE) {
                        if (bad < 0 && point_nonfinite()) bad = i;
                    }
                }
#else
                for (; i + 2 <= end; i += 2) {
                    rk4_step(i);
                    save_row(i + 1);
                    rk4_step(i + 1);
                    save_row(i + 2);
                }
                if (i < end) {
                    rk4_step(i);
                    save_row(i + 1);
                    ++i;
                }
#endif
            }
#ifdef PSA_SPLIT_TRAJ_SPREAD
            if (n_run > 0) store_traj_row(n_run);
#endif
            if (pe != pe) pm = pe;
            store_a_end();
            if (owns_signal) {
                A.p_end[idx] = pe;
                A.p_max[idx] = pm;
            }
            if (role == 0) A.first_bad[idx] = bad;
            return;
        }
    }

    // seeds on the absolute grid i = 0, RESYNC, ...: the trajectory does not depend on save_every (see rk4_sweep_kernel)
    int i = 0, row = 0;
    int next_save = (n_rows > 0) ? se : 0x7fffffff;
    int next_seed = 0;
    checkpoint(0);
    while (i < n_run) {
        if (i == next_seed) {
            seed_on(i, Er, Ei);
            next_seed = (n_run - i > RESYNC) ? i + RESYNC : 0x7fffffff;
        }
        int end = n_run < next_seed ? n_run : next_seed;
        end = end < next_save ? end : next_save;
        const int m = end - i;
        int j = 0;
        // four steps per trip, then two, then one: with one wave per SIMD a taken back-edge is ~32 exposed cycles
        // (tools/issue_probe.hip); four against two measured -0.5 % (config-5 shard) ... -1.2 % (4 096 points, four lanes)
        for (; j + 4 <= m; j += 4) {
            rk4_step(i + j);
            rk4_step(i + j + 1);
            rk4_step(i + j + 2);
            rk4_step(i + j + 3);
        }
        for (; j + 2 <= m; j += 2) {
            rk4_step(i + j);
            rk4_step(i + j + 1);
        }
        if (j < m) rk4_step(i + j);
        i = end;
        if (i == next_save) {
            ++row;
            pe = fma_(y[SIG], y[SIG], y[SIG + 1] * y[SIG + 1]);
            pm = (pe > pm || pe != pe) ? pe : pm;
            if constexpr (CHECK == CHECK_BLOCK) {
                if (bad < 0 && point_nonfinite()) bad = i - 1;
            }
            exact_test(i);
            if constexpr (TRAJ) store_traj_row(row);
            if (row == n_rows) {
                store_a_end();
                next_save = 0x7fffffff;
            } else {
                next_save += se;
            }
        }
    }
    if constexpr (CHECK == CHECK_BLOCK) {
        if (bad < 0 && n_run > 0 && point_nonfinite()) bad = n_run - 1;
    }
    if (n_run > i_chk) exact_test(n_run);   // the unsaved tail (CHECK_EXACT only)
    if (owns_signal) {
        A.p_end[idx] = pe;
        A.p_max[idx] = pm;
    }
    if (role == 0) A.first_bad[idx] = bad;
}

template <int NW, int CHECK, bool TRAJ>
static hipError_t launch_split_one(hipStream_t s, bool lossless, int block, const SweepArgs<double> &a) {
    const long long lanes = 2 * a.n_points;
    if (block == 256) {
        const dim3 grid((unsigned)((lanes + 255) / 256));
        if (lossless) hipLaunchKernelGGL((rk4_sweep_split_kernel<NW, CHECK, TRAJ, 256, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((rk4_sweep_split_kernel<NW, CHECK, TRAJ, 256, true>), grid, dim3(256), 0, s, a);
    } else {
        const dim3 grid((unsigned)((lanes + 63) / 64));
        if (lossless) hipLaunchKernelGGL((rk4_sweep_split_kernel<NW, CHECK, TRAJ, 64, false>), grid, dim3(64), 0, s, a);
        else hipLaunchKernelGGL((rk4_sweep_split_kernel<NW, CHECK, TRAJ, 64, true>), grid, dim3(64), 0, s, a);
    }
    return hipGetLastError();
}

template <int NW>
static hipError_t launch_split_nw(hipStream_t s, int check, bool lossless, int block, const SweepArgs<double> &a) {
    const bool traj = a.traj != nullptr;
    switch (check) {
        case CHECK_NONE:
            return traj ? launch_split_one<NW, CHECK_NONE, true>(s, lossless, block, a) : launch_split_one<NW, CHECK_NONE, false>(s, lossless, block, a);
        case CHECK_BLOCK:
            return traj ? launch_split_one<NW, CHECK_BLOCK, true>(s, lossless, block, a) : launch_split_one<NW, CHECK_BLOCK, false>(s, lossless, block, a);
        default:
            return traj ? launch_split_one<NW, CHECK_EXACT, true>(s, lossless, block, a) : launch_split_one<NW, CHECK_EXACT, false>(s, lossless, block, a);
    }
}

// block: 64 (one wave per workgroup: a sweep of few waves is spread over as many CUs as it has waves) or 256 (four waves
// per workgroup land on the four SIMDs of one CU: the placement that gives every wave its own SIMD when the sweep fills
// the chip -- 1 024 single-wave workgroups measured 12 % slower at N = 32 768 because some SIMDs received two).
static hipError_t launch_sweep_split(hipStream_t s, int n_waves, int check, bool lossless, int block, const SweepArgs<double> &a) {
    if (a.n_points == 0) return hipSuccess;
    return n_waves == 4 ? launch_split_nw<4>(s, check, lossless, block, a) : launch_split_nw<6>(s, check, lossless, block, a);
}

}  // namespace psa
