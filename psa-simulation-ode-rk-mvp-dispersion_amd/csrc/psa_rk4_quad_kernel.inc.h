// psa_rk4_quad_kernel.inc.h -- float64 RK4 sweep of the 4-wave model with FOUR LANES PER SWEEP POINT (gfx950).
//
// Why: the reference's own scenarios are tiny sweeps (main.py: 1, 30 and 100 points; BASELINE config 1 is a single run), where
// the chip is empty and the only cost is the length of the dependent instruction stream per lane.  One lane per point issues
// 300.7 instructions per step, two lanes per point 183.6 per lane (psa_rk4_split_kernel.inc.h); here lane r = lane & 3 of a
// quad holds ONE wave of the point (0: pump 1, 1: pump 2, 2: signal, 3: idler) and issues ~155:
//
//     dA_u/dz = (-alpha/2 + i*gamma*f_u) A_u + i*conj(A_v) * F        v = the pair partner (lane ^ 1)
//     F = E_lane * Q,   Q = (A_u A_v) of the OTHER pair (lane ^ 2),   E_lane = 2*gamma*exp(+i*dbeta*z) for the pumps,
//                                                                     its conjugate for the sidebands (yaman_model.py:174-181)
//     f_u = 2*S - |A_u|^2,  S = the quad's sum of |A|^2 (two DPP exchange levels)
//
// Per RHS evaluation and lane: |A|^2 2, S 6 (two moves + an add per level), gamma*f 2, partner's (x, y) 4 moves, A_u*A_v 4,
// the other pair's product 4 moves, F 4, the two 4-deep chains 8  =>  34 (16 of them v_mov_b32_dpp); per step 4 x 34 + 18.
// Every wave of the point goes through the arithmetic it goes through in the two-lane kernel; the only difference is that
// the second lane of a pair forms A_u*A_v with its own wave as the FMA's exact factor (one ulp), so the layouts agree to
// ~1e-12 after 1e4 steps, like one lane and two lanes do.  Measured: x0.81-0.88 of the two-lane kernel's time for
// N <= 4 096 (tools/quad_lane_probe.hip, profiles/r03_quad_lane_probe.log); chosen by
// the cost model of psa_rk4_f64.hip while 4*N lanes still give every wave its own SIMD (N <= 16 384 on MI355X).
// Same RK4 regrouping, phase recurrence on the absolute seed grid, save / NaN semantics (exact index by replay) and
// trajectory stores as rk4_sweep_kernel -- see that file.
#pragma once
#include "psa_rk4_split_kernel.inc.h"

namespace psa {

template <int CTRL> __device__ __forceinline__ double quad_xchg(const double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int QUAD_PAIR = 0xB1;    // quad_perm:[1,0,3,2]: the pair partner
constexpr int QUAD_OTHER = 0x4E;   // quad_perm:[2,3,0,1]: the same role in the other pair

// (ox, oy) = (bx, by) + c * dA/dz of the lane's wave (stage coefficient c folded into g, tg, ha, E as in yaman_stage)
template <bool LOSS>
__device__ __forceinline__ void quad_stage(const double x, const double y, const double bx, const double by, const double Er,
                                           const double Ei, const double g, const double tg, const double ha, double &ox,
                                           double &oy) {
    const double p = fma_(x, x, y * y);
    const double s1 = p + quad_xchg<QUAD_PAIR>(p);             // the pair's |A_u|^2 + |A_v|^2 (the two-lane kernel's own sum)
    const double s = s1 + quad_xchg<QUAD_OTHER>(s1);           // + the other pair's
    const double gj = fma_(-g, p, tg * s);
    const double X = quad_xchg<QUAD_PAIR>(x), Y = quad_xchg<QUAD_PAIR>(y);
    // A_own * A_partner: the two lanes of a pair form the same product with the roles of their FMA's exact and rounded
    // factor exchanged, so their copies may differ in the last bit (the pumps then see pair products one ulp apart): rounding
    // noise of the size every layout has, not worth the four selects per evaluation that would remove it
    const double qr = fma_(x, X, -(y * Y)), qi = fma_(x, Y, y * X);
    const double Qr = quad_xchg<QUAD_OTHER>(qr), Qi = quad_xchg<QUAD_OTHER>(qi);
    const double Fr = fma_(Er, Qr, -(Ei * Qi)), Fi = fma_(Er, Qi, Ei * Qr);
    if constexpr (LOSS) {
        ox = fma_(Y, Fr, fma_(-X, Fi, fma_(-gj, y, fma_(ha, x, bx))));
        oy = fma_(X, Fr, fma_(Y, Fi, fma_(gj, x, fma_(ha, y, by))));
    } else {
        ox = fma_(Y, Fr, fma_(-X, Fi, fma_(-gj, y, bx)));
        oy = fma_(X, Fr, fma_(Y, Fi, fma_(gj, x, by)));
    }
}

template <int CHECK, bool TRAJ, int BLOCK, bool LOSS>
__global__ void __launch_bounds__(BLOCK) rk4_sweep_quad_kernel(const SweepArgs<double> A) {
    constexpr int NW = 4;
    constexpr int RESYNC = Phase<double>::RESYNC;
    const long long gid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const long long idx = gid >> 2;   // sweep point: the four lanes of a quad share one
    const int role = (int)(gid & 3);  // = the wave this lane holds
    const long long N = A.n_points;
    if (idx >= N) return;             // the four lanes of a quad leave together

    double x = A.a0[(long long)(2 * role) * A.a0_ld + idx * A.a0_stride];
    double y = A.a0[(long long)(2 * role + 1) * A.a0_ld + idx * A.a0_stride];
    const double g = A.gamma[idx * A.gamma_stride];
    const double tg = g + g;
    const double ha = -0.5 * A.alpha[idx * A.alpha_stride];
    const double dbd = (role < 2) ? A.dbeta[idx] : -A.dbeta[idx];   // pumps: E, sidebands: conj(E)

    const double hd = A.z_max / (double)A.n_steps;
    const double hh = 0.5 * hd;
    const double g_d = hh * g, tg_d = hh * tg, ha_d = hh * ha;
    const double g_h = hd * g, tg_h = hd * tg, ha_h = hd * ha;
    const double third = 1.0 / 3.0;
    const double e_amp = tg_d;

    double rc, rs, Er = e_amp, Ei = 0.0;
    Phase<double>::eval(dbd * (0.5 * hd), rc, rs);

    double pe = fma_(x, x, y * y);   // meaningful on the signal's lane (role 2), which writes the summary
    double pm = pe;
    long long bad = -1;
    auto nonfinite_on = [&](const double vx, const double vy) -> bool {   // any component of the POINT non-finite
        double t = fma_(vy, 0.0, fma_(vx, 0.0, 0.0));
        t += quad_xchg<QUAD_PAIR>(t);
        t += quad_xchg<QUAD_OTHER>(t);
        return t != t;
    };

    const int se = A.save_every;
    const int n_rows = A.n_steps / se;
    const int n_run = (CHECK != CHECK_NONE) ? A.n_steps : n_rows * se;

    // trajectory rows [row][wave][ld][2]: the lane's wave is its role, so the address is a wave-uniform row base plus the
    // per-lane constant 32-bit offset (role * ld + idx) * 16 (the C-ABI keeps 4 * ld * 16 B < 2^32 for these launches)
    using Pair = typename PairOf<double>::type;
    const long long LD = A.traj_ld;
    const unsigned lane_off = (unsigned)((unsigned long long)(role * LD + idx) * sizeof(Pair));
    auto store_traj_row = [&](const int r) {
        store_pair_nt(reinterpret_cast<const char *>(A.traj) + (long long)r * NW * LD * (long long)sizeof(Pair), lane_off, Pair{x, y});
    };
    auto store_a_end = [&]() {
        A.a_end[(long long)(2 * role) * N + idx] = x;
        A.a_end[(long long)(2 * role + 1) * N + idx] = y;
    };
    if constexpr (TRAJ) store_traj_row(0);
    if (n_rows == 0) store_a_end();

    auto rk4_step_on = [&](double &x, double &y, double &Er, double &Ei) {
        double x2, y2, x3, y3, x4, y4, dx, dy;
        quad_stage<LOSS>(x, y, x, y, Er, Ei, g_d, tg_d, ha_d, x2, y2);
        rotate(Er, Ei, rc, rs);  // z + h/2
        quad_stage<LOSS>(x2, y2, x, y, Er, Ei, g_d, tg_d, ha_d, x3, y3);
        quad_stage<LOSS>(x3, y3, x, y, Er + Er, Ei + Ei, g_h, tg_h, ha_h, x4, y4);
        const double tx = fma_(2.0, x3, fma_(-4.0, x, x2)) + x4;
        const double ty = fma_(2.0, y3, fma_(-4.0, y, y2)) + y4;
        rotate(Er, Ei, rc, rs);  // z + h
        quad_stage<LOSS>(x4, y4, tx, ty, Er, Ei, g_d, tg_d, ha_d, dx, dy);
        x = fma_(dx, third, x);
        y = fma_(dy, third, y);
    };
    auto rk4_step = [&]() { rk4_step_on(x, y, Er, Ei); };
    auto seed_on = [&](const int step, double &er, double &ei) {
        double c, s;
        Phase<double>::eval(dbd * ((double)step * hd), c, s);
        er = e_amp * c;
        ei = e_amp * s;
    };

    // CHECK_EXACT by replay of the failing block (see rk4_sweep_kernel); the four lanes of a point take the branch together
    constexpr bool REPLAY = CHECK == CHECK_EXACT;
    double x_chk = x, y_chk = y, Er_chk = Er, Ei_chk = Ei;
    int i_chk = 0;
    auto checkpoint = [&](const int step) {
        if constexpr (REPLAY) {
            x_chk = x;
            y_chk = y;
            Er_chk = Er;
            Ei_chk = Ei;
            i_chk = step;
        }
    };
    auto exact_test = [&](const int i_now) {
        if constexpr (REPLAY) {
            const bool newly_bad = bad < 0 && nonfinite_on(x, y);
            if (__builtin_amdgcn_ballot_w64(newly_bad) != 0) {
                double xx = x_chk, yy = y_chk, er = Er_chk, ei = Ei_chk;
                int ii = i_chk;
                while (ii < i_now) {
                    if (ii % RESYNC == 0) seed_on(ii, er, ei);
                    const int to_seed = RESYNC - ii % RESYNC;
                    const int e = (i_now - ii > to_seed) ? ii + to_seed : i_now;
#pragma nounroll
                    for (int st = ii; st < e; ++st) {
                        rk4_step_on(xx, yy, er, ei);
                        if (bad < 0 && nonfinite_on(xx, yy)) bad = st;
                    }
                    ii = e;
                }
            }
            checkpoint(i_now);
        }
    };
    auto write_summary = [&]() {
        if (role == 2) {
            A.p_end[idx] = pe;
            A.p_max[idx] = pm;
        }
        if (role == 0) A.first_bad[idx] = bad;
    };

    // ---- save_every == 1 with a trajectory: every step is a saved row (the dedicated loop of rk4_sweep_kernel)
    if constexpr (TRAJ) {
        if (se == 1) {
            auto save_row = [&](const int r) {
                pe = fma_(x, x, y * y);
                pm = pe > pm ? pe : pm;               // NaN is made to propagate after the loop (it is sticky in y)
                if constexpr (CHECK != CHECK_NONE) {
                    if (bad < 0 && nonfinite_on(x, y)) bad = r - 1;
                }
                store_traj_row(r);
            };
            int i = 0;
            while (i < n_run) {
                seed_on(i, Er, Ei);
                const int end = (n_run - i > RESYNC) ? i + RESYNC : n_run;
                for (; i + 2 <= end; i += 2) {
                    rk4_step();
                    save_row(i + 1);
                    rk4_step();
                    save_row(i + 2);
                }
                if (i < end) {
                    rk4_step();
                    save_row(i + 1);
                    ++i;
                }
            }
            if (pe != pe) pm = pe;
            store_a_end();
            write_summary();
            return;
        }
    }

    // ---- z-loop, event driven, seeds on the absolute grid i = 0, RESYNC, ... (see rk4_sweep_kernel)
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
            rk4_step();
            rk4_step();
            rk4_step();
            rk4_step();
        }
        for (; j + 2 <= m; j += 2) {
            rk4_step();
            rk4_step();
        }
        if (j < m) rk4_step();
        i = end;
        if (i == next_save) {
            ++row;
            pe = fma_(x, x, y * y);
            pm = (pe > pm || pe != pe) ? pe : pm;
            if constexpr (CHECK == CHECK_BLOCK) {
                if (bad < 0 && nonfinite_on(x, y)) bad = i - 1;
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
        if (bad < 0 && n_run > 0 && nonfinite_on(x, y)) bad = n_run - 1;
    }
    if (n_run > i_chk) exact_test(n_run);   // the unsaved tail (CHECK_EXACT only)
    write_summary();
}

template <int CHECK, bool TRAJ>
static hipError_t launch_quad_one(hipStream_t s, bool lossless, int block, const SweepArgs<double> &a) {
    const long long lanes = 4 * a.n_points;
    if (block == 256) {
        const dim3 grid((unsigned)((lanes + 255) / 256));
        if (lossless) hipLaunchKernelGGL((rk4_sweep_quad_kernel<CHECK, TRAJ, 256, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((rk4_sweep_quad_kernel<CHECK, TRAJ, 256, true>), grid, dim3(256), 0, s, a);
    } else {
        const dim3 grid((unsigned)((lanes + 63) / 64));
        if (lossless) hipLaunchKernelGGL((rk4_sweep_quad_kernel<CHECK, TRAJ, 64, false>), grid, dim3(64), 0, s, a);
        else hipLaunchKernelGGL((rk4_sweep_quad_kernel<CHECK, TRAJ, 64, true>), grid, dim3(64), 0, s, a);
    }
    return hipGetLastError();
}

// 4-wave model only.  block: 64 | 256 as for launch_sweep_split.
static hipError_t launch_sweep_quad(hipStream_t s, int check, bool lossless, int block, const SweepArgs<double> &a) {
    if (a.n_points == 0) return hipSuccess;
    const bool traj = a.traj != nullptr;
    switch (check) {
        case CHECK_NONE:
            return traj ? launch_quad_one<CHECK_NONE, true>(s, lossless, block, a) : launch_quad_one<CHECK_NONE, false>(s, lossless, block, a);
        case CHECK_BLOCK:
            return traj ? launch_quad_one<CHECK_BLOCK, true>(s, lossless, block, a) : launch_quad_one<CHECK_BLOCK, false>(s, lossless, block, a);
        default:
            return traj ? launch_quad_one<CHECK_EXACT, true>(s, lossless, block, a) : launch_quad_one<CHECK_EXACT, false>(s, lossless, block, a);
    }
}

}  // namespace psa
