// psa_rk4_pk_kernel.inc.h -- float32 sweep with TWO sweep points per lane (packed math), gfx950.
//
// Why: on this chip a float32 VALU instruction without packing sustains the same one-per-4-cycles issue rate as float64
// (tools/sp_peak.hip: v_fma_f32 tops out at 0.239 wave-instructions per clock per SIMD = 75 TFLOP/s at ANY occupancy), so a
// one-point-per-lane float32 kernel runs no faster than the float64 one.  Packing points (2i, 2i+1) into the two halves of a
// 64-bit register pair turns every instruction of the step into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the same
// instruction count advances twice the points (129.5 TFLOP/s measured with one wave per SIMD -- the shape of BASELINE
// config 4's per-GPU shard -- and 140-147 with four or more).  Same algorithm and arithmetic as the scalar float32 kernel
// (classic low-storage RK4 on the un-fused RHS with a compensated state update, phase factor re-seeded from a
// float64-reduced sincos every 16 steps), so the two agree to rounding.
#pragma once
#include "psa_rk4_kernel.inc.h"

namespace psa {

__device__ __forceinline__ f32x2 splat2(float x) { return (f32x2){x, x}; }

template <int NW, int CHECK, bool TRAJ, int BLOCK>
__global__ void __launch_bounds__(BLOCK) rk4_sweep_pk_kernel(const SweepArgs<float> A) {
    using V = f32x2;
    constexpr int NC = 2 * NW;
    constexpr int NP = (NW - 2) / 2;
    constexpr int RESYNC = Phase<float>::RESYNC;
    const long long idx = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const long long N = A.n_points;
    const long long pt[2] = {2 * idx, (2 * idx + 1 < N) ? 2 * idx + 1 : 2 * idx};  // odd tail: slot 1 mirrors slot 0
    if (pt[0] >= N) return;
    const bool live1 = 2 * idx + 1 < N;  // slot 1 holds a real point (else computed but never stored)

    // wave-uniform: every lane of this wave holds two real points -> 8-B loads / stores of the adjacent pair, 16-B trajectory
    // stores; the (at most one) wave with the odd tail or past-the-end lanes takes the element-wise path
    const bool wave_full = __builtin_amdgcn_readfirstlane((int)(2 * (idx | 63) + 1 < N)) != 0;
    typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));   // an 8-B access at float alignment (odd N rows)
    auto load2 = [&](const float *base, const int stride) -> V {   // base[pt0 * stride], base[pt1 * stride]
        if (stride == 0) return splat2(base[0]);
        if (wave_full) return *reinterpret_cast<const f32x2_u *>(base + pt[0]);
        return (V){base[pt[0]], base[pt[1]]};
    };

    V y[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) y[c] = load2(A.a0 + (long long)c * A.a0_ld, A.a0_stride);
    const V g = load2(A.gamma, A.gamma_stride);
    const V tg = g + g;
    const V ha = splat2(-0.5f) * load2(A.alpha, A.alpha_stride);
    double dbd[NP][2];
    {
        const V d0 = load2(A.dbeta, 1);
        dbd[0][0] = (double)d0.x;
        dbd[0][1] = (double)d0.y;
        if constexpr (NP == 2) {
            const V d1 = load2(A.dbeta2, 1);
            dbd[1][0] = (double)d1.x;
            dbd[1][1] = (double)d1.y;
        }
    }
    const double hd = A.z_max / (double)A.n_steps;
    const V h = splat2((float)hd), hh = splat2((float)(0.5 * hd)), h6 = splat2((float)(hd / 6.0));
    const V two = splat2(2.0f);

    V rc[NP], rs[NP], Er[NP], Ei[NP];
    auto seed = [&](const double z, V (&outc)[NP], V (&outs)[NP], const V amp) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float c0, s0, c1, s1;
            Phase<float>::eval(dbd[p][0] * z, c0, s0);
            Phase<float>::eval(dbd[p][1] * z, c1, s1);
            outc[p] = amp * (V){c0, c1};
            outs[p] = amp * (V){s0, s1};
        }
    };
    seed(0.5 * hd, rc, rs, splat2(1.0f));   // half-step rotator exp(i*dbeta*h/2)
#pragma unroll
    for (int p = 0; p < NP; ++p) { Er[p] = tg; Ei[p] = V{}; }

    // Compensated state.  float32 loses the part of each increment (~1e-5 |y| at 1e6 steps) below ulp(y): plain y += inc
    // drifts ~n * ulp (5e-3 at BASELINE config 4's 1e6 steps).  The state is therefore kept as  yb + dl : a base yb and a
    // SMALL running offset dl that collects the increments (rounded at ulp(dl) ~ 1e-4 ulp(y)); y = fl(yb + dl) is formed once
    // per step for the stage inputs, and every FOLD steps dl is folded into yb with its rounding residue kept (Fast2Sum).
    // 16 + 8 + 24/FOLD instructions per step and component pair against 40 for a Kahan update of y every step.
    constexpr int FOLD = RESYNC;      // folded where the phase is re-seeded
    V yb[NC], dl[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        yb[c] = y[c];
        dl[c] = V{};
    }
    auto fold = [&]() {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const V sum = yb[c] + dl[c];
            dl[c] = dl[c] - (sum - yb[c]);
            yb[c] = sum;
            y[c] = sum;
        }
    };
    V pe = fma_(y[4], y[4], y[5] * y[5]);
    V pm = pe;
    long long bad[2] = {-1, -1};
    auto track = [&](const int step) {  // sum_c 0*y_c is NaN exactly for a non-finite component, per packed half
        V t = V{};
#pragma unroll
        for (int c = 0; c < NC; ++c) t = fma_(y[c], V{}, t);
        if (bad[0] < 0 && t.x != t.x) bad[0] = step;
        if (bad[1] < 0 && t.y != t.y) bad[1] = step;
    };
    auto store_rows = [&](float *base) {  // base[c * N + point]
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float *dst = base + (long long)c * N + pt[0];
            if (wave_full) {
                *reinterpret_cast<f32x2_u *>(dst) = y[c];
            } else {
                dst[0] = y[c].x;
                if (live1) dst[1] = y[c].y;
            }
        }
    };
    auto store2 = [&](float *base, const V v) {   // base[pt0], base[pt1]
        if (wave_full) {
            *reinterpret_cast<f32x2_u *>(base + pt[0]) = v;
        } else {
            base[pt[0]] = v.x;
            if (live1) base[pt[1]] = v.y;
        }
    };

    const int se = A.save_every;
    const int n_rows = A.n_steps / se;
    const int n_run = (CHECK != CHECK_NONE) ? A.n_steps : n_rows * se;
    // trajectory rows [row][wave][N][2]: the lane's two points are adjacent, so each wave of the model is ONE 16-B streaming
    // store per lane (1 KiB per wave instruction); the (row, wave) part of the address stays in SGPRs and the lane adds a
    // 32-bit byte offset (the C-ABI keeps N * 8 B < 2^31 for trajectory launches), exactly as rk4_sweep_kernel does.
    const long long LD = A.traj_ld;   // points per (row, wave) region (psa_traj_ld)
    const unsigned lane_off = (unsigned)idx * 16u;
    auto store_traj_row = [&](const int r) {
        const char *rowb = reinterpret_cast<const char *>(A.traj) + (long long)r * NW * LD * 8;
        if (wave_full) {
#pragma unroll
            for (int j = 0; j < NW; ++j)
                store_quad_nt(rowb + (long long)j * LD * 8, lane_off, (f32x4){y[2 * j].x, y[2 * j + 1].x, y[2 * j].y, y[2 * j + 1].y});
        } else {
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const char *wb = rowb + (long long)j * LD * 8;
                store_pair_nt(wb, lane_off, (f32x2){y[2 * j].x, y[2 * j + 1].x});
                if (live1) store_pair_nt(wb, lane_off + 8u, (f32x2){y[2 * j].y, y[2 * j + 1].y});
            }
        }
    };
    if constexpr (TRAJ) store_traj_row(0);
    if (n_rows == 0) store_rows(A.a_end);

    auto rk4_step = [&](const int step_index) {  // integrators.py:54-59, low storage: y, y_stage, accumulator
        V k[NC], ys[NC], acc[NC];
        yaman_rhs<V, NW>(y, Er, Ei, g, tg, ha, k);
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc[c] = k[c]; ys[c] = fma_(hh, k[c], y[c]); }
#pragma unroll
        for (int p = 0; p < NP; ++p) rotate(Er[p], Ei[p], rc[p], rs[p]);
        yaman_rhs<V, NW>(ys, Er, Ei, g, tg, ha, k);
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc[c] = fma_(two, k[c], acc[c]); ys[c] = fma_(hh, k[c], y[c]); }
        yaman_rhs<V, NW>(ys, Er, Ei, g, tg, ha, k);
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc[c] = fma_(two, k[c], acc[c]); ys[c] = fma_(h, k[c], y[c]); }
#pragma unroll
        for (int p = 0; p < NP; ++p) rotate(Er[p], Ei[p], rc[p], rs[p]);
        yaman_rhs<V, NW>(ys, Er, Ei, g, tg, ha, k);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            dl[c] = fma_(h6, acc[c] + k[c], dl[c]);   // the increment joins the small offset ...
            y[c] = yb[c] + dl[c];                     // ... and y is the rounded state again (next stage input, saved rows)
        }
        if constexpr (CHECK == CHECK_EXACT) track(step_index);
    };

    auto write_summary = [&]() {
        store2(A.p_end, pe);
        store2(A.p_max, pm);
        A.first_bad[pt[0]] = bad[0];
        if (live1) A.first_bad[pt[1]] = bad[1];
    };

    // ---- save_every == 1 with a trajectory: every step is a saved row (integrators.py:137) -- the HBM-bound regime.  A
    // dedicated loop, as in rk4_sweep_kernel: per row only |A_sig|^2, the running maximum, the block-mode finite test and
    // the NW streaming stores; two steps per trip so one row's stores issue under the next step.
    if constexpr (TRAJ) {
        if (se == 1) {
            auto save_row = [&](const int r) {
                pe = fma_(y[4], y[4], y[5] * y[5]);
                pm = (V){__builtin_fmaxf(pe.x, pm.x), __builtin_fmaxf(pe.y, pm.y)};   // NaN is made to propagate below
                if constexpr (CHECK == CHECK_BLOCK) track(r - 1);
                store_traj_row(r);
            };
            int i = 0;
            while (i < n_run) {
                seed((double)i * hd, Er, Ei, tg);
                fold();                               // RESYNC == FOLD steps since the last one
                const int end = (n_run - i > RESYNC) ? i + RESYNC : n_run;
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
            }
            if (pe.x != pe.x) pm.x = pe.x;   // np.max over the saved rows propagates NaN (sticky in y)
            if (pe.y != pe.y) pm.y = pe.y;
            store_rows(A.a_end);
            write_summary();
            return;
        }
    }

    // seeds (and the folds of the compensated state, FOLD == RESYNC) on the absolute grid i = 0, RESYNC, ...: the trajectory
    // does not depend on save_every (see rk4_sweep_kernel)
    static_assert(FOLD == RESYNC, "the state is folded where the phase is re-seeded");
    int i = 0, row = 0;
    int next_save = (n_rows > 0) ? se : 0x7fffffff;
    int next_seed = 0;
    while (i < n_run) {
        if (i == next_seed) {
            seed((double)i * hd, Er, Ei, tg);
            fold();
            next_seed = (n_run - i > RESYNC) ? i + RESYNC : 0x7fffffff;
        }
        int end = n_run < next_seed ? n_run : next_seed;
        end = end < next_save ? end : next_save;
        const int m = end - i;
        int j = 0;
        for (; j + 2 <= m; j += 2) {
            rk4_step(i + j);
            rk4_step(i + j + 1);
        }
        if (j < m) rk4_step(i + j);
        i = end;
        if (i == next_save) {
            ++row;
            pe = fma_(y[4], y[4], y[5] * y[5]);
            pm.x = (pe.x > pm.x || pe.x != pe.x) ? pe.x : pm.x;  // np.max propagates NaN
            pm.y = (pe.y > pm.y || pe.y != pe.y) ? pe.y : pm.y;
            if constexpr (CHECK == CHECK_BLOCK) track(i - 1);
            if constexpr (TRAJ) store_traj_row(row);
            if (row == n_rows) {
                store_rows(A.a_end);
                next_save = 0x7fffffff;
            } else {
                next_save += se;
            }
        }
    }
    if constexpr (CHECK == CHECK_BLOCK) {
        if (n_run > 0) track(n_run - 1);
    }
    write_summary();
}

template <int NW, int CHECK, bool TRAJ>
static hipError_t launch_pk_one(hipStream_t s, int block, const SweepArgs<float> &a) {
    const long long lanes = (a.n_points + 1) / 2;
    if (block == 64) {
        hipLaunchKernelGGL((rk4_sweep_pk_kernel<NW, CHECK, TRAJ, 64>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, a);
    } else {
        hipLaunchKernelGGL((rk4_sweep_pk_kernel<NW, CHECK, TRAJ, 256>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

template <int NW>
static hipError_t launch_pk_nw(hipStream_t s, int check, int block, const SweepArgs<float> &a) {
    const bool traj = a.traj != nullptr;
    switch (check) {
        case CHECK_NONE:
            return traj ? launch_pk_one<NW, CHECK_NONE, true>(s, block, a) : launch_pk_one<NW, CHECK_NONE, false>(s, block, a);
        case CHECK_BLOCK:
            return traj ? launch_pk_one<NW, CHECK_BLOCK, true>(s, block, a) : launch_pk_one<NW, CHECK_BLOCK, false>(s, block, a);
        default:
            return traj ? launch_pk_one<NW, CHECK_EXACT, true>(s, block, a) : launch_pk_one<NW, CHECK_EXACT, false>(s, block, a);
    }
}

static hipError_t launch_sweep_pk(hipStream_t s, int n_waves, int check, int block, const SweepArgs<float> &a) {
    if (a.n_points == 0) return hipSuccess;
    return n_waves == 4 ? launch_pk_nw<4>(s, check, block, a) : launch_pk_nw<6>(s, check, block, a);
}

}  // namespace psa
