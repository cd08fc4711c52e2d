// psa_rk4_kernel.inc.h -- the RK4 sweep kernel for gfx950 (included by psa_rk4_f64.hip / psa_rk4_f32.hip).
//
// One sweep point per lane, the whole z-loop inside the kernel, state in VGPRs.
// Replaces, per point:  integrators.integrate_fixed_step (integrators.py:68-142) driving
// integrators.rk4_step (:25-61) on yaman_model.rhs_yaman_simplified (yaman_model.py:10-52),
// plus the saved-row reduction of the sweep drivers (scan_mismtach.py:376-381).
//
// What bounds it: FP64 vector FMA issue (16 lanes/clk/SIMD on CDNA4), NOT HBM and not MFMA:
// a point reads 8..88 B and writes 88 B for its entire z-loop, and the RHS is an elementwise
// complex polynomial (no contraction to tile).  So the design rules here are
//   * minimum DP instructions per step (298 for 4 waves; see the count in DESIGN.md),
//   * no transcendental in the steady-state loop: E(z) = 2*gamma*exp(i*dbeta*z) is carried by a
//     complex rotation per half step and re-seeded from an exact sincos at every multiple of RESYNC steps
//     (64 in float64: drift <= 128 multiplications ~1.4e-14, far inside the 1e-9 parity budget),
//   * all per-lane arrays statically indexed and in VGPRs (178 for the bench instantiation: 2 waves/SIMD; capping
//     at 168 (3 waves) or 128 (4 waves, 16 spilled) was measured no faster -- the loop is issue-bound),
//   * wave-uniform control flow only (save stride, resync and NaN tracking never diverge),
//   * SoA global layout: every load/store instruction of a wave is one contiguous 512-B run.
//
// The independent variable follows the reference grid: z_i = i * (z_max / n_steps)
// (np.linspace, integrators.py:195) formed from the INTEGER step index, never accumulated.
#pragma once
#include "psa_internal.h"

namespace psa {

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// two float32 sweep points per lane: <2 x float> arithmetic selects v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma_(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// (re, im) as one naturally aligned 2-element vector: 16-B (f64) / 8-B (f32) global stores
template <typename T> struct PairOf;
template <> struct PairOf<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct PairOf<float> { typedef float type __attribute__((ext_vector_type(2))); };

// Streaming store of one (re, im) pair at  sbase + voff : sbase wave-uniform (an SGPR pair), voff the lane's 32-bit byte
// offset.  This is the global_store "saddr" form; written as inline assembly because the compiler, left to itself, widens
// the lane offset to 64 bits inside the z-loop and then spends a v_lshl_add_u64 per store on the address.
// HAZARD: on gfx90a / gfx940 / gfx950 a VMEM store of MORE than 64 bits of data must not be followed within two wait
// states by a VALU instruction that overwrites the data VGPRs (LLVM's GCNHazardRecognizer inserts the s_nop for stores it
// can see; it cannot see into inline assembly).  The 16-B forms therefore carry their own two wait states (as two 4-byte
// `s_nop 0`, so that the 8-byte encodings after them stay on 8-byte boundaries): without them the packed
// float32 kernel, which assembles each store's four floats in a temporary it reuses at once, wrote corrupt rows.
#ifndef PSA_TRAJ_F64_MOD      // A/B hook (tools/ab_build.sh): -DPSA_TRAJ_F64_MOD='""' = default (write-back) stores
#define PSA_TRAJ_F64_MOD " nt"
#endif
__device__ __forceinline__ void store_pair_nt(const void *sbase, const unsigned voff, const PairOf<double>::type v) {
    asm volatile("global_store_dwordx4 %0, %1, %2" PSA_TRAJ_F64_MOD "\n\ts_nop 0\n\ts_nop 0" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void store_pair_nt(const void *sbase, const unsigned voff, const PairOf<float>::type v) {
    asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
// two adjacent float32 points' (re, im) pairs in one 16-B store (the packed kernel: points 2i and 2i+1 share a lane)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_quad_nt(const void *sbase, const unsigned voff, const f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0\n\ts_nop 0" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}

// cos/sin of a float64 phase, delivered in the working precision.
template <typename T> struct Phase;
template <> struct Phase<double> {
    static constexpr int RESYNC = 64;  // steps after which the rotation recurrence is re-seeded exactly
    static __device__ __forceinline__ void eval(double ph, double &c, double &s) { sincos(ph, &s, &c); }
};
#ifndef PSA_F32_RESYNC          // A/B hook (tools/ab_build_f32.sh)
#define PSA_F32_RESYNC 16   // 20 / 32 / 64 measured within 2 % of each other on the config-4 shard; accuracy guard identical
#endif
template <> struct Phase<float> {
    static constexpr int RESYNC = PSA_F32_RESYNC;
    // dbeta*z reaches 1e4..1e5 rad at 1e6 steps: reduce in f64, then an f32 sincos on [-pi, pi].
    static __device__ __forceinline__ void eval(double ph, float &c, float &s) {
        const double r = ph - 6.283185307179586 * rint(ph * 0.15915494309189535);
        sincosf((float)r, &s, &c);
    }
};

// ---- right-hand side, fused with the Runge-Kutta stage update -------------------------------------------
// a  = [Re A1, Im A1, Re A2, ...];  returns  out = base + c * dA/dz(a)   (FUSED)   or   out = dA/dz(a)  (c = 1).
// The stage coefficient c never appears as an instruction: the caller passes it folded into the constants
//   g = c*gamma, tg = 2*c*gamma, ha = -c*alpha/2, (Er, Ei) = 2*c*gamma*exp(+i*dbeta*z)  per sideband pair,
// and `base` simply seeds the FMA chain that the un-fused form would start with a multiply.
//
//   dA_j/dz = (-alpha/2 + i*gamma*f_j) A_j + i*conj(partner) * F            (yaman_model.py:123-186)
//   f_j = P_j + 2*sum_{k!=j} P_k = 2*S - P_j                                 (yaman_model.py:148-151)
//   pumps:     F = 2*gamma*exp(+i dbeta z) * (A_s A_i)                        (yaman_model.py:174,177-178)
//   sidebands: F = 2*gamma*exp(-i dbeta z) * (A_p1 A_p2)                      (yaman_model.py:175,180-181)
// 64 DP instructions for NW = 4 either way (p: 8, S/g_j: 8, two products: 8, two F: 8, eight 4-deep chains: 32).
// LOSS = false is the reference's own `alpha == 0.0` branch (_linear_loss_terms returns zeros, yaman_model.py:130-131):
// the -alpha/2 links disappear from all 2*NW chains (8 instructions per evaluation for 4 waves).
template <typename T, int NW, bool FUSED, bool LOSS = true>
__device__ __forceinline__ void yaman_stage(const T (&a)[2 * NW], const T (&base)[2 * NW],
                                            const T (&Er)[(NW - 2) / 2], const T (&Ei)[(NW - 2) / 2], const T g,
                                            const T tg, const T ha, T (&out)[2 * NW]) {
    constexpr int NP = (NW - 2) / 2;
    T p[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) p[j] = fma_(a[2 * j], a[2 * j], a[2 * j + 1] * a[2 * j + 1]);
    T s = (p[0] + p[1]) + (p[2] + p[3]);
    if constexpr (NW == 6) s += (p[4] + p[5]);
    const T gs = tg * s;  // c*gamma * 2S
    T gj[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) gj[j] = fma_(-g, p[j], gs);  // c*gamma * f_j

    // first two links of each chain: gsig * v  +  [ha * component]  [+ base]
    auto link = [&](const T gsig, const T v, const int c) -> T {
        if constexpr (LOSS) return fma_(gsig, v, FUSED ? fma_(ha, a[c], base[c]) : ha * a[c]);
        else return FUSED ? fma_(gsig, v, base[c]) : gsig * v;
    };

    const T x1 = a[0], y1 = a[1], x2 = a[2], y2 = a[3];
    const T q12r = fma_(x1, x2, -(y1 * y2)), q12i = fma_(x1, y2, y1 * x2);  // A1*A2

    T Fpr = T{}, Fpi = T{};  // sum over pairs of E_p * (A_s A_i): drives both pumps
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
        const int cs = 4 + 4 * pr;  // component index of Re A_signal of this pair
        const T xs = a[cs], ys = a[cs + 1], xi = a[cs + 2], yi = a[cs + 3];
        const T qr = fma_(xs, xi, -(ys * yi)), qi = fma_(xs, yi, ys * xi);  // A_s*A_i
        if (pr == 0) {
            Fpr = fma_(Er[pr], qr, -(Ei[pr] * qi));
            Fpi = fma_(Er[pr], qi, Ei[pr] * qr);
        } else {
            Fpr = fma_(Er[pr], qr, fma_(-Ei[pr], qi, Fpr));
            Fpi = fma_(Er[pr], qi, fma_(Ei[pr], qr, Fpi));
        }
        // conj(E_p) * (A1 A2): drives this pair's signal and idler
        const T Fsr = fma_(Er[pr], q12r, Ei[pr] * q12i);
        const T Fsi = fma_(Er[pr], q12i, -(Ei[pr] * q12r));
        const T gS = gj[2 + 2 * pr], gI = gj[3 + 2 * pr];
        // signal: (ha + i gS) A_s + i conj(A_i) Fs
        out[cs] = fma_(yi, Fsr, fma_(-xi, Fsi, link(-gS, ys, cs)));
        out[cs + 1] = fma_(xi, Fsr, fma_(yi, Fsi, link(gS, xs, cs + 1)));
        // idler:  (ha + i gI) A_i + i conj(A_s) Fs
        out[cs + 2] = fma_(ys, Fsr, fma_(-xs, Fsi, link(-gI, yi, cs + 2)));
        out[cs + 3] = fma_(xs, Fsr, fma_(ys, Fsi, link(gI, xi, cs + 3)));
    }
    // pump1: (ha + i g1) A1 + i conj(A2) Fp ;  pump2: (ha + i g2) A2 + i conj(A1) Fp
    out[0] = fma_(y2, Fpr, fma_(-x2, Fpi, link(-gj[0], y1, 0)));
    out[1] = fma_(x2, Fpr, fma_(y2, Fpi, link(gj[0], x1, 1)));
    out[2] = fma_(y1, Fpr, fma_(-x1, Fpi, link(-gj[1], y2, 2)));
    out[3] = fma_(x1, Fpr, fma_(y1, Fpi, link(gj[1], x2, 3)));
}

// plain dA/dz (used by the LDS-staged A/B variant, which keeps k1..k4 as such)
template <typename T, int NW, bool LOSS = true>
__device__ __forceinline__ void yaman_rhs(const T (&a)[2 * NW], const T (&Er)[(NW - 2) / 2],
                                          const T (&Ei)[(NW - 2) / 2], const T g, const T tg, const T ha,
                                          T (&k)[2 * NW]) {
    yaman_stage<T, NW, false, LOSS>(a, a, Er, Ei, g, tg, ha, k);
}

// (Er,Ei) *= (rc,rs)
template <typename T>
__device__ __forceinline__ void rotate(T &Er, T &Ei, const T rc, const T rs) {
    const T nr = fma_(Er, rc, -(Ei * rs));
    const T ni = fma_(Er, rs, Ei * rc);
    Er = nr;
    Ei = ni;
}

// true iff any component is NaN/Inf: x*0 is NaN exactly for non-finite x.
template <typename T, int NC>
__device__ __forceinline__ bool any_nonfinite(const T (&y)[NC]) {
    T t = T(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) t = fma_(y[c], T(0), t);
    return t != t;
}

// ---- the sweep kernel ------------------------------------------------------------------------------
// LDS = true is the layout the north-star sketches (state and k1..k4 staged in LDS, [component][lane] so a wave's
// ds_read/ds_write_b64 touches 64 consecutive 8-byte words: conflict-free).  It exists for the A/B in DESIGN.md
// section 5: the register-resident form wins because the LDS round trips buy nothing (no data is shared
// between lanes) and cost issue slots next to an already saturated FP64 pipe.
#ifndef PSA_SWEEP_KERNEL_ATTR   // A/B hook (tools/ab_build.sh): e.g. -DPSA_SWEEP_KERNEL_ATTR='__attribute__((amdgpu_waves_per_eu(3)))'
#define PSA_SWEEP_KERNEL_ATTR
#endif
template <typename T, int NW, int CHECK, bool TRAJ, int BLOCK, bool LDS = false, bool LOSS = true>
__global__ void __launch_bounds__(BLOCK) PSA_SWEEP_KERNEL_ATTR rk4_sweep_kernel(const SweepArgs<T> A) {
    constexpr int NC = 2 * NW;
    constexpr int NP = (NW - 2) / 2;
    constexpr int RESYNC = Phase<T>::RESYNC;
    const long long idx = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const long long N = A.n_points;
    if (idx >= N) return;

    // -- per-point inputs: one coalesced load per array (512 B per wave instruction in f64)
    T y[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) y[c] = A.a0[(long long)c * A.a0_ld + idx * A.a0_stride];
    const T g = A.gamma[idx * A.gamma_stride];
    const T tg = g + g;
    const T ha = T(-0.5) * A.alpha[idx * A.alpha_stride];
    double dbd[NP];
    dbd[0] = (double)A.dbeta[idx];
    if constexpr (NP == 2) dbd[1] = (double)A.dbeta2[idx];

    const double hd = A.z_max / (double)A.n_steps;  // np.linspace step
    const T h = (T)hd, hh = (T)(0.5 * hd), h6 = (T)(hd / 6.0);
    // stage coefficients folded into the physics constants: d = h/2 (stages 1, 2, 4) and h (stage 3)
    const T g_d = hh * g, tg_d = hh * tg, ha_d = hh * ha;
    const T g_h = h * g, tg_h = h * tg, ha_h = h * ha;
    const T third = T(1.0 / 3.0);
    constexpr bool FUSE = (sizeof(T) == 8) && !LDS;   // fused-stage step: float64 register variant only
    const T e_amp = FUSE ? tg_d : tg;  // modulus of the carried phase factor: 2*d*gamma (fused) or 2*gamma

    T rc[NP], rs[NP], Er[NP], Ei[NP];  // half-step rotator and the running 2*gamma*exp(i dbeta z)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        Phase<T>::eval(dbd[p] * (0.5 * hd), rc[p], rs[p]);
        Er[p] = e_amp;
        Ei[p] = T(0);
    }

    T pe = fma_(y[4], y[4], y[5] * y[5]);  // |A_sig|^2 at the last saved row (z = 0 is a saved row)
    T pm = pe;                             // np.max over saved rows
    long long bad = -1;

    const int se = A.save_every;
    const int n_rows = A.n_steps / se;                                 // saved rows after z = 0
    const int n_run = (CHECK != CHECK_NONE) ? A.n_steps : n_rows * se;  // the tail only matters for check_nan

    // trajectory rows: device layout [row][wave][N][2] -- each lane stores one (re, im) pair = 16 B (f64), so a wave
    // instruction writes 1 KiB contiguously (the widest coalesced store; half the store instructions of per-component rows).
    // The wave-uniform part of the address (row, wave) stays in SGPRs and the lane contributes a 32-bit byte offset
    // (global_store ... saddr form): no vector instruction is spent on addressing.  The C-ABI keeps N * sizeof(Pair) < 2^32
    // for trajectory launches.
    using Pair = typename PairOf<T>::type;
    const long long LD = A.traj_ld;   // points per (row, wave) region: N, or N padded off a power of two (psa_traj_ld)
    const unsigned lane_off = (unsigned)idx * (unsigned)sizeof(Pair);
    auto store_traj_row = [&](const int r) {
        const char *rowb = reinterpret_cast<const char *>(A.traj) + (long long)r * NW * LD * (long long)sizeof(Pair);
#pragma unroll
        for (int j = 0; j < NW; ++j)
            store_pair_nt(rowb + (long long)j * LD * (long long)sizeof(Pair), lane_off, Pair{y[2 * j], y[2 * j + 1]});
    };
    if constexpr (TRAJ) store_traj_row(0);
    if (n_rows == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) A.a_end[(long long)c * N + idx] = y[c];
    }

    // LDS-staged variant: sm_k[stage][component][lane] and sm_y[component][lane] (volatile: the traffic is the point)
    __shared__ T sm_store[LDS ? 5 * NC * BLOCK : 1];
    volatile T *sm_y = sm_store + threadIdx.x;
    volatile T *sm_k = sm_store + NC * BLOCK + threadIdx.x;
    if constexpr (LDS) {
#pragma unroll
        for (int c = 0; c < NC; ++c) sm_y[c * BLOCK] = y[c];
    }
    auto rk4_step_lds = [&](const int step_index) {
        T k[NC], ys[NC];
        const T coef[3] = {hh, hh, h};
#pragma unroll
        for (int c = 0; c < NC; ++c) ys[c] = sm_y[c * BLOCK];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            yaman_rhs<T, NW>(ys, Er, Ei, g, tg, ha, k);
#pragma unroll
            for (int c = 0; c < NC; ++c) sm_k[(st * NC + c) * BLOCK] = k[c];
            if (st == 0 || st == 2) {
#pragma unroll
                for (int p = 0; p < NP; ++p) rotate(Er[p], Ei[p], rc[p], rs[p]);
            }
            if (st < 3) {
#pragma unroll
                for (int c = 0; c < NC; ++c) ys[c] = fma_(coef[st], (T)sm_k[(st * NC + c) * BLOCK], (T)sm_y[c * BLOCK]);
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const T k1 = sm_k[(0 * NC + c) * BLOCK], k2 = sm_k[(1 * NC + c) * BLOCK];
            const T k3 = sm_k[(2 * NC + c) * BLOCK], k4 = sm_k[(3 * NC + c) * BLOCK];
            y[c] = fma_(h6, fma_(T(2), k3, fma_(T(2), k2, k1)) + k4, (T)sm_y[c * BLOCK]);
            sm_y[c * BLOCK] = y[c];
        }
        if constexpr (CHECK == CHECK_EXACT) {
            if (bad < 0 && any_nonfinite<T, NC>(y)) bad = step_index;
        }
    };

    // ---- one classic RK4 step (integrators.py:54-59) in 298 DP instructions (4 waves).
    // Each stage's axpy is folded into the RHS chains (yaman_stage, FUSED): with d = h/2
    //     Y2 = y + d*f(z, y)          Y3 = y + d*f(z+d, Y2)          Y4 = y + 2d*f(z+d, Y3)
    //     t  = Y2 + 2*Y3 + Y4 - 4*y                 ( = d*k1 + 2d*k2 + 2d*k3 )
    //     D  = t + d*f(z+h, Y4)                     ( = d*(k1 + 2*k2 + 2*k3 + k4) )
    //     y <- y + D/3                              ( = y + h/6*(k1 + 2*k2 + 2*k3 + k4) )
    // which is the reference's k1..k4 combination regrouped (no change of variables, same truncation error; the
    // regrouping costs ~1 ulp(y) of rounding noise per step, ~1e-13 after 1e5 steps).  (Ed_r, Ed_i) carries
    // 2*d*gamma*exp(i*dbeta*z): on entry at z_step, on exit rotated to z_step + h.
    auto rk4_step_on = [&](T (&y)[NC], T (&Er)[NP], T (&Ei)[NP]) {
        T Y2[NC], Y3[NC], Y4[NC], t[NC], D[NC];
        yaman_stage<T, NW, true, LOSS>(y, y, Er, Ei, g_d, tg_d, ha_d, Y2);  // Y2 = y + d k1
#pragma unroll
        for (int p = 0; p < NP; ++p) rotate(Er[p], Ei[p], rc[p], rs[p]);  // z + h/2
        yaman_stage<T, NW, true, LOSS>(Y2, y, Er, Ei, g_d, tg_d, ha_d, Y3);  // Y3 = y + d k2
        T E2r[NP], E2i[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            E2r[p] = Er[p] + Er[p];
            E2i[p] = Ei[p] + Ei[p];
        }
        yaman_stage<T, NW, true, LOSS>(Y3, y, E2r, E2i, g_h, tg_h, ha_h, Y4);  // Y4 = y + h k3
#pragma unroll
        for (int c = 0; c < NC; ++c) t[c] = fma_(T(2), Y3[c], fma_(T(-4), y[c], Y2[c])) + Y4[c];
#pragma unroll
        for (int p = 0; p < NP; ++p) rotate(Er[p], Ei[p], rc[p], rs[p]);  // z + h
        yaman_stage<T, NW, true, LOSS>(Y4, t, Er, Ei, g_d, tg_d, ha_d, D);  // D = t + d k4
#pragma unroll
        for (int c = 0; c < NC; ++c) y[c] = fma_(D[c], third, y[c]);
    };
    // The per-step finite test of the reference (integrators.py:132-135) is NOT in the float64 step: CHECK_EXACT finds the
    // exact index by REPLAY (below) -- the forward pass tests once per saved row, like CHECK_BLOCK.
    auto rk4_step_reg = [&](const int) { rk4_step_on(y, Er, Ei); };

    // ---- float32: the classic low-storage form (y, y_stage, accumulator; 320 instructions).  The regrouping above
    // quantises every stage increment to ulp(y); harmless at 1e-16 but measured 17x worse at float32 (6.7e-3 vs
    // 3.8e-4 relative after 1e4 steps), so single precision keeps k1..k4 at full precision.
    T y_lo[NC];  // Kahan residue of the state (float32 path only)
#pragma unroll
    for (int c = 0; c < NC; ++c) y_lo[c] = T{};
    auto rk4_step_classic = [&](const int step_index) {
        T k[NC], ys[NC], acc[NC];
        yaman_rhs<T, NW, LOSS>(y, Er, Ei, g, tg, ha, k);  // k1 at z
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[c] = k[c];
            ys[c] = fma_(hh, k[c], y[c]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) rotate(Er[p], Ei[p], rc[p], rs[p]);  // z + h/2
        yaman_rhs<T, NW, LOSS>(ys, Er, Ei, g, tg, ha, k);  // k2
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[c] = fma_(T(2), k[c], acc[c]);
            ys[c] = fma_(hh, k[c], y[c]);
        }
        yaman_rhs<T, NW, LOSS>(ys, Er, Ei, g, tg, ha, k);  // k3
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[c] = fma_(T(2), k[c], acc[c]);
            ys[c] = fma_(h, k[c], y[c]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) rotate(Er[p], Ei[p], rc[p], rs[p]);  // z + h
        yaman_rhs<T, NW, LOSS>(ys, Er, Ei, g, tg, ha, k);  // k4
        // compensated (Kahan) state update: keeps the part of the increment that y + inc rounds away (see the
        // packed kernel); without it float32 drifts ~n * ulp and misses its 1e-3 tolerance at 1e6 steps.
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const T inc = fma_(h6, acc[c] + k[c], y_lo[c]);
            const T sum = y[c] + inc;
            y_lo[c] = inc - (sum - y[c]);
            y[c] = sum;
        }
        if constexpr (CHECK == CHECK_EXACT) {
            if (bad < 0 && any_nonfinite<T, NC>(y)) bad = step_index;
        }
    };
    auto rk4_step = [&](const int step_index) {
        if constexpr (FUSE) rk4_step_reg(step_index);
        else rk4_step_classic(step_index);
    };

    auto write_summary = [&]() {
        A.p_end[idx] = pe;
        A.p_max[idx] = pm;
        A.first_bad[idx] = bad;
    };
    auto seed_phase_on = [&](const int step, T (&Er)[NP], T (&Ei)[NP]) {   // exact re-seed of the phase recurrence at z = step * h
        const double z = (double)step * hd;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            T c, s;
            Phase<T>::eval(dbd[p] * z, c, s);
            Er[p] = e_amp * c;
            Ei[p] = e_amp * s;
        }
    };
    auto seed_phase = [&](const int step) { seed_phase_on(step, Er, Ei); };

    // ---- CHECK_EXACT for the float64 register variant: exact first_bad_step at the price of the block test.  The state at
    // the last test point (y and the carried phase factor) is kept; when a test finds a lane of the
    // wave newly non-finite, the steps since then are REPLAYED on a copy with the reference's per-step test
    // (integrators.py:132-135).  The replay repeats the forward pass operation for operation (same chunks, same re-seeds,
    // same FMA sequence), so it reproduces this kernel's own trajectory bit for bit and the index it finds is exact.  Only
    // waves with a failing lane ever take the (wave-uniform) branch: +20 VGPRs, no instruction in the steady-state loop
    // (the per-step test cost 9.9 of 310.6 instructions per step, profiles/r03_c2x_pmc.csv).
    constexpr bool REPLAY = FUSE && CHECK == CHECK_EXACT;
    T y_chk[REPLAY ? NC : 1], Er_chk[REPLAY ? NP : 1], Ei_chk[REPLAY ? NP : 1];
    int i_chk = 0;
    auto checkpoint = [&](const int step) {
        if constexpr (REPLAY) {
#pragma unroll
            for (int c = 0; c < NC; ++c) y_chk[c] = y[c];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                Er_chk[p] = Er[p];
                Ei_chk[p] = Ei[p];
            }
            i_chk = step;
        }
    };
    auto exact_test = [&](const int i_now) {   // at a test point: y is the state after step i_now - 1
        if constexpr (REPLAY) {
            const bool newly_bad = bad < 0 && any_nonfinite<T, NC>(y);
            if (__builtin_amdgcn_ballot_w64(newly_bad) != 0) {
                T yy[NC], er[NP], ei[NP];
#pragma unroll
                for (int c = 0; c < NC; ++c) yy[c] = y_chk[c];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    er[p] = Er_chk[p];
                    ei[p] = Ei_chk[p];
                }
                int ii = i_chk;
                while (ii < i_now) {
                    if (ii % RESYNC == 0) seed_phase_on(ii, er, ei);      // the forward pass seeds at the same steps
                    const int to_seed = RESYNC - ii % RESYNC;
                    const int e = (i_now - ii > to_seed) ? ii + to_seed : i_now;
#pragma nounroll
                    for (int st = ii; st < e; ++st) {
                        rk4_step_on(yy, er, ei);
                        if (bad < 0 && any_nonfinite<T, NC>(yy)) bad = st;
                    }
                    ii = e;
                }
            }
            checkpoint(i_now);
        }
    };

    // ---- save_every == 1 with a trajectory: EVERY step is a saved row (integrators.py:137), the path's HBM-bound regime
    // (64 B per point per step against ~300 FP64 instructions: right at the ridge).  A dedicated loop keeps the per-row
    // work to what the row needs -- |A_sig|^2, a running maximum, the block-mode finite test, four streaming stores -- with
    // no event bookkeeping between steps, two steps per trip so the stores of one row issue under the next step.
    if constexpr (TRAJ && !LDS) {
        if (se == 1) {
            auto save_row = [&](const int r) {
                pe = fma_(y[4], y[4], y[5] * y[5]);
                pm = pe > pm ? pe : pm;               // NaN is made to propagate after the loop (it is sticky in y)
                if constexpr (CHECK == CHECK_BLOCK || (CHECK == CHECK_EXACT && FUSE)) {   // a row is a step here: exact either way
                    if (bad < 0 && any_nonfinite<T, NC>(y)) bad = r - 1;
                }
                store_traj_row(r);
            };
            int i = 0;
            while (i < n_run) {                       // n_run == n_steps == n_rows
                seed_phase(i);
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
            if (pe != pe) pm = pe;                    // np.max over the saved rows propagates NaN
#pragma unroll
            for (int c = 0; c < NC; ++c) A.a_end[(long long)c * N + idx] = y[c];
            write_summary();
            return;
        }
    }

    // ---- z-loop, event driven: the steps between two events (a saved row, a phase re-seed, the end) run in a
    // branch-free 2x-unrolled inner loop.  With one wave per SIMD (65 536 points fill the chip exactly once) every
    // taken branch is an exposed instruction refetch, so per-step `if`s cost ~6 % -- see DESIGN.md section 5.
    // Seeds fall on the ABSOLUTE grid i = 0, RESYNC, 2*RESYNC, ... whatever save_every is (the save_every == 1 loop above does
    // the same), so the computed trajectory does not depend on which rows are saved -- as upstream, where the stride only
    // selects rows (integrators.py:137-140): A[-1] at any stride equals the same row of the every-step run bit for bit.
    int i = 0;
    int row = 0;
    checkpoint(0);
    int next_save = (n_rows > 0) ? se : 0x7fffffff;
    int next_seed = 0;
    while (i < n_run) {
        if (i == next_seed) {          // wave-uniform: exact re-seed of the phase recurrence at z_i = i*h
            seed_phase(i);
            next_seed = (n_run - i > RESYNC) ? i + RESYNC : 0x7fffffff;   // no overflow near 2^31 steps
        }
        int end = n_run < next_seed ? n_run : next_seed;
        end = end < next_save ? end : next_save;
        const int m = end - i;
        int j = 0;
        if constexpr (LDS) {
            for (; j < m; ++j) rk4_step_lds(i + j);
        } else {
            for (; j + 2 <= m; j += 2) {
                rk4_step(i + j);
                rk4_step(i + j + 1);
            }
            if (j < m) rk4_step(i + j);
        }
        i = end;
        if (i == next_save) {  // (i % save_every == 0), integrators.py:137 -- wave-uniform
            ++row;
            pe = fma_(y[4], y[4], y[5] * y[5]);
            pm = (pe > pm || pe != pe) ? pe : pm;  // np.max propagates NaN
            if constexpr (CHECK == CHECK_BLOCK) {
                if (bad < 0 && any_nonfinite<T, NC>(y)) bad = i - 1;
            }
            exact_test(i);
            if constexpr (TRAJ) store_traj_row(row);
            if (row == n_rows) {  // A[-1]: the last saved row, not necessarily z_max (R8)
#pragma unroll
                for (int c = 0; c < NC; ++c) A.a_end[(long long)c * N + idx] = y[c];
                next_save = 0x7fffffff;
            } else {
                next_save += se;
            }
        }
    }
    if constexpr (CHECK == CHECK_BLOCK) {  // covers the unsaved tail
        if (bad < 0 && n_run > 0 && any_nonfinite<T, NC>(y)) bad = n_run - 1;
    }
    if (n_run > i_chk) exact_test(n_run);   // the unsaved tail (REPLAY only; compiled out otherwise)
    write_summary();
}

template <typename T, int NW, int CHECK, bool TRAJ>
static hipError_t launch_one(hipStream_t s, int block, bool lds, bool lossless, const SweepArgs<T> &a) {
    if (a.n_points == 0) return hipSuccess;
    if (lossless && !lds) {  // alpha == 0 for every point (caller's promise): the 8 loss links per RHS are compiled out
        if (block == 64) {
            hipLaunchKernelGGL((rk4_sweep_kernel<T, NW, CHECK, TRAJ, 64, false, false>), dim3((unsigned)((a.n_points + 63) / 64)),
                               dim3(64), 0, s, a);
        } else {
            hipLaunchKernelGGL((rk4_sweep_kernel<T, NW, CHECK, TRAJ, 256, false, false>),
                               dim3((unsigned)((a.n_points + 255) / 256)), dim3(256), 0, s, a);
        }
        return hipGetLastError();
    }
    if (lds) {  // A/B variant: one wave per workgroup, 5 * 2*NW * 64 * sizeof(T) bytes of LDS (20 KB for f64, 4 waves)
        const unsigned grid = (unsigned)((a.n_points + 63) / 64);
        hipLaunchKernelGGL((rk4_sweep_kernel<T, NW, CHECK, TRAJ, 64, true>), dim3(grid), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    if (block == 64) {
        const unsigned grid = (unsigned)((a.n_points + 63) / 64);
        hipLaunchKernelGGL((rk4_sweep_kernel<T, NW, CHECK, TRAJ, 64>), dim3(grid), dim3(64), 0, s, a);
    } else {
        const unsigned grid = (unsigned)((a.n_points + 255) / 256);
        hipLaunchKernelGGL((rk4_sweep_kernel<T, NW, CHECK, TRAJ, 256>), dim3(grid), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

template <typename T, int NW>
static hipError_t launch_nw(hipStream_t s, int check, int block, bool lds, bool lossless, const SweepArgs<T> &a) {
    const bool traj = a.traj != nullptr;
    switch (check) {
        case CHECK_NONE:
            return traj ? launch_one<T, NW, CHECK_NONE, true>(s, block, lds, lossless, a)
                        : launch_one<T, NW, CHECK_NONE, false>(s, block, lds, lossless, a);
        case CHECK_BLOCK:
            return traj ? launch_one<T, NW, CHECK_BLOCK, true>(s, block, lds, lossless, a)
                        : launch_one<T, NW, CHECK_BLOCK, false>(s, block, lds, lossless, a);
        default:
            return traj ? launch_one<T, NW, CHECK_EXACT, true>(s, block, lds, lossless, a)
                        : launch_one<T, NW, CHECK_EXACT, false>(s, block, lds, lossless, a);
    }
}

template <typename T>
static hipError_t launch_sweep_t(hipStream_t s, int n_waves, int check, bool lds, int block, bool lossless,
                                 const SweepArgs<T> &a) {
    if (n_waves == 4) return launch_nw<T, 4>(s, check, block, lds, lossless, a);
    return launch_nw<T, 6>(s, check, block, lds, lossless, a);
}

}  // namespace psa
