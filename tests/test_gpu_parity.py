"""GPU parity tests (run on the MI355X box with ``-m gpu``).  Every call goes through the C-ABI of
libpsa_hip.so; the checker is the CPU oracle (pinned by tests/test_oracle_golden.py) and the golden vectors
produced by the reference itself.

Tolerances (north-star: float64 within 1e-9 relative on final amplitudes and gain):
  RTOL_F64 = 1e-9 elementwise on A_end / |A3|^2 / linear gain;  ATOL_DB = 5e-9 dB on gain in dB
  (= 10*log10(1 + 1e-9)).  Measured agreement is ~1e-12.  float32: RTOL_F32 = 1e-4 (build-defined).
"""
import numpy as np
import pytest

import psa_amd._native as nat
from conftest import ATOL_DB, RTOL_F32, RTOL_F64, rel_err

pytestmark = pytest.mark.gpu

P_IN = np.array([0.5, 0.5, 1e-5, 1e-5])
A0 = np.sqrt(P_IN).astype(complex)


def _a0(p, ph=None):
    a = np.sqrt(np.asarray(p, float)).astype(complex)
    return a if ph is None or not np.any(np.asarray(ph) != 0) else a * np.exp(1j * np.asarray(ph))


def test_native_library_is_the_one_in_tree_and_sees_the_gpu():
    assert nat.LIB_PATH.endswith("psa-simulation-ode-rk-mvp-dispersion_amd/libpsa_hip.so")
    assert nat.device_count() >= 1


# ---- sweep kernel vs oracle: ragged sizes, strides, step counts ---------------------------------------------
@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 257, 1000])
@pytest.mark.parametrize("n,se", [(1000, 10), (1005, 10), (37, 1), (7, 10)])
def test_sweep_matches_oracle_on_ragged_sizes(oracle, N, n, se):
    rng = np.random.default_rng(100 * N + n)
    db = rng.uniform(-0.1, 0.1, N)
    ref = oracle.sweep(db, z_max=100.0, n=n, save_every=se, gamma=0.0115, alpha=1.15e-4, a0=A0)
    # one lane per point (256- and 64-thread workgroups), two lanes per point, and the library's own choice
    for flags in (nat.OPT_ONE_LANE, nat.OPT_ONE_LANE | nat.OPT_BLOCK64, nat.OPT_SPLIT_POINT, nat.OPT_SPLIT_POINT | nat.OPT_BLOCK64,
                  nat.OPT_QUAD_POINT, nat.OPT_QUAD_POINT | nat.OPT_BLOCK64, 0):
        got = nat.sweep_host(db, n_steps=n, z_max=100.0, save_every=se, gamma=0.0115, alpha=1.15e-4, a0=A0,
                             check_nan=True, exact_step=True, extra_flags=flags)
        assert rel_err(got["a_end"], ref["a_end"]) < RTOL_F64
        assert rel_err(got["p_end"], ref["p_end"]) < RTOL_F64
        assert rel_err(got["p_max"], ref["p_max"]) < RTOL_F64
        assert np.array_equal(got["first_bad_step"], ref["first_bad_step"])


def test_per_point_gamma_alpha_a0_robustness_draw(oracle):
    """SURVEY 8(d) robustness run: dbeta, gamma, pump powers, phases all per point (defeats value shortcuts)."""
    rng = np.random.default_rng(2026)
    N = 777
    db = rng.uniform(-0.1, 0.1, N)
    gamma = rng.uniform(5e-3, 2e-2, N)
    alpha = rng.uniform(0.0, 3e-4, N)
    alpha[::5] = 0.0
    P = np.stack([rng.uniform(0.05, 1, N), rng.uniform(0.05, 1, N), 10 ** rng.uniform(-7, -4, N),
                  10 ** rng.uniform(-7, -4, N)], 1)
    a0 = np.sqrt(P) * np.exp(1j * rng.uniform(-np.pi, np.pi, (N, 4)))
    ref = oracle.sweep(db, z_max=400.0, n=4000, save_every=10, gamma=gamma, alpha=alpha, a0=a0)
    for lanes in (nat.OPT_ONE_LANE, nat.OPT_SPLIT_POINT, nat.OPT_QUAD_POINT):
        got = nat.sweep_host(db, n_steps=4000, z_max=400.0, save_every=10, gamma=gamma, alpha=alpha, a0=a0, extra_flags=lanes)
        assert rel_err(got["a_end"], ref["a_end"]) < RTOL_F64
        assert rel_err(got["p_max"], ref["p_max"]) < RTOL_F64
        assert (got["first_bad_step"] == -1).all()
    # broadcast flags one at a time
    for kw in (dict(gamma=0.0115), dict(alpha=1e-4), dict(a0=a0[3])):
        args = dict(gamma=gamma, alpha=alpha, a0=a0)
        args.update(kw)
        ref = oracle.sweep(db, z_max=400.0, n=400, save_every=10, **args)
        got = nat.sweep_host(db, n_steps=400, z_max=400.0, save_every=10, **args)
        assert rel_err(got["a_end"], ref["a_end"]) < RTOL_F64


def test_lds_staged_variant_matches_register_variant(oracle):
    """PSA_OPT_LDS_STAGING (state + k1..k4 through LDS, the north-star's sketch) is numerically the same algorithm."""
    db = np.linspace(-0.05, 0.05, 300)
    ref = oracle.sweep(db, z_max=300.0, n=3000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0)
    reg = nat.sweep_host(db, n_steps=3000, z_max=300.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0)
    lds = nat.sweep_host(db, n_steps=3000, z_max=300.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0,
                         extra_flags=nat.OPT_LDS_STAGING, exact_step=True)
    assert rel_err(lds["a_end"], ref["a_end"]) < RTOL_F64 and rel_err(lds["p_max"], ref["p_max"]) < RTOL_F64
    assert rel_err(lds["a_end"], reg["a_end"]) < 1e-12
    a06 = np.concatenate([A0, np.sqrt([2e-5, 1e-6])])
    r6 = oracle.sweep(db, z_max=100.0, n=1000, save_every=10, gamma=0.0115, alpha=0.0, a0=a06, dbeta2=-db)
    l6 = nat.sweep_host(db, n_steps=1000, z_max=100.0, save_every=10, gamma=0.0115, alpha=0.0, a0=a06, dbeta2=-db,
                        extra_flags=nat.OPT_LDS_STAGING)
    assert rel_err(l6["a_end"], r6["a_end"]) < RTOL_F64


def test_lossless_instantiation_equals_the_generic_kernel_at_alpha_zero(oracle):
    """alpha == 0 as a broadcast scalar selects the instantiation without the loss links (the reference's own
    `alpha == 0.0` branch); a per-point array of zeros runs the generic kernel.  Same numbers, to rounding."""
    db = np.linspace(-0.05, 0.05, 300)
    for a0v, d2 in ((A0, None), (np.concatenate([A0, np.sqrt([2e-5, 1e-6])]), -0.5 * db)):
        kw = dict(n_steps=3000, z_max=300.0, save_every=10, gamma=0.0115, a0=a0v, dbeta2=d2)
        ref = oracle.sweep(db, z_max=300.0, n=3000, save_every=10, gamma=0.0115, alpha=0.0, a0=a0v, dbeta2=d2)
        fast = nat.sweep_host(db, alpha=0.0, **kw)
        slow = nat.sweep_host(db, alpha=np.zeros(db.size), **kw)
        assert rel_err(fast["a_end"], ref["a_end"]) < RTOL_F64 and rel_err(slow["a_end"], ref["a_end"]) < RTOL_F64
        assert rel_err(fast["a_end"], slow["a_end"]) < 1e-12
        P = np.abs(fast["a_end"]) ** 2
        np.testing.assert_allclose(P.sum(1), (np.abs(a0v) ** 2).sum(), rtol=1e-11)     # lossless: total power conserved
    f32 = nat.sweep_host(db, alpha=0.0, dtype=np.float32, extra_flags=nat.OPT_F32_SCALAR, n_steps=3000, z_max=300.0,
                         save_every=10, gamma=0.0115, a0=A0)
    ref = oracle.sweep(db, z_max=300.0, n=3000, save_every=10, gamma=0.0115, alpha=0.0, a0=A0)
    assert rel_err(f32["a_end"].astype(complex), ref["a_end"]) < RTOL_F32


def test_empty_sweep_is_a_noop():
    got = nat.sweep_host(np.zeros(0), n_steps=10, z_max=1.0, save_every=1, gamma=1.0, alpha=0.0, a0=A0)
    assert got["a_end"].shape == (0, 4) and got["p_max"].shape == (0,)


# ---- golden fixtures (reference outputs) -------------------------------------------------------------------------
@pytest.mark.parametrize("key,n,ai", [("n1e4_a0", 10_000, 0), ("n1e4_a1", 10_000, 1)])
def test_g8_direct_dbeta_sweep(golden, key, n, ai):
    g = golden("G8")
    got = nat.sweep_host(g["dbeta257"], n_steps=n, z_max=1000.0, save_every=10, gamma=float(g["gamma"]),
                         alpha=float(g["alphas"][ai]), a0=_a0(g["p_in"]))
    assert rel_err(got["a_end"], g[key + "_A_end"]) < RTOL_F64
    assert rel_err(got["p_end"], g[key + "_p_end"]) < RTOL_F64
    assert rel_err(got["p_max"], g[key + "_p_max"]) < RTOL_F64


def test_g8_1e5_steps(golden):
    g = golden("G8")
    got = nat.sweep_host(g["dbeta33"], n_steps=100_000, z_max=1000.0, save_every=10, gamma=float(g["gamma"]),
                         alpha=float(g["alphas"][1]), a0=_a0(g["p_in"]))
    assert rel_err(got["a_end"], g["n1e5_a1_A_end"]) < RTOL_F64
    assert rel_err(got["p_max"], g["n1e5_a1_p_max"]) < RTOL_F64


def test_g2_sweep_a_end_and_gain(golden):
    g = golden("G2")
    got = nat.sweep_host(g["dbeta"], n_steps=2500, z_max=500.0, save_every=10, gamma=float(g["gamma"]),
                         alpha=float(g["alpha"]), a0=_a0(g["p_in"]))
    assert rel_err(got["a_end"], g["A_end"]) < RTOL_F64
    gain, bi, bg, nf = nat.gain_summary_host(got["p_max"], got["first_bad_step"], g["p_in"][2], gain_db=True)
    np.testing.assert_allclose(gain, g["gain_db"], rtol=RTOL_F64, atol=ATOL_DB)
    assert bi == int(np.argmax(g["gain_db"])) == 14 and nf == 30
    assert bg == pytest.approx(7.6893941298573782, abs=ATOL_DB)


def test_g5_rhs_kernel_and_terms(golden):
    g = golden("G5")
    out, lin, kerr, fwm = nat.yaman_rhs_host(g["z"], g["a"], g["gamma"], g["alpha"], g["dbeta"], terms=True)
    for got, key in ((out, "rhs"), (lin, "linear"), (kerr, "kerr"), (fwm, "fwm")):
        scale = np.max(np.abs(g[key]), axis=1, keepdims=True)
        assert np.max(np.abs(got - g[key]) / np.maximum(scale, 1e-300)) < 1e-14, key


def test_g7_trajectory_rows_and_stride_edges(golden):
    g = golden("G7")
    a0 = _a0(g["p_in"], g["phase_in"])
    for tag in ("n1005_se10", "n1005_se1", "n3_se1", "n3_se2", "n7_se10"):
        z_max, dz, se = g[tag + "_cfg"]
        n = int(round(z_max / dz))
        got = nat.sweep_host([float(g["dbeta"])], n_steps=n, z_max=z_max, save_every=int(se), gamma=float(g["gamma"]),
                             alpha=float(g["alpha"]), a0=a0, want_traj=True, exact_step=True)
        ref = g[tag + "_A"]
        assert got["traj"].shape == (1,) + ref.shape, tag
        assert rel_err(got["traj"][0], ref) < RTOL_F64, tag
        assert rel_err(got["a_end"][0], ref[-1]) < RTOL_F64, tag     # A[-1] = last SAVED row


def test_trajectory_layout_for_many_points(oracle):
    """N > 1 exercises the SoA [row][comp][N] -> AoS [N][row][wave] transpose kernel (ragged N, rows*comps % 64 != 0)."""
    N, n, se = 130, 205, 4
    db = np.linspace(-0.05, 0.05, N)
    got = nat.sweep_host(db, n_steps=n, z_max=50.0, save_every=se, gamma=0.0115, alpha=1.15e-4, a0=A0, want_traj=True)
    assert got["traj"].shape == (N, n // se + 1, 4)
    for i in (0, 1, 63, 64, 129):
        z, A, _ = oracle.integrate(A0, z_max=50.0, n=n, save_every=se, gamma=0.0115, alpha=1.15e-4, dbeta=db[i])
        assert rel_err(got["traj"][i], A) < RTOL_F64
    assert np.array_equal(got["traj"][:, -1, :], got["a_end"])
    assert np.array_equal(np.abs(got["traj"][:, :, 2]).max(axis=1) ** 2 > 0, np.ones(N, bool))


# ---- failure path -------------------------------------------------------------------------------------------------
def test_g9_first_bad_step_exact_and_block_modes(golden):
    g = golden("G9")
    a0 = _a0(g["p_in"])
    gam = g["gammas"]
    db = np.full(gam.size, float(g["dbeta"]))
    exact = nat.sweep_host(db, n_steps=1000, z_max=100.0, save_every=10, gamma=gam, alpha=0.0, a0=a0, check_nan=True,
                           exact_step=True)
    assert np.array_equal(exact["first_bad_step"], g["first_bad_step"])
    block = nat.sweep_host(db, n_steps=1000, z_max=100.0, save_every=10, gamma=gam, alpha=0.0, a0=a0, check_nan=True,
                           exact_step=False)
    want = np.where(g["first_bad_step"] >= 0, 9, -1)          # last step of the first bad save block
    assert np.array_equal(block["first_bad_step"], want)
    off = nat.sweep_host(db, n_steps=1000, z_max=100.0, save_every=10, gamma=gam, alpha=0.0, a0=a0, check_nan=False)
    assert (off["first_bad_step"] == -1).all()
    failed = g["first_bad_step"] >= 0
    assert np.isnan(off["p_max"][failed]).all() and np.isfinite(off["p_max"][~failed]).all()
    gain, bi, bg, nf = nat.gain_summary_host(block["p_max"], block["first_bad_step"], g["p_in"][2])
    assert np.array_equal(np.isnan(gain), failed) and nf == int((~failed).sum()) and bi == int(np.flatnonzero(~failed)[0])


def test_failure_in_the_unsaved_tail_is_seen_only_with_check_nan(oracle):
    """n = 4 < save_every = 10: no row is ever saved after z = 0, so the blow-up at step 2 (gamma = 12, h = 0.1, as in
    G9) lies in the unsaved tail; check_nan must still report it, and A[-1] stays the input."""
    ref = oracle.sweep(np.array([0.01]), z_max=0.4, n=4, save_every=10, gamma=12.0, alpha=0.0, a0=A0)
    assert ref["first_bad_step"][0] == 2
    got = nat.sweep_host([0.01], n_steps=4, z_max=0.4, save_every=10, gamma=12.0, alpha=0.0, a0=A0, check_nan=True,
                         exact_step=True)
    assert got["first_bad_step"][0] == 2
    assert np.array_equal(got["a_end"][0], A0) and got["p_max"][0] == got["p_end"][0] == abs(A0[2]) ** 2
    dflt = nat.sweep_host([0.01], n_steps=4, z_max=0.4, save_every=10, gamma=12.0, alpha=0.0, a0=A0, check_nan=True)
    assert dflt["first_bad_step"][0] == 2                     # float64 default: the exact index (free, by replay)
    blk = nat.sweep_host([0.01], n_steps=4, z_max=0.4, save_every=10, gamma=12.0, alpha=0.0, a0=A0, check_nan=True,
                         exact_step=False)
    assert blk["first_bad_step"][0] == 3                      # block mode: last step of the (partial) block
    off = nat.sweep_host([0.01], n_steps=4, z_max=0.4, save_every=10, gamma=12.0, alpha=0.0, a0=A0, check_nan=False)
    assert off["first_bad_step"][0] == -1 and np.isfinite(off["p_max"][0])


# ---- gain summary kernel ----------------------------------------------------------------------------------------------
def test_gain_summary_reduction_matches_numpy():
    rng = np.random.default_rng(5)
    for N in (1, 64, 300, 70_001):
        p = 10 ** rng.uniform(-9, -2, N)
        bad = np.full(N, -1, np.int64)
        p[rng.integers(0, N, max(1, N // 50))] = np.nan
        p[rng.integers(0, N, max(1, N // 50))] = np.inf
        p[rng.integers(0, N, max(1, N // 50))] = 0.0
        bad[rng.integers(0, N, max(1, N // 50))] = 7
        for db in (True, False):
            gain, bi, bg, nf = nat.gain_summary_host(p, bad, 1e-7, gain_db=db)
            with np.errstate(all="ignore"):
                g = p / 1e-7
                ok = np.isfinite(p) & np.isfinite(g) & (g > 0) & (bad < 0)
                ref = np.where(ok, 10 * np.log10(np.where(ok, g, 1.0)) if db else g, np.nan)
            np.testing.assert_allclose(gain, ref, rtol=1e-14, equal_nan=True)
            assert nf == int(ok.sum())
            if ok.any():
                assert bi == int(np.nanargmax(ref)) and bg == pytest.approx(np.nanmax(ref), rel=1e-14)
            else:
                assert bi == -1 and np.isnan(bg)
    gain, bi, bg, nf = nat.gain_summary_host(np.array([np.nan, np.nan]), None, 1.0)
    assert bi == -1 and nf == 0 and np.isnan(bg)


# ---- float32 and 6-wave variants (build-defined extensions) -----------------------------------------------------
def test_float32_variant_within_build_defined_tolerance(oracle):
    db = np.linspace(-0.05, 0.05, 513)
    ref = oracle.sweep(db, z_max=1000.0, n=10_000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0)
    got = nat.sweep_host(db, n_steps=10_000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0,
                         dtype=np.float32, want_traj=False)
    assert got["a_end"].dtype == np.complex64
    assert rel_err(got["a_end"].astype(complex), ref["a_end"]) < RTOL_F32
    assert rel_err(got["p_max"].astype(float), ref["p_max"]) < RTOL_F32
    gd = 10 * np.log10(got["p_max"].astype(float) / 1e-5) - 10 * np.log10(ref["p_max"] / 1e-5)
    assert np.max(np.abs(gd)) < 5e-4                             # dB


def test_float32_at_config4_step_count(oracle):
    """BASELINE config 4 runs 1e6 z-steps in float32.  Without the compensated state update the error grows ~n
    (5e-3 at 1e6 steps); with it the run must stay inside RTOL_F32 against float64 (GPU, 384 points) and against the
    oracle (4 points at full length)."""
    N, n = 384, 1_000_000
    db = np.linspace(-0.02, 0.02, N)
    a0 = _a0([0.1, 0.1, 1e-7, 1e-7])
    kw = dict(n_steps=n, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    r64 = nat.sweep_host(db, **kw)
    r32 = nat.sweep_host(db, dtype=np.float32, **kw)
    assert rel_err(r32["a_end"].astype(complex), r64["a_end"]) < RTOL_F32
    assert rel_err(r32["p_max"].astype(float), r64["p_max"]) < RTOL_F32
    pick = np.array([0, 100, 200, 383])
    ref = oracle.sweep(db[pick], z_max=1000.0, n=n, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    assert rel_err(r64["a_end"][pick], ref["a_end"]) < RTOL_F64
    assert rel_err(r32["a_end"][pick].astype(complex), ref["a_end"]) < RTOL_F32


@pytest.mark.parametrize("N", [1, 2, 7, 128, 1001])
def test_float32_packed_two_points_per_lane_matches_scalar_float32(oracle, N):
    """PSA_OPT_F32_PACKED (v_pk_fma_f32, points 2i and 2i+1 share a lane) runs the same float32 algorithm as the
    one-point-per-lane kernel: they must agree to float32 rounding, odd N and per-point arguments included."""
    rng = np.random.default_rng(N)
    db = rng.uniform(-0.05, 0.05, N)
    gam = rng.uniform(0.008, 0.014, N)
    al = rng.uniform(0.0, 2e-4, N)
    a0 = np.sqrt(rng.uniform([0.3, 0.3, 1e-6, 1e-6], [0.6, 0.6, 1e-4, 1e-4], (N, 4))) * np.exp(1j * rng.uniform(-3, 3, (N, 4)))
    kw = dict(n_steps=2000, z_max=200.0, save_every=10, gamma=gam, alpha=al, a0=a0, dtype=np.float32, want_traj=(N <= 7),
              exact_step=True)
    sc = nat.sweep_host(db, extra_flags=nat.OPT_F32_SCALAR, **kw)
    pk = nat.sweep_host(db, extra_flags=nat.OPT_F32_PACKED, **kw)
    ref = oracle.sweep(db, z_max=200.0, n=2000, save_every=10, gamma=gam, alpha=al, a0=a0)
    assert rel_err(pk["a_end"].astype(complex), sc["a_end"].astype(complex)) < 2e-5
    assert rel_err(pk["p_max"].astype(float), sc["p_max"].astype(float)) < 2e-5
    assert rel_err(pk["a_end"].astype(complex), ref["a_end"]) < RTOL_F32
    assert np.array_equal(pk["first_bad_step"], sc["first_bad_step"]) and (pk["first_bad_step"] == -1).all()
    if N <= 7:
        assert pk["traj"].shape == sc["traj"].shape == (N, 201, 4)
        assert rel_err(pk["traj"].astype(complex), sc["traj"].astype(complex)) < 2e-5
    # failure tracking per packed half: make exactly one point of a pair blow up
    if N >= 2:
        g2 = gam.copy()
        g2[1] = 40.0
        b = nat.sweep_host(db, n_steps=200, z_max=20.0, save_every=10, gamma=g2, alpha=al, a0=a0, dtype=np.float32,
                           extra_flags=nat.OPT_F32_PACKED, exact_step=True)
        assert b["first_bad_step"][1] >= 0 and b["first_bad_step"][0] == -1 and np.isfinite(b["p_max"][0])


def test_float32_six_wave_packed_matches_scalar(oracle):
    db = np.linspace(-0.04, 0.04, 65)
    a06 = np.sqrt(np.array([0.5, 0.4, 1e-5, 1e-5, 3e-5, 2e-6])).astype(complex)
    kw = dict(n_steps=1000, z_max=100.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a06, dbeta2=0.5 * db, dtype=np.float32)
    sc = nat.sweep_host(db, extra_flags=nat.OPT_F32_SCALAR, **kw)
    pk = nat.sweep_host(db, extra_flags=nat.OPT_F32_PACKED, **kw)
    assert rel_err(pk["a_end"].astype(complex), sc["a_end"].astype(complex)) < 2e-5


def test_six_wave_kernel_vs_oracle_and_reduction(oracle):
    """6-wave RHS is build-defined ("parity unpinned" vs the reference, which has none): the kernel must match the
    oracle's statement of the same equations, and reduce to the 4-wave system when pair 2 is dark."""
    db = np.linspace(-0.04, 0.04, 129)
    db2 = 0.6 * db + 0.003
    a06 = np.sqrt(np.array([0.5, 0.4, 1e-5, 1e-5, 3e-5, 2e-6])) * np.exp(1j * np.array([0.0, 0.3, 0.1, -0.2, 0.5, 1.0]))
    ref = oracle.sweep(db, z_max=600.0, n=6000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a06, dbeta2=db2)
    got = nat.sweep_host(db, n_steps=6000, z_max=600.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a06, dbeta2=db2)
    assert got["a_end"].shape == (129, 6)
    assert rel_err(got["a_end"], ref["a_end"]) < RTOL_F64 and rel_err(got["p_max"], ref["p_max"]) < RTOL_F64
    dark = np.concatenate([A0, [0, 0]])
    got6 = nat.sweep_host(db, n_steps=6000, z_max=600.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=dark, dbeta2=db2)
    got4 = nat.sweep_host(db, n_steps=6000, z_max=600.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0)
    assert rel_err(got6["a_end"][:, :4], got4["a_end"]) < 1e-12 and np.all(got6["a_end"][:, 4:] == 0)


def test_six_wave_trajectory_rows(oracle):
    a06 = np.sqrt(np.array([0.5, 0.4, 1e-5, 1e-5, 3e-5, 2e-6])) * np.exp(1j * np.array([0.0, 0.3, 0.1, -0.2, 0.5, 1.0]))
    db, db2 = np.array([0.011, -0.02, 0.0]), np.array([-0.004, 0.015, 0.03])
    got = nat.sweep_host(db, n_steps=333, z_max=33.3, save_every=4, gamma=0.0115, alpha=1.15e-4, a0=a06, dbeta2=db2,
                         want_traj=True, exact_step=True)
    assert got["traj"].shape == (3, 333 // 4 + 1, 6)
    for i in range(3):
        z, A, bad = oracle.integrate(a06, z_max=33.3, n=333, save_every=4, gamma=0.0115, alpha=1.15e-4, dbeta=db[i],
                                     dbeta2=db2[i])
        assert rel_err(got["traj"][i], A) < RTOL_F64 and bad == -1
    assert np.array_equal(got["traj"][:, -1], got["a_end"])


# ---- BASELINE.json full sizes: size-independent properties + sampled oracle check -------------------------------------
def _c2_inputs(N=65536):
    return np.linspace(-0.05, 0.05, N)


def test_full_size_c2_properties_and_sampled_parity(oracle):
    """Config 2: 65 536 points x 100 000 steps, fp64.  (i) 64 randomly sampled points against the oracle at full
    length; (ii) power balance sum_j |A_j|^2 (L) = sum_j |A_j|^2 (0) * exp(-alpha L) (exact for the ODE, RK4 keeps
    it to ~1e-12 here); (iii) Manley-Rowe: |A3|^2 - |A4|^2 follows the same decay; (iv) two shards == one launch,
    bit for bit; (v) rerun is bit-identical."""
    N, n, L, alpha = 65536, 100_000, 1000.0, 1.15e-4
    db = _c2_inputs(N)
    got = nat.sweep_host(db, n_steps=n, z_max=L, save_every=10, gamma=0.0115, alpha=alpha, a0=A0, check_nan=True)
    assert (got["first_bad_step"] == -1).all()
    pick = np.random.default_rng(65536).choice(N, 64, replace=False)
    ref = oracle.sweep(db[pick], z_max=L, n=n, save_every=10, gamma=0.0115, alpha=alpha, a0=A0)
    assert rel_err(got["a_end"][pick], ref["a_end"]) < RTOL_F64
    assert rel_err(got["p_max"][pick], ref["p_max"]) < RTOL_F64
    P = np.abs(got["a_end"]) ** 2
    np.testing.assert_allclose(P.sum(1), P_IN.sum() * np.exp(-alpha * L), rtol=1e-10)
    np.testing.assert_allclose(P[:, 2] - P[:, 3], (P_IN[2] - P_IN[3]) * np.exp(-alpha * L), rtol=0, atol=1e-12)
    np.testing.assert_allclose(P[:, 0] - P[:, 1], 0.0, atol=1e-12)
    assert np.all(got["p_max"] >= got["p_end"]) and np.all(got["p_max"] >= P_IN[2] * (1 - 1e-15))
    h = N // 2
    lo = nat.sweep_host(db[:h], n_steps=n, z_max=L, save_every=10, gamma=0.0115, alpha=alpha, a0=A0)
    hi = nat.sweep_host(db[h:], n_steps=n, z_max=L, save_every=10, gamma=0.0115, alpha=alpha, a0=A0)
    assert np.array_equal(np.concatenate([lo["a_end"], hi["a_end"]]), got["a_end"])
    assert np.array_equal(np.concatenate([lo["p_max"], hi["p_max"]]), got["p_max"])


def test_full_size_c3_grid_sampled_parity(oracle):
    """Config 3 shape: 1 048 576 points x 100 000 steps (about one second of GPU time); sampled oracle check."""
    N, n, L, alpha = 1 << 20, 100_000, 1000.0, 1.15e-4
    rng = np.random.default_rng(3)
    db = rng.uniform(-0.02, 0.02, N)
    a0 = _a0([0.1, 0.1, 1e-7, 1e-7])
    got = nat.sweep_host(db, n_steps=n, z_max=L, save_every=10, gamma=0.0115, alpha=alpha, a0=a0, check_nan=True)
    pick = rng.choice(N, 48, replace=False)
    ref = oracle.sweep(db[pick], z_max=L, n=n, save_every=10, gamma=0.0115, alpha=alpha, a0=a0)
    assert rel_err(got["a_end"][pick], ref["a_end"]) < RTOL_F64
    P = np.abs(got["a_end"]) ** 2
    np.testing.assert_allclose(P.sum(1), 0.2000002 * np.exp(-alpha * L), rtol=1e-10)


def _six_wave_invariants(a_end, P6, alpha, L, rtol):
    """What the six-wave equations conserve up to the common loss factor exp(-alpha L): the total power and
    the Manley-Rowe differences |s_k|^2 - |i_k|^2 (k = 1, 2) and |p1|^2 - |p2|^2 -- every photon pair taken from the two
    pumps goes into one signal/idler pair."""
    P = np.abs(a_end) ** 2
    decay = np.exp(-alpha * L)
    np.testing.assert_allclose(P.sum(1), P6.sum() * decay, rtol=rtol)
    scale = P6.sum() * decay                      # differences of O(1e-6) are held to the same ABSOLUTE accuracy
    for a, b in ((2, 3), (4, 5), (0, 1)):
        np.testing.assert_allclose(P[:, a] - P[:, b], (P6[a] - P6[b]) * decay, rtol=0, atol=rtol * scale)


def test_config5_shard_shape_six_wave_sampled_parity(oracle):
    """BASELINE config 5 per-GPU shard: 32 768 points (a 128 x 256 slice of the (Omega1, Omega2) grid) x 6 waves x
    100 000 z-steps, float64 -- both lane layouts (the library's choice at this size is two lanes per point).  The 6-wave
    RHS is build-defined (parity unpinned vs the reference); the kernel must match the oracle's statement of the same
    equations on sampled points, and every point must keep the model's invariants at full length."""
    n, L, alpha = 100_000, 1000.0, 1.15e-4
    d1, d2 = np.meshgrid(np.linspace(-0.03, 0.03, 128), np.linspace(-0.02, 0.04, 256), indexing="ij")
    db, db2 = d1.ravel(), d2.ravel()
    P6 = np.array([0.3, 0.25, 1e-6, 1e-6, 2e-6, 5e-7])
    a06 = np.sqrt(P6).astype(complex)
    pick = np.random.default_rng(5).choice(db.size, 12, replace=False)
    ref = oracle.sweep(db[pick], z_max=L, n=n, save_every=10, gamma=0.0115, alpha=alpha, a0=a06, dbeta2=db2[pick])
    ms = {}
    for name, lanes in (("two lanes/point", nat.OPT_SPLIT_POINT), ("one lane/point", nat.OPT_ONE_LANE)):
        got = nat.sweep_host(db, n_steps=n, z_max=L, save_every=10, gamma=0.0115, alpha=alpha, a0=a06, dbeta2=db2,
                             extra_flags=lanes)
        assert (got["first_bad_step"] == -1).all()
        assert rel_err(got["a_end"][pick], ref["a_end"]) < RTOL_F64
        assert rel_err(got["p_max"][pick], ref["p_max"]) < RTOL_F64
        _six_wave_invariants(got["a_end"], P6, alpha, L, 1e-10)
        ms[name] = got["elapsed_ms"]
    print("config 5 shard kernel ms:", {k: round(v, 2) for k, v in ms.items()},
          f"-> x{ms['two lanes/point'] / ms['one lane/point']:.3f}")


def test_config5_full_grid_through_the_six_wave_driver(oracle):
    """BASELINE config 5 whole: scan_six_wave_grid on the real 512 x 512 (Omega1, Omega2) grid, 262 144 points x 6 waves x
    100 000 z-steps in one launch (~0.4 s).  Invariants on every point, sampled points against the oracle, and the grid's
    structure: dbeta_1 depends on the row only, dbeta_2 on the column only."""
    from psa_amd import config, dispersion, frequency_plan, scan_mismtach
    om = frequency_plan.plan_from_wavelengths(1550e-9, 1558e-9, 1540e-9)
    sp = frequency_plan.infer_symmetry_from_omegas(*om)
    d = dispersion.dispersion_params_from_D_S(frequency_plan.lambda_from_omega(sp.omega_c), 0.1, 0.02, 0,
                                              D_units="ps/nm/km", S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km",
                                              omega_ref=sp.omega_c)
    P6 = np.array([0.3, 0.25, 1e-6, 1e-6, 2e-6, 5e-7])
    O1, O2 = np.linspace(2e12, 2.4e13, 512), np.linspace(3e12, 2.0e13, 512)
    cfg = config.custom_simulation_config(z_max=1000.0, dz=0.01)
    out = scan_mismtach.scan_six_wave_grid(cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, Omega1=O1, Omega2=O2,
                                           gamma=0.0115, alpha=1.15e-4, p_in=P6, dispersion=d, gain_unit="linear")
    assert out["gain"].shape == (512, 512) and (out["first_bad_step"] == -1).all()
    assert out["result"].n_steps == 100_000
    _six_wave_invariants(out["a_end"].reshape(-1, 6), P6, 1.15e-4, 1000.0, 1e-10)
    rng = np.random.default_rng(55)
    for iy, ix in zip(rng.integers(0, 512, 10), rng.integers(0, 512, 10)):
        ref = oracle.sweep(np.array([out["dbeta1"][iy]]), dbeta2=np.array([out["dbeta2"][ix]]), z_max=1000.0, n=100_000,
                           save_every=10, gamma=0.0115, alpha=1.15e-4, a0=np.sqrt(P6).astype(complex))
        assert rel_err(out["a_end"][iy, ix], ref["a_end"][0]) < RTOL_F64
        assert out["gain"][iy, ix] == pytest.approx(ref["p_max"][0] / P6[2], rel=RTOL_F64)
    assert out["gain"].max() > 10.0                  # the grid crosses pair 1's phase-matched band


def test_config4_per_gpu_shard_at_full_length_float32(oracle):
    """BASELINE config 4, one GPU's shard exactly: 131 072 sweep points x 4 fields x 1 000 000 z-steps, float32
    (packed kernel + compensated state), ~0.7 s of kernel time.  Eight sampled points against the float64 oracle at full
    length; every point finite; total power follows exp(-alpha L) to float32 accuracy."""
    N, n, L, alpha = 131_072, 1_000_000, 1000.0, 1.15e-4
    rng = np.random.default_rng(44)
    db = rng.uniform(-0.02, 0.02, N).astype(np.float32)
    p_in = np.array([0.1, 0.1, 1e-7, 1e-7])
    got = nat.sweep_host(db, n_steps=n, z_max=L, save_every=10, gamma=0.0115, alpha=alpha, a0=_a0(p_in), dtype=np.float32)
    assert (got["first_bad_step"] == -1).all() and got["a_end"].dtype == np.complex64
    pick = rng.choice(N, 8, replace=False)
    ref = oracle.sweep(db[pick].astype(np.float64), z_max=L, n=n, save_every=10, gamma=0.0115, alpha=alpha, a0=_a0(p_in))
    assert rel_err(got["a_end"][pick].astype(complex), ref["a_end"]) < RTOL_F32
    assert rel_err(got["p_max"][pick].astype(float), ref["p_max"]) < RTOL_F32
    P = (np.abs(got["a_end"].astype(complex)) ** 2).sum(1)
    np.testing.assert_allclose(P, p_in.sum() * np.exp(-alpha * L), rtol=2e-5)
    print(f"config 4 shard: kernel {got['elapsed_ms']:.1f} ms -> {4 * N * n / got['elapsed_ms'] / 1e6:.0f} G updates/s")


def test_randomized_differential_against_oracle(oracle):
    """60 seeded random configurations: point count, step count, save stride (below / at / above the 32-step chunk and
    the 64-step re-seed period), check mode, trajectory on/off, 4 or 6 waves, broadcast or per-point gamma/alpha/A0,
    one or two lanes per sweep point."""
    rng = np.random.default_rng(20261004)
    strides = [1, 2, 3, 7, 10, 31, 32, 33, 63, 64, 65, 100, 257, 1000]
    for case in range(60):
        N = int(rng.integers(1, 200))
        n = int(rng.integers(1, 700))
        se = int(rng.choice(strides))
        nw = int(rng.choice([4, 6]))
        check, exact = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        traj = bool(rng.integers(0, 2)) and (n // se + 1) * N < 50_000
        L = float(rng.uniform(5.0, 60.0))
        db = rng.uniform(-0.08, 0.08, N)
        db2 = rng.uniform(-0.08, 0.08, N) if nw == 6 else None
        gamma = rng.uniform(5e-3, 2e-2, N) if rng.integers(0, 2) else float(rng.uniform(5e-3, 2e-2))
        alpha = rng.uniform(0, 3e-4, N) if rng.integers(0, 2) else float(rng.choice([0.0, 1.15e-4]))
        amp = np.sqrt(rng.uniform(1e-6, 0.8, (N, nw))) * np.exp(1j * rng.uniform(-3.1, 3.1, (N, nw)))
        a0 = amp if rng.integers(0, 2) else amp[0]
        lanes = int(rng.choice([nat.OPT_ONE_LANE, nat.OPT_SPLIT_POINT]))
        tag = f"case {case}: N={N} n={n} se={se} nw={nw} check={check} exact={exact} traj={traj} lanes={lanes:#x}"
        ref = oracle.sweep(db, z_max=L, n=n, save_every=se, check_nan=check, gamma=gamma, alpha=alpha, a0=a0, dbeta2=db2)
        got = nat.sweep_host(db, n_steps=n, z_max=L, save_every=se, gamma=gamma, alpha=alpha, a0=a0, dbeta2=db2,
                             check_nan=check, exact_step=exact, want_traj=traj, extra_flags=lanes)
        assert rel_err(got["a_end"], ref["a_end"]) < RTOL_F64, tag
        assert rel_err(got["p_end"], ref["p_end"]) < RTOL_F64 and rel_err(got["p_max"], ref["p_max"]) < RTOL_F64, tag
        assert np.array_equal(got["first_bad_step"], ref["first_bad_step"]) and (got["first_bad_step"] == -1).all(), tag
        if traj:
            i = int(rng.integers(0, N))
            g_i = gamma[i] if np.ndim(gamma) else gamma
            al_i = alpha[i] if np.ndim(alpha) else alpha
            a_i = a0[i] if a0.ndim == 2 else a0
            z, A, _ = oracle.integrate(a_i, z_max=L, n=n, save_every=se, check_nan=check, gamma=g_i, alpha=al_i,
                                       dbeta=db[i], dbeta2=(db2[i] if nw == 6 else 0.0))
            assert got["traj"].shape == (N, n // se + 1, nw), tag
            assert rel_err(got["traj"][i], A) < RTOL_F64, tag


def test_extreme_parameters(oracle):
    """Corners of the input space: |dbeta*h| ~ 0.25 rad per step and dbeta*z up to ~1.2e4 rad (exercises the phase
    recurrence and the large-argument sincos re-seed), seed powers down to 1e-24 W and exactly zero, dark pumps,
    negative gamma, strong loss."""
    rng = np.random.default_rng(99)
    N = 96
    db = np.concatenate([rng.uniform(-12.0, 12.0, N - 4), [12.0, -12.0, 0.0, 1e-12]])
    P = np.stack([10 ** rng.uniform(-3, 0, N), 10 ** rng.uniform(-3, 0, N), 10 ** rng.uniform(-24, -3, N),
                  10 ** rng.uniform(-24, -3, N)], 1)
    P[0] = [0.5, 0.5, 0.0, 0.0]        # no seed at all: sidebands must stay exactly zero
    P[1] = [0.0, 0.0, 1e-3, 1e-3]      # dark pumps
    P[2] = [0.0, 0.7, 1e-6, 0.0]
    a0 = np.sqrt(P) * np.exp(1j * rng.uniform(-3.1, 3.1, (N, 4)))
    gamma = rng.uniform(-0.02, 0.02, N)
    alpha = 10 ** rng.uniform(-6, -2, N)
    n, L = 50_000, 1000.0              # h = 0.02 m
    ref = oracle.sweep(db, z_max=L, n=n, save_every=100, gamma=gamma, alpha=alpha, a0=a0)
    for lanes in (nat.OPT_ONE_LANE, nat.OPT_SPLIT_POINT):
        got = nat.sweep_host(db, n_steps=n, z_max=L, save_every=100, gamma=gamma, alpha=alpha, a0=a0, exact_step=True,
                             extra_flags=lanes)
        assert (got["first_bad_step"] == -1).all() and (ref["first_bad_step"] == -1).all()
        scale = np.abs(ref["a_end"]).max(axis=1, keepdims=True)
        # per-point normalisation: a wave that is ~1e-12 of the pumps carries absolute, not relative, rounding noise
        assert np.max(np.abs(got["a_end"] - ref["a_end"]) / scale) < RTOL_F64
        strong = np.abs(ref["a_end"]) > 1e-6 * scale
        assert rel_err(got["a_end"][strong], ref["a_end"][strong]) < 1e-7
        assert np.all(got["a_end"][0, 2:] == 0) and np.all(got["a_end"][1, :2] == 0)
        live = ref["p_max"] > 0
        assert rel_err(got["p_max"][live], ref["p_max"][live]) < 1e-7


def test_trajectory_leaves_the_device_in_chunks(oracle):
    """Host-buffer API: a 453 MB trajectory (70 001 points x 101 rows) is transposed and copied in two chunks through the
    bounded staging buffers (psa_capi.hip, TRAJ_STAGE_BYTES = 256 MB), the second one ragged (not a multiple of the 32-point
    transpose tile).  Rows of points on both sides of the chunk boundary against the oracle; A[-1] == last row everywhere."""
    N, n = 70_001, 100
    db = np.linspace(-0.06, 0.06, N)
    got = nat.sweep_host(db, n_steps=n, z_max=10.0, save_every=1, gamma=0.0115, alpha=1.15e-4, a0=A0, want_traj=True)
    assert got["traj"].shape == (N, n + 1, 4)
    assert np.array_equal(got["traj"][:, -1, :], got["a_end"]) and np.all(got["traj"][:, 0, :] == A0)
    boundary = (256 * 2**20 // ((n + 1) * 64)) // 32 * 32            # first point of the second chunk
    for i in (0, 31, 32, boundary - 1, boundary, boundary + 1, N - 2, N - 1):
        z, A, _ = oracle.integrate(A0, z_max=10.0, n=n, save_every=1, gamma=0.0115, alpha=1.15e-4, dbeta=db[i])
        assert rel_err(got["traj"][i], A) < RTOL_F64, i


def test_a_trajectory_that_cannot_fit_is_refused_before_any_allocation():
    """PSA_E_TOO_LARGE (-9), not hipErrorOutOfMemory: 2^26 points x 100 001 rows = 4.3e14 B.  The call returns from its
    size check, so the small dummy buffers are never read or written."""
    import ctypes as C
    buf = np.zeros(64)
    p = buf.ctypes.data_as(C.c_void_p)
    L = nat.lib()
    flags = nat.BCAST_GAMMA | nat.BCAST_ALPHA | nat.BCAST_A0
    rc = L.psa_rk4_sweep_f64(0, 4, 2**26, 100_000, 1.0, 1, p, None, p, p, p, flags, p, p, p, p, p, None)
    assert rc == -9 and b"does not fit" in L.psa_last_error()
    rc = L.psa_rk4_sweep_f64_dev(None, 4, 2**28, 10, 1.0, 1, p, None, p, p, p, flags, p, p, p, p, p)
    assert rc == -9                                                   # trajectory launches address lanes with 32 bits


def test_g15_reference_robustness_draw_both_layouts(golden):
    """G15 (generated from the reference): 32 single runs with every physical and numerical parameter random.  Each run is
    its own launch (own step count and save stride), in both float64 lane layouts, against the reference's final row, max
    signal power and -- for four runs, one with save_every = 1 -- the whole trajectory."""
    g = golden("G15")
    for i in range(32):
        a0 = _a0(g["p_in"][i], g["phase_in"][i])
        full = f"A_full_{i}" in g.files
        for lanes in (nat.OPT_ONE_LANE, nat.OPT_SPLIT_POINT):
            got = nat.sweep_host([float(g["dbeta"][i])], n_steps=int(g["n"][i]), z_max=float(g["L"][i]),
                                 save_every=int(g["save_every"][i]), gamma=float(g["gamma"][i]), alpha=float(g["alpha"][i]),
                                 a0=a0, exact_step=True, want_traj=full, extra_flags=lanes)
            assert got["first_bad_step"][0] == -1
            assert rel_err(got["a_end"][0], g["A_end"][i]) < RTOL_F64, (i, lanes)
            assert rel_err(got["p_max"][0], g["p_max"][i]) < RTOL_F64, (i, lanes)
            if full:
                assert got["traj"].shape[1] == int(g["n_rows"][i]) and rel_err(got["traj"][0], g[f"A_full_{i}"]) < RTOL_F64
    # all 32 through the reference-shaped single-run API as well (its own n = int(round(L / dz)) rule)
    from psa_amd import config, simulation
    from psa_amd.phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
    for i in (0, 7, 19, 31):
        cfg = config.custom_simulation_config(z_max=float(g["L"][i]), dz=float(g["L"][i]) / int(g["n"][i]), save_every=int(g["save_every"][i]))
        pm = PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=float(g["dbeta"][i]))
        z, A = simulation.run_single_simulation(cfg, gamma=float(g["gamma"][i]), alpha=float(g["alpha"][i]), omega=np.full(4, 1.2e15),
                                                p_in=g["p_in"][i], phase_in=g["phase_in"][i], phase_matching_cfg=pm)
        assert A.shape == g[f"A_full_{i}"].shape and rel_err(A, g[f"A_full_{i}"]) < RTOL_F64
        assert abs(z[-1] - g["z_last"][i]) <= 1e-12 * g["L"][i]


def test_sixteen_million_points_in_one_launch(oracle):
    """N = 2^24 + 12 345 (65 585 workgroups, a ragged last one; 1.5 GB of outputs): indices, the grid-stride gain reduction
    and the ragged tail at a size where 32-bit products of (row, N) would overflow.  Sampled points against the oracle."""
    N = (1 << 24) + 12_345
    db = np.linspace(-0.05, 0.05, N)
    got = nat.sweep_host(db, n_steps=40, z_max=4.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0)
    pick = np.array([0, 1, 255, 256, 65_535, 65_536, N // 3, N // 2, (1 << 24) - 1, 1 << 24, N - 2, N - 1])
    ref = oracle.sweep(db[pick], z_max=4.0, n=40, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0)
    assert rel_err(got["a_end"][pick], ref["a_end"]) < RTOL_F64 and rel_err(got["p_max"][pick], ref["p_max"]) < RTOL_F64
    assert (got["first_bad_step"] == -1).all() and np.isfinite(got["p_max"]).all()
    gain, bi, bg, nf = nat.gain_summary_host(got["p_max"], got["first_bad_step"], P_IN[2])
    assert nf == N and bi == int(np.argmax(gain)) and bg == gain[bi]
