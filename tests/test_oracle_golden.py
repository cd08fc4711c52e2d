"""Pin the CPU oracle against golden vectors produced by the reference itself (tests/golden/gen_golden.py).

If these pass, ``oracle/`` is a trustworthy checker for the HIP path on inputs the reference was never run on.
Expected agreement is libm-vs-NumPy ulp noise amplified by the dynamics (<= 1e-12 on the main.py scenarios, <= 5e-11 elementwise at |dbeta| = 0.05 over 1e4..1e5 steps); the structurally faithful NumPy
restatement must be BIT-exact.
"""
import numpy as np
import pytest

from conftest import rel_err

TOL = 1e-12


def _a0(p_in, phase=None):
    a = np.sqrt(np.asarray(p_in, dtype=float)).astype(complex)
    if phase is not None and np.any(np.asarray(phase) != 0):
        a = a * np.exp(1j * np.asarray(phase))
    return a


def test_g1_single_point_full_trajectory(golden, oracle):
    g = golden("G1")
    z, A, bad = oracle.integrate(_a0(g["p_in"]), z_max=float(g["z_max"]), dz=float(g["dz"]), save_every=10,
                                 check_nan=True, gamma=float(g["gamma"]), alpha=float(g["alpha"]),
                                 dbeta=float(g["dbeta_sym"]))
    assert bad == -1 and A.shape == (1001, 4)
    assert np.array_equal(z, g["z"])
    assert rel_err(A, g["A"]) < TOL
    gain = 10 * np.log10(abs(A[-1, 2]) ** 2 / g["p_in"][2])
    assert abs(gain - float(g["gain_db"])) < 1e-11
    assert abs(float(g["gain_db"]) - 45.292443557977066) < 1e-12   # SURVEY G1 anchor


def test_g1_numpy_restatement_is_bit_exact(golden, oracle):
    g = golden("G1")
    z, A = oracle.np_integrate(_a0(g["p_in"]), z_max=float(g["z_max"]), dz=float(g["dz"]), save_every=10,
                               check_nan=True, gamma=float(g["gamma"]), alpha=float(g["alpha"]),
                               dbeta=float(g["dbeta_sym"]))
    assert np.array_equal(z, g["z"]) and np.array_equal(A, g["A"])


@pytest.mark.parametrize("name", ["G2", "G3"])
def test_sweeps_gain_and_a_end(golden, oracle, name):
    g = golden(name)
    n = oracle.n_steps(float(g["z_max"]), float(g["dz"]))
    r = oracle.sweep(g["dbeta"], z_max=float(g["z_max"]), n=n, save_every=10, gamma=float(g["gamma"]),
                     alpha=float(g["alpha"]), a0=_a0(g["p_in"]))
    gain = oracle.gain_from_summary(r["p_max"], r["first_bad_step"], g["p_in"][2], "db")
    assert not np.isnan(g["gain_db"]).any()
    np.testing.assert_allclose(gain, g["gain_db"], rtol=1e-11, atol=1e-11)
    if name == "G2":
        assert rel_err(r["a_end"], g["A_end"]) < TOL
        assert rel_err(r["p_max"], g["p3_max"]) < TOL
        assert g["gain_db"][0] == pytest.approx(9.6432746655328694e-16, rel=1e-9)   # floor: max is at z = 0
        assert np.argmax(g["gain_db"]) == 14
    else:
        assert np.argmax(g["gain_db"]) == 4 and g["gain_db"].max() == pytest.approx(45.48939456294602, rel=1e-12)


def test_g4_km_unit_examples(golden, oracle):
    g = golden("G4")
    # example_zero_signal: gamma 1.3 /(W km), L 0.5 km, dz 1e-3 km -> metres: /1000, *1000
    z, A, _ = oracle.integrate(_a0([0.5, 0.5, 0, 0]), z_max=500.0, dz=1.0, save_every=10, gamma=1.3 / 1000, alpha=0.0,
                               dbeta=0.0)
    assert rel_err(A[:, :2], g["zero_A"][:, :2]) < TOL and np.all(A[:, 2:] == 0) and np.all(g["zero_A"][:, 2:] == 0)
    np.testing.assert_allclose(z / 1000.0, g["zero_z"], rtol=0, atol=1e-15)
    z, A, _ = oracle.integrate(_a0([1e-1, 1e-1, 1e-4, 1e-6]), z_max=500.0, dz=0.1, save_every=10, gamma=10.0 / 1000,
                               alpha=0.0, dbeta=0.0)
    assert rel_err(A, g["seeded_A"]) < TOL


def test_g5_rhs_and_terms(golden, oracle):
    g = golden("G5")
    for i in range(g["z"].size):
        r, lin, kerr, fwm = oracle.rhs4(g["z"][i], g["a"][i], g["gamma"][i], g["alpha"][i], g["dbeta"][i])
        for got, key in ((r, "rhs"), (lin, "linear"), (kerr, "kerr"), (fwm, "fwm")):
            ref = g[key][i]
            assert np.max(np.abs(got - ref)) <= 4e-16 * max(np.max(np.abs(ref)), 1e-300) * 4


def test_g7_save_stride_and_rounding_edges(golden, oracle):
    g = golden("G7")
    a0 = _a0(g["p_in"], g["phase_in"])
    for tag in ("n1005_se10", "n1005_se1", "n3_se1", "n3_se2", "n7_se10"):
        z_max, dz, se = g[tag + "_cfg"]
        z, A, bad = oracle.integrate(a0, z_max=z_max, dz=dz, save_every=int(se), gamma=float(g["gamma"]),
                                     alpha=float(g["alpha"]), dbeta=float(g["dbeta"]))
        assert A.shape == g[tag + "_A"].shape, tag
        assert np.array_equal(z, g[tag + "_z"]), tag
        assert rel_err(A, g[tag + "_A"]) < TOL, tag
    assert g["n1005_se10_A"].shape[0] == 101      # last saved row is step 1000, not z_max (R8)
    assert g["n3_se2_A"].shape[0] == 2            # 1.0/0.3 -> n = 3 (R7)
    assert g["n7_se10_A"].shape[0] == 1           # fewer steps than save_every: only z = 0


@pytest.mark.parametrize("key,n,alpha_i", [("n1e4_a0", 10_000, 0), ("n1e4_a1", 10_000, 1)])
def test_g8_direct_dbeta_sweep(golden, oracle, key, n, alpha_i):
    g = golden("G8")
    r = oracle.sweep(g["dbeta257"], z_max=float(g["z_max"]), n=n, save_every=10, gamma=float(g["gamma"]),
                     alpha=float(g["alphas"][alpha_i]), a0=_a0(g["p_in"]))
    assert rel_err(r["a_end"], g[key + "_A_end"]) < 5e-11
    assert rel_err(r["p_end"], g[key + "_p_end"]) < 5e-11
    assert rel_err(r["p_max"], g[key + "_p_max"]) < 5e-11


def test_g8_1e5_steps(golden, oracle):
    g = golden("G8")
    r = oracle.sweep(g["dbeta33"], z_max=1000.0, n=100_000, save_every=10, gamma=float(g["gamma"]),
                     alpha=float(g["alphas"][1]), a0=_a0(g["p_in"]))
    assert rel_err(r["a_end"], g["n1e5_a1_A_end"]) < 5e-11
    assert rel_err(r["p_max"], g["n1e5_a1_p_max"]) < 5e-11


def test_g9_failure_step_index_and_nan_rows(golden, oracle):
    g = golden("G9")
    a0 = _a0(g["p_in"])
    for gam, want in zip(g["gammas"], g["first_bad_step"]):
        r = oracle.sweep(np.array([float(g["dbeta"])]), z_max=100.0, n=1000, save_every=10, gamma=float(gam),
                         alpha=0.0, a0=a0)
        assert r["first_bad_step"][0] == want, gam
    z, A, bad = oracle.integrate(a0, z_max=100.0, dz=0.1, save_every=10, check_nan=False, gamma=200.0, alpha=0.0,
                                 dbeta=0.01)
    assert bad == -1 and A.shape[0] == int(g["nocheck_n_rows"])
    assert int(np.argmax(~np.isfinite(A).all(axis=1))) == int(g["nocheck_first_bad_row"])
    assert np.isnan(g["drv_gain_bad"]).all()
    assert list(np.isnan(g["mixed_gain"])) == [False, True, False]


def test_g11_km_equals_m_and_oracle_matches(golden, oracle):
    g = golden("G11")
    np.testing.assert_allclose(g["km_gain"], g["m_gain"], rtol=1e-10)
    np.testing.assert_allclose(g["km_dbeta"], g["m_dbeta"] * 1e3, rtol=1e-13)
    z_max, dz, se, gamma, alpha = g["m_cfg"]
    n = oracle.n_steps(z_max, dz)
    r = oracle.sweep(g["m_dbeta"], z_max=z_max, n=n, save_every=int(se), gamma=gamma, alpha=alpha,
                     a0=_a0(g["p_in"], g["phase_in"]))
    gain = oracle.gain_from_summary(r["p_max"], r["first_bad_step"], g["p_in"][2], "linear")
    np.testing.assert_allclose(gain, g["m_gain"], rtol=1e-11)


def test_six_wave_reduces_to_four_wave(golden, oracle):
    """Build-defined 6-wave RHS (parity unpinned): with the second pair at zero it must reproduce the 4-wave run."""
    g = golden("G8")
    db = g["dbeta257"][::16]
    a0 = _a0(g["p_in"])
    r4 = oracle.sweep(db, z_max=1000.0, n=10_000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    r6 = oracle.sweep(db, z_max=1000.0, n=10_000, save_every=10, gamma=0.0115, alpha=1.15e-4,
                      a0=np.concatenate([a0, [0, 0]]), dbeta2=db * 0.37)
    assert rel_err(r6["a_end"][:, :4], r4["a_end"]) < 1e-12 and np.all(r6["a_end"][:, 4:] == 0)


def test_six_wave_c_oracle_matches_an_independent_numpy_statement(oracle):
    """The 6-wave model has no reference: pin the C oracle to a second, independently written NumPy form of the same
    equations (oracle.np_rhs6), over random states, and check both collapse to the 4-wave golden G5 when pair 2 is dark."""
    rng = np.random.default_rng(6)
    for _ in range(5):
        a0 = np.sqrt(rng.uniform(1e-6, 0.6, 6)) * np.exp(1j * rng.uniform(-3, 3, 6))
        g, al, d1, d2 = rng.uniform(5e-3, 2e-2), rng.choice([0.0, 2e-4]), rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05)
        want = oracle.np_integrate6(a0, z_max=40.0, n=400, gamma=g, alpha=al, dbeta1=d1, dbeta2=d2)
        z, A, bad = oracle.integrate(a0, z_max=40.0, n=400, save_every=400, gamma=g, alpha=al, dbeta=d1, dbeta2=d2)
        assert bad == -1 and rel_err(A[-1], want) < 1e-12


def test_np_rhs6_reduces_to_the_four_wave_reference_rhs(golden, oracle):
    g = golden("G5")
    for i in range(0, 64, 5):
        a6 = np.concatenate([g["a"][i], [0, 0]])
        r = oracle.np_rhs6(g["z"][i], a6, g["gamma"][i], g["alpha"][i], g["dbeta"][i], 0.123)
        assert np.max(np.abs(r[:4] - g["rhs"][i])) <= 1e-14 * np.max(np.abs(g["rhs"][i])) and np.all(r[4:] == 0)


def test_batched_numpy_form_agrees_with_the_c_oracle(oracle):
    db = np.linspace(-0.05, 0.05, 33)
    a0 = _a0([0.5, 0.5, 1e-5, 1e-5])
    got = oracle.np_sweep_batched(db, z_max=50.0, n=500, gamma=0.0115, alpha=1.15e-4, a0=a0)
    ref = oracle.sweep(db, z_max=50.0, n=500, save_every=500, gamma=0.0115, alpha=1.15e-4, a0=a0)
    assert rel_err(got, ref["a_end"]) < 1e-12


def test_g13_legacy_betas_general_taylor_and_the_grid_rows(golden, oracle):
    """G13: paths the other fixtures do not walk.  (a) only legacy beta(w_j) given: the reference's default becomes PROVIDED
    with dbeta = (b3 + b4) - (b1 + b2) (yaman_model.py:112) -- and in km units the legacy value is divided by the length
    scale TWICE upstream (simulation.py: beta_legacy / scale, then the PROVIDED config / scale again), a quirk parity
    depends on; (b) GENERAL_TAYLOR single run; (c) every row of a 4 x 9 grid through the gain + dbeta driver."""
    g = golden("G13")
    a0 = _a0(g["p_in"], g["phase_in"])
    b = g["beta_legacy_m"]
    db_m = float((b[2] + b[3]) - (b[0] + b[1]))
    bk = (b * 1e3) / 1e3
    db_km = float((bk[2] + bk[3]) - (bk[0] + bk[1])) / 1e3                                    # the double scaling
    for key, db in (("legacy_m", db_m), ("legacy_km", db_km)):
        z, A, bad = oracle.integrate(a0, z_max=200.0, n=1000, save_every=8, gamma=0.0115, alpha=1.0e-4, dbeta=db)
        assert bad == -1 and rel_err(A, g[key + "_A"]) < TOL, key
        np.testing.assert_allclose(z, g[key + "_z"], rtol=1e-15)
    assert np.max(np.abs(g["legacy_km_A"][-1] - g["legacy_m_A"][-1])) > 1e-3                  # the quirk is visible
    a_grid = _a0(g["grid_p_in"])
    for tag in ("sym", "gen"):
        db = g["grid_dbeta_" + tag].ravel()
        r = oracle.sweep(db, z_max=250.0, n=1000, save_every=5, gamma=0.0115, alpha=1.0e-4, a0=a_grid)
        gain = oracle.gain_from_summary(r["p_max"], r["first_bad_step"], g["grid_p_in"][2], "db")
        np.testing.assert_allclose(gain, g["grid_gain_" + tag].ravel(), rtol=1e-11, atol=1e-11)


def test_g15_robustness_draw_through_the_reference(golden, oracle):
    """G15: 32 single runs of the REFERENCE with dbeta, gamma, alpha, all four powers and phases, fibre length, step count
    and save stride drawn at random (SURVEY 8(d)'s robustness draw).  Final row, max signal power and four whole
    trajectories (one of them with save_every = 1)."""
    g = golden("G15")
    worst = 0.0
    for i in range(32):
        a0 = _a0(g["p_in"][i], g["phase_in"][i])
        z, A, bad = oracle.integrate(a0, z_max=float(g["L"][i]), n=int(g["n"][i]), save_every=int(g["save_every"][i]),
                                     gamma=float(g["gamma"][i]), alpha=float(g["alpha"][i]), dbeta=float(g["dbeta"][i]))
        assert bad == -1 and A.shape[0] == int(g["n_rows"][i])
        assert abs(z[-1] - g["z_last"][i]) <= 1e-12 * g["L"][i]
        worst = max(worst, rel_err(A[-1], g["A_end"][i]), rel_err(np.max(np.abs(A[:, 2]) ** 2), g["p_max"][i]))
        if f"A_full_{i}" in g.files:
            worst = max(worst, rel_err(A, g[f"A_full_{i}"]))
    assert worst < 1e-11, worst
