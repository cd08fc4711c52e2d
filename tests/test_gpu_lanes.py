"""The two float64 lane layouts of the sweep -- one lane per point (rk4_sweep_kernel) and two lanes per point
(rk4_sweep_split_kernel, chosen automatically for sweeps smaller than the chip) -- against the reference's golden
vectors and the oracle, through the C-ABI.  Same tolerances as tests/test_gpu_parity.py (1e-9 relative, float64).

The library picks the split layout by itself whenever 2*N lanes still give every wave its own SIMD (N <= 32 768 on
MI355X), so the reference-shaped API tests exercise it implicitly; here both layouts are forced explicitly.
"""
import numpy as np
import pytest

import psa_amd._native as nat
from conftest import RTOL_F64, rel_err

pytestmark = pytest.mark.gpu

LANES = [pytest.param(nat.OPT_ONE_LANE, id="one-lane"), pytest.param(nat.OPT_SPLIT_POINT, id="two-lanes"),
         pytest.param(nat.OPT_SPLIT_POINT | nat.OPT_BLOCK64, id="two-lanes-wg64"),
         pytest.param(nat.OPT_QUAD_POINT, id="four-lanes"), pytest.param(nat.OPT_QUAD_POINT | nat.OPT_BLOCK64, id="four-lanes-wg64")]
P_IN = np.array([0.5, 0.5, 1e-5, 1e-5])
A0 = np.sqrt(P_IN).astype(complex)
A06 = np.concatenate([A0, np.sqrt([2e-5, 1e-6])])


def _a0(p):
    return np.sqrt(np.asarray(p, float)).astype(complex)


@pytest.mark.parametrize("lanes", LANES)
@pytest.mark.parametrize("key,n,ai", [("n1e4_a0", 10_000, 0), ("n1e4_a1", 10_000, 1)])
def test_g8_reference_sweep_both_layouts(golden, lanes, key, n, ai):
    g = golden("G8")
    got = nat.sweep_host(g["dbeta257"], n_steps=n, z_max=1000.0, save_every=10, gamma=float(g["gamma"]),
                         alpha=float(g["alphas"][ai]), a0=_a0(g["p_in"]), extra_flags=lanes)
    assert rel_err(got["a_end"], g[key + "_A_end"]) < RTOL_F64
    assert rel_err(got["p_end"], g[key + "_p_end"]) < RTOL_F64
    assert rel_err(got["p_max"], g[key + "_p_max"]) < RTOL_F64


@pytest.mark.parametrize("lanes", LANES)
def test_g1_main_scenario_trajectory_both_layouts(golden, lanes):
    """main.py's single run (BASELINE config 1): all 1001 saved rows of the one point, 10 000 steps."""
    g = golden("G1")
    got = nat.sweep_host([float(g["dbeta_sym"])], n_steps=10_000, z_max=float(g["z_max"]), save_every=int(g["save_every"]),
                         gamma=float(g["gamma"]), alpha=float(g["alpha"]), a0=_a0(g["p_in"]), want_traj=True,
                         exact_step=True, extra_flags=lanes)
    assert got["traj"].shape == (1, 1001, 4)
    assert rel_err(got["traj"][0], g["A"]) < RTOL_F64
    assert got["first_bad_step"][0] == -1


@pytest.mark.parametrize("lanes", LANES)
def test_g7_stride_edges_both_layouts(golden, lanes):
    g = golden("G7")
    a0 = _a0(g["p_in"]) * np.exp(1j * g["phase_in"])
    for tag in ("n1005_se10", "n1005_se1", "n3_se1", "n3_se2", "n7_se10"):
        z_max, dz, se = g[tag + "_cfg"]
        got = nat.sweep_host([float(g["dbeta"])], n_steps=int(round(z_max / dz)), z_max=float(z_max), save_every=int(se),
                             gamma=float(g["gamma"]), alpha=float(g["alpha"]), a0=a0, want_traj=True, exact_step=True,
                             extra_flags=lanes)
        A = g[tag + "_A"]
        assert got["traj"].shape == (1,) + A.shape, tag
        assert rel_err(got["traj"][0], A) < RTOL_F64, tag
        assert rel_err(got["a_end"][0], A[-1]) < RTOL_F64, tag       # A[-1] is the last SAVED row


@pytest.mark.parametrize("lanes", LANES)
def test_g9_first_bad_step_both_layouts(golden, lanes):
    """The reference's FloatingPointError step indices [1, 1, 1, 2, 3, 4, 5, -1]: the two lanes of a point must agree on
    the first step after which ANY of the point's waves is non-finite."""
    g = golden("G9")
    gam = g["gammas"]
    db = np.full(gam.size, float(g["dbeta"]))
    kw = dict(n_steps=1000, z_max=100.0, save_every=10, gamma=gam, alpha=0.0, a0=_a0(g["p_in"]), extra_flags=lanes)
    exact = nat.sweep_host(db, check_nan=True, exact_step=True, **kw)
    assert np.array_equal(exact["first_bad_step"], g["first_bad_step"])
    block = nat.sweep_host(db, check_nan=True, exact_step=False, **kw)
    assert np.array_equal(block["first_bad_step"], np.where(g["first_bad_step"] >= 0, 9, -1))
    off = nat.sweep_host(db, check_nan=False, **kw)
    failed = g["first_bad_step"] >= 0
    assert (off["first_bad_step"] == -1).all()
    assert np.isnan(off["p_max"][failed]).all() and np.isfinite(off["p_max"][~failed]).all()


@pytest.mark.parametrize("lanes", LANES)
def test_unsaved_tail_failure_both_layouts(oracle, lanes):
    ref = oracle.sweep(np.array([0.01]), z_max=0.4, n=4, save_every=10, gamma=12.0, alpha=0.0, a0=A0)
    assert ref["first_bad_step"][0] == 2
    got = nat.sweep_host([0.01], n_steps=4, z_max=0.4, save_every=10, gamma=12.0, alpha=0.0, a0=A0, check_nan=True,
                         exact_step=True, extra_flags=lanes)
    assert got["first_bad_step"][0] == 2
    assert np.array_equal(got["a_end"][0], A0) and got["p_max"][0] == got["p_end"][0] == abs(A0[2]) ** 2


@pytest.mark.parametrize("N", [1, 2, 31, 32, 33, 127, 128, 129, 1000])
def test_two_lane_layout_on_ragged_sizes_six_waves(oracle, N):
    """Six waves: even lane (p1, s1, i1), odd lane (p2, s2, i2); sizes around the 32-point wave boundary."""
    rng = np.random.default_rng(7 * N + 1)
    db, db2 = rng.uniform(-0.08, 0.08, N), rng.uniform(-0.08, 0.08, N)
    gam = rng.uniform(5e-3, 2e-2, N)
    a0 = np.sqrt(rng.uniform(1e-6, 0.8, (N, 6))) * np.exp(1j * rng.uniform(-3.1, 3.1, (N, 6)))
    ref = oracle.sweep(db, dbeta2=db2, z_max=60.0, n=600, save_every=7, gamma=gam, alpha=2e-4, a0=a0)
    one = nat.sweep_host(db, dbeta2=db2, n_steps=600, z_max=60.0, save_every=7, gamma=gam, alpha=2e-4, a0=a0,
                         extra_flags=nat.OPT_ONE_LANE, want_traj=True)
    two = nat.sweep_host(db, dbeta2=db2, n_steps=600, z_max=60.0, save_every=7, gamma=gam, alpha=2e-4, a0=a0,
                         extra_flags=nat.OPT_SPLIT_POINT, want_traj=True)
    for got in (one, two):
        assert rel_err(got["a_end"], ref["a_end"]) < RTOL_F64
        assert rel_err(got["p_end"], ref["p_end"]) < RTOL_F64 and rel_err(got["p_max"], ref["p_max"]) < RTOL_F64
        assert (got["first_bad_step"] == -1).all()
    assert rel_err(two["traj"], one["traj"]) < 1e-11                 # every saved row of every wave
    i = N // 2
    z, A, _ = oracle.integrate(a0[i], z_max=60.0, n=600, save_every=7, gamma=gam[i], alpha=2e-4, dbeta=db[i], dbeta2=db2[i])
    assert rel_err(two["traj"][i], A) < RTOL_F64


def test_two_lane_lossless_and_dark_pair_reduction(oracle):
    """alpha == 0 selects the split kernel's lossless instantiation; with pair 2 dark the six-wave point equals the
    four-wave point at dbeta_1 in both layouts."""
    db = np.linspace(-0.05, 0.05, 200)
    a_dark = np.concatenate([A0, [0.0, 0.0]])
    four = nat.sweep_host(db, n_steps=2000, z_max=200.0, save_every=10, gamma=0.0115, alpha=0.0, a0=A0,
                          extra_flags=nat.OPT_SPLIT_POINT)
    ref = oracle.sweep(db, z_max=200.0, n=2000, save_every=10, gamma=0.0115, alpha=0.0, a0=A0)
    assert rel_err(four["a_end"], ref["a_end"]) < RTOL_F64
    np.testing.assert_allclose((np.abs(four["a_end"]) ** 2).sum(1), P_IN.sum(), rtol=1e-11)   # lossless: power conserved
    for lanes in (nat.OPT_ONE_LANE, nat.OPT_SPLIT_POINT):
        six = nat.sweep_host(db, dbeta2=0.3 * db, n_steps=2000, z_max=200.0, save_every=10, gamma=0.0115, alpha=0.0,
                             a0=a_dark, extra_flags=lanes)
        assert rel_err(six["a_end"][:, :4], ref["a_end"]) < RTOL_F64
        assert np.all(six["a_end"][:, 4:] == 0)


def test_automatic_layout_choice_is_invisible_in_the_results(oracle):
    """N = 32 768 (BASELINE config 5's per-GPU shard, the largest sweep that still gets two lanes per point) and
    N = 32 769 (one lane per point) against the oracle on sampled points; explicit flags reproduce the automatic choice
    bit for bit."""
    rng = np.random.default_rng(3)
    for N in (32_768, 32_769):
        db = rng.uniform(-0.05, 0.05, N)
        kw = dict(n_steps=2000, z_max=200.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0)
        auto = nat.sweep_host(db, **kw)
        forced = nat.sweep_host(db, extra_flags=(nat.OPT_SPLIT_POINT if N <= 32_768 else nat.OPT_ONE_LANE), **kw)
        assert np.array_equal(auto["a_end"], forced["a_end"]) and np.array_equal(auto["p_max"], forced["p_max"])
        pick = rng.choice(N, 16, replace=False)
        ref = oracle.sweep(db[pick], z_max=200.0, n=2000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A0)
        assert rel_err(auto["a_end"][pick], ref["a_end"]) < RTOL_F64
        assert (auto["first_bad_step"] == -1).all()


@pytest.mark.parametrize("se", [7, 50, 64, 192, 1024])
@pytest.mark.parametrize("lanes", [nat.OPT_ONE_LANE, nat.OPT_SPLIT_POINT, nat.OPT_QUAD_POINT])
def test_exact_first_bad_step_comes_from_a_replay_of_the_failing_block(oracle, lanes, se):
    """PSA_OPT_EXACT_STEP in float64 costs nothing in the steady-state loop: the forward pass tests once per saved row and a
    wave with a newly failing point REPLAYS the steps since the previous test with the reference's per-step test
    (integrators.py:132-135).  Sixty points get a graded NEGATIVE loss (gain), so they blow up at step indices spread from 2
    to beyond 100 -- inside saved blocks, across the 32-step chunks and 64-step re-seeds the replay must repeat, and
    (se = 1024: no saved row at all, one replay of the whole run) in the unsaved tail -- while most points and whole waves
    stay finite.  Three bars:
      * every se: the block-mode run of the same sweep must name the block the exact index lies in, and finite points must be
        bit-identical in both modes (the forward pass is the same code);
      * every se: the phase re-seeds sit on the absolute step grid, so the forward pass is bit-identical to the
        save_every = 1 trajectory run, whose per-row test IS a per-step test: the replayed index must EQUAL it, for every
        failing point, chaotic or not;
      * against the oracle: equal for the points whose blow-up is abrupt; the late ones (|alpha| ~ 1: hundreds of radians of
        nonlinear phase before they fail) are chaotic -- implementations 1e-12 apart fail at different steps -- and only
        have to fail."""
    n, N = 450, 331
    rng = np.random.default_rng(se)
    db = rng.uniform(-0.05, 0.05, N)
    al = np.full(N, 1.15e-4)
    hot = rng.choice(N, 60, replace=False)
    al[hot] = -np.geomspace(1.0, 60.0, hot.size)
    a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
    kw = dict(n_steps=n, z_max=45.0, gamma=0.0115, alpha=al, a0=a0, extra_flags=lanes)
    ref = oracle.sweep(db, z_max=45.0, n=n, save_every=se, gamma=0.0115, alpha=al, a0=a0)
    got = nat.sweep_host(db, save_every=se, exact_step=True, **kw)
    blk = nat.sweep_host(db, save_every=se, exact_step=False, **kw)
    want = ref["first_bad_step"]
    failed = want >= 0
    assert failed.sum() >= 40 and len(set(want[failed])) >= 20 and want.max() >= 100      # a real spread of indices
    assert np.array_equal(got["first_bad_step"] >= 0, failed)
    exact = got["first_bad_step"][failed]
    last_saved = n // se * se
    assert np.array_equal(blk["first_bad_step"][failed], np.where(exact // se * se + se <= last_saved, exact // se * se + se - 1, n - 1))
    assert (blk["first_bad_step"][~failed] == -1).all()
    ok = ~failed
    assert rel_err(got["a_end"][ok], ref["a_end"][ok]) < RTOL_F64 and np.array_equal(got["a_end"][ok], blk["a_end"][ok])
    assert np.array_equal(got["p_max"][ok], blk["p_max"][ok])
    every = nat.sweep_host(db, save_every=1, exact_step=True, want_traj=True, **kw)   # per-row == per-step test, no replay
    assert np.array_equal(got["first_bad_step"], every["first_bad_step"])
    if last_saved:
        assert np.array_equal(got["a_end"][ok], every["traj"][ok, last_saved, :])
    abrupt = failed & (al < -3.0)
    assert abrupt.sum() >= 30 and np.array_equal(got["first_bad_step"][abrupt], want[abrupt])


@pytest.mark.parametrize("N", [1, 2, 15, 16, 17, 63, 64, 65, 1000, 16_384, 16_385])
def test_four_lane_layout_matches_the_others(oracle, N):
    """Four lanes per point (one wave of the 4-wave model per lane, PSA_OPT_QUAD_POINT): sizes around the 16-point wave
    boundary and the largest sweep that takes this layout by itself; per-point gamma / alpha / amplitudes with phases, a
    lossless run, saved rows, every check mode -- against the oracle (1e-9), the two-lane layout (1e-11: the layouts differ
    by which factor of A_u*A_v is the FMA's exact one) and, for N <= 16 384, the library's own choice (bit for bit)."""
    rng = np.random.default_rng(N)
    db = rng.uniform(-0.08, 0.08, N)
    gam, al = rng.uniform(5e-3, 2e-2, N), rng.uniform(0.0, 3e-4, N)
    a0 = np.sqrt(rng.uniform([0.2, 0.2, 1e-6, 1e-6], [0.8, 0.8, 1e-3, 1e-3], (N, 4))) * np.exp(1j * rng.uniform(-3.1, 3.1, (N, 4)))
    n, se = (600, 7) if N <= 1000 else (200, 10)
    small = N <= 1000
    kw = dict(n_steps=n, z_max=0.1 * n, save_every=se, gamma=gam, alpha=al, a0=a0, want_traj=small)
    pick = np.arange(N) if small else rng.choice(N, 64, replace=False)
    ref = oracle.sweep(db[pick], z_max=0.1 * n, n=n, save_every=se, gamma=gam[pick], alpha=al[pick], a0=a0[pick])
    two = nat.sweep_host(db, extra_flags=nat.OPT_SPLIT_POINT, **kw)
    for flags in (nat.OPT_QUAD_POINT, nat.OPT_QUAD_POINT | nat.OPT_BLOCK64):
        for check in (dict(check_nan=True, exact_step=True), dict(check_nan=True, exact_step=False), dict(check_nan=False)):
            four = nat.sweep_host(db, extra_flags=flags, **kw, **check)
            assert rel_err(four["a_end"][pick], ref["a_end"]) < RTOL_F64 and rel_err(four["p_max"][pick], ref["p_max"]) < RTOL_F64
            assert rel_err(four["p_end"][pick], ref["p_end"]) < RTOL_F64 and (four["first_bad_step"] == -1).all()
            assert rel_err(four["a_end"], two["a_end"]) < 1e-11
            if small:
                assert rel_err(four["traj"], two["traj"]) < 1e-11 and np.array_equal(four["traj"][:, -1, :], four["a_end"])
    auto = nat.sweep_host(db, **kw)
    forced = nat.sweep_host(db, extra_flags=(nat.OPT_QUAD_POINT if N <= 16_384 else nat.OPT_SPLIT_POINT), **kw)
    assert np.array_equal(auto["a_end"], forced["a_end"]) and np.array_equal(auto["p_max"], forced["p_max"])
    lossless = nat.sweep_host(db, extra_flags=nat.OPT_QUAD_POINT, **{**kw, "alpha": 0.0, "want_traj": False})
    power = (np.abs(lossless["a_end"]) ** 2).sum(1)
    np.testing.assert_allclose(power, (np.abs(a0) ** 2).sum(1), rtol=1e-11)


def test_four_lanes_are_for_the_four_wave_model_only():
    with pytest.raises(nat.PsaNativeError) as e:
        nat.sweep_host(np.zeros(4), dbeta2=np.zeros(4), n_steps=10, z_max=1.0, save_every=1, gamma=0.01, alpha=0.0, a0=A06,
                       extra_flags=nat.OPT_QUAD_POINT)
    assert e.value.code == -11
    with pytest.raises(nat.PsaNativeError) as e:
        nat.sweep_host(np.zeros(4), n_steps=10, z_max=1.0, save_every=1, gamma=0.01, alpha=0.0, a0=A0,
                       extra_flags=nat.OPT_QUAD_POINT | nat.OPT_SPLIT_POINT)
    assert e.value.code == -11
