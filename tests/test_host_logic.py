"""Host-side logic (no GPU): config carriers, the dbeta producers against the reference's golden values, the
generic callable stepper (the reference's own passing integrator tests, restated), argument validation of the
drivers, and -- importantly -- that the product path REFUSES to run without the HIP device instead of falling
back to a CPU implementation."""
import math

import numpy as np
import pytest

import psa_amd
from psa_amd import (_native, config, constants, dispersion, frequency_plan, integrators, parameters,
                     phase_matching, scan_mismtach, simulation, yaman_model)
from psa_amd.phase_matching import PhaseMatchingConfig, PhaseMatchingMethod

HAS_GPU = _native.device_count() > 0


# ---- config (reference tests.py:27-88) -------------------------------------------------------------------
def test_default_config_is_valid():
    cfg = config.default_simulation_config()
    config.validate_config(cfg)
    assert (cfg.z_max, cfg.dz, cfg.save_every, cfg.check_nan, cfg.verbose) == (0.5, 1e-3, 10, True, False)
    assert cfg.integrator.lower() == "rk4"
    with pytest.raises(Exception):
        cfg.dz = 1.0   # frozen


@pytest.mark.parametrize("kw", [dict(z_max=0.0), dict(z_max=-1.0), dict(dz=0.0), dict(dz=2.0, z_max=1.0),
                                dict(integrator="euler"), dict(save_every=0)])
def test_validate_config_rejects_invalid(kw):
    with pytest.raises(ValueError):
        config.validate_config(config.custom_simulation_config(**kw))


def test_constants():
    assert isinstance(constants.c, float) and constants.c == 299_792_458.0


def test_n_steps_rounding_matches_python_round():
    assert config.n_steps_of(1.0, 0.3) == 3 and config.n_steps_of(1000.0, 0.1) == 10000
    assert config.n_steps_of(2.5, 1.0) == 2 and config.n_steps_of(3.5, 1.0) == 4   # round-half-even


# ---- dbeta producers vs golden (bit-exact where the reference is scalar Python) -------------------------
def test_g1_plan_dispersion_dbeta(golden):
    g = golden("G1")
    om = frequency_plan.plan_from_wavelengths(*g["lam"])
    assert np.array_equal(om, g["omega"])
    sp = frequency_plan.infer_symmetry_from_omegas(*om)
    assert (sp.omega_c, sp.omega_d, sp.Omega) == (float(g["omega_c"]), float(g["omega_d"]), float(g["Omega"]))
    assert frequency_plan.lambda_from_omega(sp.omega_c) == float(g["lambda_c"])
    d = dispersion.dispersion_params_from_D_S(float(g["lambda_c"]), 0.02, 0.02, 0, D_units="ps/nm/km",
                                              S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km", omega_ref=sp.omega_c)
    assert (d.beta2, d.beta3, d.beta4) == (float(g["beta2"]), float(g["beta3"]), float(g["beta4"]))
    assert d.beta2 == -2.5673272511996503e-29 and d.beta4 == -1.632334080166221e-55      # SURVEY anchors (R4 quirk)
    assert dispersion.delta_beta_symmetric(sp.omega_c, sp.omega_d, sp.Omega, d) == float(g["dbeta_sym"])
    assert dispersion.delta_beta_from_omegas(om, d) == float(g["dbeta_gen"])
    res = phase_matching.compute_phase_mismatch(om, d, PhaseMatchingConfig())
    assert res.delta_beta == float(g["dbeta_sym"]) and res.symmetric == sp


@pytest.mark.parametrize("name", ["G2", "G3"])
def test_sweep_dbeta_batch_is_bit_exact(golden, name):
    g = golden(name)
    d = dispersion.DispersionParams(omega_ref=float(g["omega_ref"]), beta2=float(g["beta2"]), beta3=float(g["beta3"]),
                                    beta4=float(g["beta4"]))
    om, ok = frequency_plan.plan_from_wavelengths_batch(float(g["lambda_p1"]), float(g["lambda_p2"]), g["lambda3"])
    db, ok2 = phase_matching.compute_phase_mismatch_batch(om, d, PhaseMatchingConfig())
    assert ok.all() and ok2.all()
    assert np.array_equal(db, g["dbeta"])
    assert np.array_equal(g["lambda3"] * 1e9, g["x"])


def test_g10_dispersion_builder_and_methods(golden):
    g = golden("G10")
    units = {0: ("SI", "SI", "SI"), 1: ("ps/nm/km", "ps/nm^2/km", "ps/nm^3/km")}
    for lam, D, S, dS, u, wref, b2, b3, b4 in g["ds_rows"]:
        du, su, dsu = units[int(u)]
        d = dispersion.dispersion_params_from_D_S(lam, D, None if np.isnan(S) else S, None if np.isnan(dS) else dS,
                                                  D_units=du, S_units=su, dSdlmbd_units=dsu)
        assert (d.omega_ref, d.beta2, d.beta3, d.beta4) == (wref, b2, b3, b4)
    dv = g["disp"]
    d = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3], extra={6: dv[4]})
    om, ok = frequency_plan.plan_from_wavelengths_batch(g["lp1"], g["lp2"], g["l3"])
    assert ok.all() and np.array_equal(om, g["omega"])
    oc, od, Om, oks = frequency_plan.symmetry_arrays(*om.T)
    assert oks.all() and np.array_equal(np.stack([oc, od, Om], 1), g["sym"])
    cases = {"sym24": dict(method="symmetric_even", even_orders=(2, 4)), "sym2": dict(method="symmetric_even", even_orders=(2,)),
             "sym246": dict(method="symmetric_even", even_orders=(2, 4, 6)), "gen4": dict(method="general_taylor", max_order=4),
             "gen2": dict(method="general_taylor", max_order=2), "gen6": dict(method="general_taylor", max_order=6)}
    for key, kw in cases.items():
        cfg = PhaseMatchingConfig(**kw)
        scalar = np.array([phase_matching.compute_phase_mismatch(o, d, cfg).delta_beta for o in om])
        assert np.array_equal(scalar, g[key]), key                       # scalar API: bit-exact
        batch, okb = phase_matching.compute_phase_mismatch_batch(om, d, cfg)
        assert okb.all()
        np.testing.assert_allclose(batch, g[key], rtol=1e-14, atol=0)    # array pow may differ by an ulp


def test_batch_plan_marks_impossible_points_invalid(golden):
    g = golden("G9")
    om, ok = frequency_plan.plan_from_wavelengths_batch(1550e-9, 1558e-9, g["mixed_lambda3"])
    assert list(ok) == [True, False, True]
    with pytest.raises(ValueError):
        frequency_plan.plan_from_wavelengths(1550e-9, 1558e-9, 0.7e-6)
    d = dispersion.DispersionParams(omega_ref=float(g["drv_omega_ref"]), beta2=float(g["drv_beta2"]),
                                    beta3=float(g["drv_beta3"]), beta4=float(g["drv_beta4"]))
    db, okd = phase_matching.compute_phase_mismatch_batch(om, d, PhaseMatchingConfig())
    assert list(okd) == [True, False, True]
    np.testing.assert_array_equal(np.isnan(db), np.isnan(g["mixed_dbeta"]))
    assert np.array_equal(db[[0, 2]], g["mixed_dbeta"][[0, 2]])


def test_frequency_plan_scalar_errors_and_symmetric_plan():
    with pytest.raises(ValueError):
        frequency_plan.omega_from_lambda(0.0)
    with pytest.raises(TypeError):
        frequency_plan.omega_from_lambda("abc")
    with pytest.raises(ValueError):
        frequency_plan.SymmetricPlan(omega_c=1.0, omega_d=2.0, Omega=0.1)
    with pytest.raises(ValueError):
        frequency_plan.plan_from_omegas(1.0, 1.0, 1.0, 2.0)          # energy conservation
    sp = frequency_plan.SymmetricPlan(omega_c=10.0, omega_d=1.0, Omega=3.0)
    assert list(sp.omegas()) == [11.0, 9.0, 13.0, 7.0]
    assert list(frequency_plan.plan_from_symmetry(10.0, 1.0, 3.0)) == [11.0, 9.0, 13.0, 7.0]
    assert "pump1" in frequency_plan.describe_plan(sp.omegas())
    assert frequency_plan.f_from_omega(frequency_plan.omega_from_f(2e14)) == pytest.approx(2e14)


def test_phase_matching_config_validation():
    assert PhaseMatchingConfig(method="provided", provided_delta_beta=1).provided_delta_beta == 1.0
    for bad in (dict(method="nope"), dict(max_order=-1), dict(even_orders=()), dict(even_orders=(3,)),
                dict(atol=-1.0), dict(method="provided")):
        with pytest.raises((ValueError, TypeError)):
            PhaseMatchingConfig(**bad)
    with pytest.raises(ValueError):
        phase_matching.compute_phase_mismatch([1.0, 1.0, 1.0, 1.0], None, PhaseMatchingConfig())
    assert PhaseMatchingConfig(method="provided", provided_delta_beta=2.0).scaled(1000.0).provided_delta_beta == 2e-3


def test_parameters_carriers():
    w = parameters.WavesParams.from_symmetry(10.0, 1.0, 3.0)
    f = parameters.FiberParams(length_m=1.0, gamma_W_m=0.01, alpha_1_m=0.0)
    mp = parameters.make_model_params(waves=w, fiber=f, grid=parameters.SimulationGrid(dz_m=0.1))
    assert mp.cache.delta_beta_1_m is None and mp.cache.symmetric == w.symmetric
    mp.cache.set_phase_mismatch(0.25)
    assert yaman_model.extract_gamma_alpha_dbeta(mp) == (0.01, 0.0, 0.25)
    with pytest.raises(ValueError):
        mp.cache.set_phase_mismatch(float("nan"))
    for bad in (dict(length_m=0.0, gamma_W_m=1.0), dict(length_m=1.0, gamma_W_m=1.0, alpha_1_m=-1.0)):
        with pytest.raises(ValueError):
            parameters.FiberParams(**bad)
    with pytest.raises(ValueError):
        parameters.WavesParams(omega=[1.0, 2.0, 3.0])

    class Legacy:   # legacy containers: gamma / alpha / beta fallbacks (yaman_model.py:59-116)
        class fiber:
            gamma, alpha, beta = 2.0, 0.5, [1.0, 2.0, 4.0, 8.0]
    assert yaman_model.extract_gamma_alpha_dbeta(Legacy) == (2.0, 0.5, 9.0)


# ---- generic callable stepper: the reference's passing integrator tests (tests.py:146-226) + G6 ------------
def test_rk4_step_matches_exp_for_simple_ode(golden):
    y1 = integrators.rk4_step(lambda z, y, p: y, 0.0, np.array([1.0]), 0.1, None)
    np.testing.assert_allclose(y1, [math.exp(0.1)], rtol=1e-7, atol=0)
    assert np.array_equal(y1, golden("G6")["rk4_step_exp"])


def test_integrate_interval_shapes_and_saving(golden):
    z_out, y_out = integrators.integrate_interval(lambda z, y, p: y, 1.0, 0.1, np.array([1.0]), None, save_every=2,
                                                  check_nan=True)
    assert z_out.shape == (6,) and y_out.shape == (6, 1)
    np.testing.assert_allclose(z_out, [0.0, 0.2, 0.4, 0.6, 0.8, 1.0], rtol=0.0, atol=1e-15)
    np.testing.assert_allclose(y_out[:, 0], np.exp(z_out), rtol=0.0, atol=3e-6)
    g = golden("G6")
    assert np.array_equal(z_out, g["interval_z"]) and np.array_equal(y_out, g["interval_y"])
    M = g["lin_M"]
    z2, y2 = integrators.integrate_interval(lambda z, y, p: M @ y * (1.0 + 0.1 * z), 2.0, 0.01,
                                            np.array([1.0 + 0j, 0.5j]), None, save_every=7)
    assert np.array_equal(z2, g["lin_z"]) and np.array_equal(y2, g["lin_y"])


def test_integrate_fixed_step_rejects_bad_inputs():
    f = lambda z, y, p: y  # noqa: E731
    with pytest.raises(ValueError):
        integrators.integrate_fixed_step(f, np.array([[0.0, 0.1]]), np.array([1.0]), None)
    with pytest.raises(ValueError):
        integrators.integrate_fixed_step(f, np.array([0.0, 0.1]), np.array([1.0]), None, save_every=0)
    for kw in (dict(z_max=0.0, dz=0.1), dict(z_max=1.0, dz=0.0)):
        with pytest.raises(ValueError):
            integrators.integrate_interval(f, y0=np.array([1.0]), params=None, **kw)


def test_check_nan_raises():
    f = lambda z, y, p: np.array([np.nan])  # noqa: E731
    with pytest.raises(FloatingPointError, match="step 0"):
        integrators.integrate_interval(f, 0.2, 0.1, np.array([0.0]), None, save_every=1, check_nan=True)
    z_out, y_out = integrators.integrate_interval(f, 0.2, 0.1, np.array([0.0]), None, save_every=1, check_nan=False)
    assert np.isnan(y_out).any()


# ---- driver argument validation happens before any device work ------------------------------------------------
def _drv_kwargs(**over):
    kw = dict(cfg=config.custom_simulation_config(z_max=10.0, dz=0.1), lambda_p1_m=1550e-9, lambda_p2_m=1558e-9,
              lambda_signal_m=[1552e-9], gamma=0.0115, alpha=0.0, p_in=[0.1, 0.1, 1e-7, 1e-7],
              dispersion=dispersion.DispersionParams(omega_ref=1.2e15, beta2=-1e-28), show=False)
    kw.update(over)
    return kw


@pytest.mark.parametrize("over", [dict(lambda_signal_m=[]), dict(lambda_signal_m=[-1.0]), dict(p_in=[1, 1, 1]),
                                  dict(p_in=[0.1, 0.1, 0.0, 0.0]), dict(p_in=[0.1, -0.1, 1e-7, 0.0]),
                                  dict(phase_in=[0, 0, 0]), dict(gain_unit="nepers"), dict(xscale="cubic"),
                                  dict(return_wavelength_unit="furlong", lambda_signal_m=[0.7e-6])])
def test_drivers_raise_value_error_on_malformed_arguments(over):
    with pytest.raises(ValueError):
        scan_mismtach.plot_max_signal_gain_vs_lambda_signal(**_drv_kwargs(**over))
    with pytest.raises(ValueError):
        scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(**_drv_kwargs(**over))


def test_driver_log_scale_rules_and_missing_dispersion():
    with pytest.raises(ValueError):
        scan_mismtach.plot_max_signal_gain_vs_lambda_signal(**_drv_kwargs(yscale="log", gain_unit="dB"))
    with pytest.raises(ValueError):
        scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(**_drv_kwargs(yscale_gain="log"))
    with pytest.raises(ValueError):
        scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(**_drv_kwargs(dispersion=None))


def test_driver_per_point_failures_become_nan_without_touching_the_gpu():
    # every point invalid (omega4 <= 0) or the cfg invalid -> NaN everywhere, no launch, no exception
    x, g = scan_mismtach.plot_max_signal_gain_vs_lambda_signal(**_drv_kwargs(lambda_signal_m=[0.7e-6, 0.6e-6]))
    assert np.isnan(g).all() and np.allclose(x, [700.0, 600.0])
    bad_cfg = config.custom_simulation_config(z_max=1.0, dz=2.0)
    x, g, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(**_drv_kwargs(cfg=bad_cfg))
    assert np.isnan(g).all() and np.isfinite(db).all()
    x, g = scan_mismtach.plot_max_signal_gain_vs_lambda_signal(**_drv_kwargs(dispersion=None))
    assert np.isnan(g).all()


def test_select_power_metric():
    P = np.array([1.0, 5.0, 2.0])
    assert scan_mismtach._select_power_metric(P, "end") == 2.0 and scan_mismtach._select_power_metric(P, "max") == 5.0
    with pytest.raises(ValueError):
        scan_mismtach._select_power_metric(P, "avg")
    with pytest.raises(ValueError):
        scan_mismtach._select_power_metric(P[None], "end")


def test_run_single_simulation_validation_errors():
    cfg = config.custom_simulation_config(z_max=1.0, dz=0.1)
    pm = PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=0.0)
    ok = dict(gamma=1.0, alpha=0.0, omega=[1.0, 1.0, 1.0, 1.0], p_in=[1, 1, 0, 0], phase_matching_cfg=pm)
    for over, exc in ((dict(omega=[1, 1, 1]), ValueError), (dict(omega=[1, 1, -1, 1]), ValueError),
                      (dict(p_in=[1, 1, -1, 0]), ValueError), (dict(length_unit="mile"), ValueError),
                      (dict(dispersion="x"), TypeError), (dict(phase_matching_cfg="x"), TypeError),
                      (dict(phase_matching_cfg=None), ValueError), (dict(beta_legacy=[1, 2, 3]), ValueError)):
        with pytest.raises(exc):
            simulation.run_single_simulation(cfg, **{**ok, **over})
    a0 = simulation.make_initial_amplitudes([0.25, 1.0, 0.0, 4.0], [0.0, np.pi / 2, 0.0, np.pi])
    np.testing.assert_allclose(a0, [0.5, 1j, 0.0, -2.0], atol=1e-15)
    assert np.array_equal(simulation.make_initial_amplitudes([1e-7, 0, 0, 0]), np.sqrt([1e-7, 0, 0, 0]).astype(complex))


# ---- the product path has NO CPU fallback ---------------------------------------------------------------------
@pytest.mark.skipif(HAS_GPU, reason="only meaningful on a box without a GPU")
def test_product_path_fails_loudly_without_a_gpu():
    cfg = config.custom_simulation_config(z_max=1.0, dz=0.1)
    pm = PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=0.0)
    with pytest.raises(_native.PsaNativeError, match="no CPU fallback"):
        simulation.run_single_simulation(cfg, gamma=1.0, alpha=0.0, omega=[1.0] * 4, p_in=[1, 1, 0, 0],
                                         phase_matching_cfg=pm)
    with pytest.raises(_native.PsaNativeError, match="no CPU fallback"):
        scan_mismtach.plot_max_signal_gain_vs_lambda_signal(**_drv_kwargs())
    with pytest.raises(_native.PsaNativeError, match="no CPU fallback"):
        yaman_model.rhs_yaman_simplified(0.0, np.ones(4, complex), _cached_params())


def _cached_params():
    mp = parameters.make_model_params(waves=parameters.WavesParams(omega=[1.0] * 4),
                                      fiber=parameters.FiberParams(length_m=1.0, gamma_W_m=1.0),
                                      grid=parameters.SimulationGrid(dz_m=0.1))
    mp.cache.set_phase_mismatch(0.0)
    return mp


def test_product_package_never_imports_the_oracle():
    import os, re
    pkg = os.path.dirname(psa_amd.__file__)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libpsa_oracle" not in src and "psa_oracle" not in src, f


def test_g13_host_producers_against_the_reference_grid_rows(golden):
    """The 4 x 9 (lambda_p2 x lambda_signal) grid of golden G13, which the reference produced row by row through its
    gain + dbeta driver: the array producer must give the same dbeta for the whole grid at once -- bit-equal for the
    symmetric closed form, within the array-pow ulp for GENERAL_TAYLOR -- and the legacy-beta defaults must reproduce the
    reference's PROVIDED value including its km double scaling."""
    from psa_amd.scan_mismtach import _grid_dbeta
    g = golden("G13")
    dv = g["disp"]
    d = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3])
    db, ok = _grid_dbeta(1550e-9, g["lambda2"], g["lambda3"], d, PhaseMatchingConfig(), "host", 0)
    assert ok.all() and np.array_equal(db.reshape(4, 9), g["grid_dbeta_sym"])
    dbg, ok = _grid_dbeta(1550e-9, g["lambda2"], g["lambda3"], d, PhaseMatchingConfig(method="general_taylor", max_order=4), "host", 0)
    assert ok.all()
    np.testing.assert_allclose(dbg.reshape(4, 9), g["grid_dbeta_gen"], rtol=1e-14, atol=0)
    from psa_amd import config, simulation
    b = g["beta_legacy_m"]
    pre = simulation._prepare(config.custom_simulation_config(z_max=200.0, dz=0.2, save_every=8), gamma=0.0115, alpha=1e-4,
                              dispersion=None, phase_matching_cfg=None, beta_legacy=b, length_unit="m")
    assert pre["pm"].config.provided_delta_beta == float((b[2] + b[3]) - (b[0] + b[1]))
    pre = simulation._prepare(config.custom_simulation_config(z_max=0.2, dz=0.2e-3, save_every=8), gamma=11.5, alpha=0.1,
                              dispersion=None, phase_matching_cfg=None, beta_legacy=b * 1e3, length_unit="km")
    bk = (b * 1e3) / 1e3
    assert pre["pm"].config.provided_delta_beta == float((bk[2] + bk[3]) - (bk[0] + bk[1])) / 1e3


def test_public_call_surface_equals_the_reference(golden):
    """tests/golden/api_signatures.json (written by gen_golden.py from the reference): every public function of the
    hot-path modules exists here with the same parameter names, kinds and defaults, and every public dataclass with the same
    fields, so reference code switches by changing imports.  scan_mismatch_seeded_signal, which upstream
    cannot run (it passes a keyword run_single_simulation does not have), has a working counterpart instead:
    scan_dbeta_seeded_signal; plot_dbeta_vs_lambda_signal (upstream: NaN for every point) keeps its signature."""
    import dataclasses
    import importlib
    import inspect
    import json
    import os
    spec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "api_signatures.json")))
    dead = {("scan_mismtach", "scan_mismatch_seeded_signal")}
    checked = 0
    for modname, entry in spec.items():
        mod = importlib.import_module("psa_amd." + modname)
        for name, want in entry.items():
            if (modname, name) in dead:
                assert not hasattr(mod, name)
                continue
            assert hasattr(mod, name), f"{modname}.{name} is missing"
            obj = getattr(mod, name)
            if want["kind"] == "class":
                if want["fields"] is not None:
                    assert [f.name for f in dataclasses.fields(obj)] == want["fields"], f"{modname}.{name}"
            else:
                got = [[n, q.kind.name, repr(q.default) if q.default is not inspect._empty else "<required>"]
                       for n, q in inspect.signature(obj).parameters.items()]
                assert got[:len(want["params"])] == want["params"], f"{modname}.{name}: {got} != {want['params']}"
            checked += 1
    assert checked >= 55


def test_error_contract_equals_the_reference():
    """tests/golden/error_contract.json records what the REFERENCE raises (exception type, message) for the 44 invalid calls
    of tests/golden/error_cases.py; the package must raise the same exception TYPE for every one of them (or, like the
    reference, nothing), and it must do so in argument validation -- this test runs without a GPU.  Messages are free to
    differ in wording; where they are identical is reported."""
    import importlib
    import json
    import os
    import sys
    here = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, here)
    try:
        import error_cases
    finally:
        sys.path.remove(here)
    want = json.load(open(os.path.join(here, "error_contract.json")))
    got = error_cases.evaluate(lambda m: importlib.import_module("psa_amd." + m))
    assert set(got) == set(want) and len(want) >= 44
    wrong = {k: (want[k], got[k]) for k in want if want[k][0] != got[k][0]}
    assert not wrong, wrong
    same_text = sum(want[k][1] == got[k][1] for k in want)
    print(f"{len(want)} invalid calls: same exception type in all, identical message in {same_text}")
    assert same_text >= 30


def test_scalar_host_api_is_bit_identical_to_the_reference_along_a_random_walk():
    """tests/golden/host_api_records.json.gz: ~1 850 return values the REFERENCE produced along the seeded walk of
    host_api_cases.py -- frequency plans (every 7th without an idler: exception types recorded), symmetric decompositions,
    describe_plan text, dispersion builders incl. the dS/dlambda slot quirk, Taylor and symmetric mismatch in several orders,
    every phase-matching method, configuration objects.  The package must reproduce every record exactly (floats bit for bit)."""
    import gzip
    import importlib
    import json
    import os
    import sys
    here = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, here)
    try:
        import host_api_cases
    finally:
        sys.path.remove(here)
    with gzip.open(os.path.join(here, "host_api_records.json.gz"), "rt", encoding="utf-8") as f:
        want = json.load(f)
    got = json.loads(json.dumps(host_api_cases.evaluate(lambda m: importlib.import_module("psa_amd." + m))))
    assert set(got) == set(want) and len(want) > 1800
    wrong = [k for k in want if want[k] != got[k]]
    assert not wrong, (wrong[:5], want[wrong[0]], got[wrong[0]])
    assert sum(1 for v in want.values() if isinstance(v, list) and v and v[0] == "EXC") > 50


def test_batch_producers_agree_with_the_scalar_api_point_by_point():
    """Property test (hypothesis): for arbitrary wavelength triples -- in band, far out of band, zero, negative, NaN -- the
    array producers that feed the sweep kernel (plan_from_wavelengths_batch, compute_phase_mismatch_batch) are valid exactly
    where the scalar reference-shaped functions do not raise, and give the scalar dbeta there (bit-equal for the symmetric
    closed form with squares only, within an ulp-scale relative error when x**4 goes through NumPy's array pow)."""
    from hypothesis import given, settings, strategies as st
    lam = st.one_of(st.floats(1.2e-6, 1.9e-6), st.floats(0.3e-6, 6e-6), st.sampled_from([0.0, -1.5e-6, float("nan"), float("inf")]))
    d = dispersion.DispersionParams(omega_ref=frequency_plan.omega_from_lambda(1552e-9), beta2=-2.3e-28, beta3=4.1e-41, beta4=-3.0e-55)

    @settings(max_examples=300, deadline=None, derandomize=True)
    @given(l1=lam, l2=lam, l3=lam, orders=st.sampled_from([(2, 4), (2,), (4, 2)]), general=st.booleans())
    def check(l1, l2, l3, orders, general):
        cfg = PhaseMatchingConfig(method="general_taylor", max_order=4) if general else PhaseMatchingConfig(even_orders=orders)
        with np.errstate(all="ignore"):
            om, ok = frequency_plan.plan_from_wavelengths_batch(np.array([l1]), np.array([l2]), np.array([l3]))
            db, ok2 = phase_matching.compute_phase_mismatch_batch(om, d, cfg)
        try:
            om_s = frequency_plan.plan_from_wavelengths(l1, l2, l3)
            db_s = phase_matching.compute_phase_mismatch(om_s, d, cfg).delta_beta
            scalar_ok = bool(np.isfinite(db_s))
        except (ValueError, TypeError):
            scalar_ok = False
        assert bool(ok[0] and ok2[0]) == scalar_ok, (l1, l2, l3)
        if scalar_ok:
            assert np.array_equal(om[0], om_s)
            if not general and orders == (2,):
                assert db[0] == db_s
            else:
                # NumPy's array pow and the scalar pow may differ by an ulp of x**4 (even pow(-x, 4) vs pow(x, 4)): allow a
                # few ulp of the TERMS, which is all that is left when they cancel (e.g. lambda_signal == lambda_pump2)
                w = om_s
                oc = 0.5 * (w[0] + w[1])
                big = max(abs(w[2] - oc), abs(0.5 * (w[0] - w[1])), abs(w[3] - d.omega_ref), abs(w[0] - d.omega_ref))
                terms = abs(d.beta2) * big ** 2 + abs(d.beta3) * big ** 3 + abs(d.beta4) * big ** 4
                assert abs(db[0] - db_s) <= 1e-12 * abs(db_s) + 4e-15 * terms
        else:
            assert np.isnan(db[0])

    check()


def test_dbeta_only_driver_returns_the_general_taylor_mismatch(golden):
    """plot_dbeta_vs_lambda_signal: upstream (scan_mismtach.py:473-585) hands back NaN for every point; the
    quantity it describes is the reference's delta_beta_from_omegas, pinned here by G10's `gen4` column (rtol 1e-14: the
    array producer's pow may differ from the scalar API by an ulp) and by G9's impossible plan -> NaN."""
    from psa_amd.scan_mismtach import plot_dbeta_vs_lambda_signal
    g = golden("G10")
    dv = g["disp"]
    d = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3], extra={6: dv[4]})
    kw = dict(gamma=0.0115, p_in=[0.5, 0.5, 1e-5, 0.0], dispersion=d, show=False, show_progress=False)
    for i in range(len(g["l3"])):
        x, db = plot_dbeta_vs_lambda_signal(lambda_p1_m=g["lp1"][i], lambda_p2_m=g["lp2"][i], lambda_signal_m=[g["l3"][i]], **kw)
        assert x[0] == g["l3"][i] * 1e9
        np.testing.assert_allclose(db[0], g["gen4"][i], rtol=1e-14, atol=0)
    # a sweep: one impossible plan in the middle is NaN, the others are what the single-point calls give
    lam3 = np.array([1555e-9, 0.7e-6, 1560e-9])
    x, db = plot_dbeta_vs_lambda_signal(lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=lam3,
                                        return_wavelength_unit="m", **kw)
    assert np.array_equal(x, lam3) and list(np.isnan(db)) == [False, True, False]
    one = plot_dbeta_vs_lambda_signal(lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=[1560e-9], **kw)[1]
    assert db[2] == one[0]
    # the reference's argument checks
    for bad, msg in ((dict(lambda_signal_m=[]), "non-empty"), (dict(lambda_signal_m=[-1.0]), "finite positive"),
                     (dict(p_in=[1, 1, 1]), "shape"), (dict(p_in=[1, -1, 1, 1]), "non-negative"),
                     (dict(xscale="cubic"), "xscale"), (dict(return_wavelength_unit="um"), "return_wavelength_unit"),
                     (dict(yscale="log"), "strictly > 0")):
        args = dict(lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=lam3, **kw)
        args.update(bad)
        with pytest.raises(ValueError, match=msg):
            plot_dbeta_vs_lambda_signal(**args)
