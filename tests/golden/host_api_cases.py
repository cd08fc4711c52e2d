"""A seeded random walk over the scalar host API (frequency plans, dispersion builders, Taylor / symmetric phase mismatch,
every phase-matching method, configuration objects), written once and evaluated twice: by gen_golden.py against the
reference (stored in host_api_records.json) and by tests/test_host_logic.py against this package.  Records are plain JSON
(floats survive the round trip exactly); a call that raises is recorded as ["EXC", <exception type>].

``evaluate(imp)``: imp(module_name) -> module of whichever implementation is under test.
"""
import numpy as np


def evaluate(imp):
    import warnings
    with warnings.catch_warnings():
        # expected on both sides: complex increments stored into a real state array (the stepper keeps y0's dtype), overflow
        # in the deliberate blow-up
        warnings.simplefilter("ignore")
        return _evaluate(imp)


def _evaluate(imp):
    fp, dp, pm, config, prm, integ = (imp(m) for m in ("frequency_plan", "dispersion", "phase_matching", "config", "parameters",
                                                      "integrators"))
    rng = np.random.default_rng(99)
    out = {}

    def rec(k, v):
        out[k] = v.tolist() if isinstance(v, np.ndarray) else v

    def tryrec(k, fn):
        try:
            rec(k, fn())
        except Exception as e:   # noqa: BLE001 -- record whatever is raised
            rec(k, ["EXC", type(e).__name__])

    for i in range(300):
        l1, l2 = rng.uniform(1.50e-6, 1.60e-6, 2)
        l3 = rng.uniform(1.40e-6, 1.70e-6) if i % 7 else rng.uniform(0.6e-6, 0.9e-6)   # every 7th plan has no idler
        f0 = rng.uniform(1e14, 3e14)
        tryrec(f"omega_from_lambda{i}", lambda: fp.omega_from_lambda(l1))
        tryrec(f"lambda_from_omega{i}", lambda: fp.lambda_from_omega(fp.omega_from_lambda(l2)))
        tryrec(f"f_from_omega{i}", lambda: fp.f_from_omega(fp.omega_from_f(f0)))
        tryrec(f"plan{i}", lambda: fp.plan_from_wavelengths(l1, l2, l3))

        def symmetric():
            om = fp.plan_from_wavelengths(l1, l2, l3)
            sp = fp.infer_symmetry_from_omegas(*om)
            return [sp.omega_c, sp.omega_d, sp.Omega, list(fp.plan_from_symmetry(sp.omega_c, sp.omega_d, sp.Omega)),
                    fp.describe_plan(om)]
        tryrec(f"symmetric{i}", symmetric)
        D, S = rng.uniform(-2, 2), rng.uniform(0, 0.1)
        dS = [0.0, rng.uniform(-1e-3, 1e-3)][i % 2]

        def dispersion():
            d = dp.dispersion_params_from_D_S(l1, D, S, dS, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                              dSdlmbd_units="ps/nm^3/km")
            om = fp.plan_from_wavelengths(l1, l2, l3)
            sp = fp.infer_symmetry_from_omegas(*om)
            res = [d.omega_ref, d.beta2, d.beta3, d.beta4, list(d.available_orders()),
                   dp.beta_taylor(float(om[2]), d), dp.beta_taylor(float(om[3]), d, max_order=3),
                   dp.delta_beta_from_omegas(om, d), dp.delta_beta_from_omegas(om, d, max_order=2),
                   dp.delta_beta_symmetric(sp.omega_c, sp.omega_d, sp.Omega, d),
                   dp.delta_beta_symmetric(sp.omega_c, sp.omega_d, sp.Omega, d, even_orders=(4, 2))]
            for kw in (dict(), dict(method="general_taylor", max_order=3), dict(method="symmetric_even", even_orders=(2,)),
                       dict(method="provided", provided_delta_beta=0.25)):
                r = pm.compute_phase_mismatch(om, d, pm.PhaseMatchingConfig(**kw))
                res.append(r.delta_beta)
                res.append(None if r.symmetric is None else [r.symmetric.omega_c, r.symmetric.omega_d, r.symmetric.Omega])
            res.append(pm.PhaseMismatchCalculator(d, pm.PhaseMatchingConfig())(om).delta_beta)
            return res
        tryrec(f"dispersion{i}", dispersion)
    for i in range(40):
        zmax, dz, se = rng.uniform(0.1, 100), rng.uniform(1e-3, 1.0), int(rng.integers(1, 20))

        def cfg():
            c = config.custom_simulation_config(z_max=zmax, dz=dz, save_every=se)
            config.validate_config(c)
            return [c.z_max, c.dz, c.save_every, c.check_nan]
        tryrec(f"config{i}", cfg)
    # the parameter carriers (parameters.py): valid and invalid constructions, the cache setter, the default factories
    disp = dp.DispersionParams(omega_ref=1.2e15, beta2=-2e-28, beta4=1e-55)
    om_ok = fp.plan_from_wavelengths(1550e-9, 1558e-9, 1554e-9)
    carriers = {
        "fiber.ok": lambda: prm.FiberParams(length_m=100.0, gamma_W_m=0.01, alpha_1_m=1e-4, dispersion=disp, beta_legacy_1_m=None),
        "fiber.legacy": lambda: prm.FiberParams(length_m=100.0, gamma_W_m=0.01, alpha_1_m=0.0, dispersion=None,
                                               beta_legacy_1_m=np.array([1.0, 2.0, 3.0, 4.0])),
        "fiber.length<=0": lambda: prm.FiberParams(length_m=0.0, gamma_W_m=0.01, alpha_1_m=0.0, dispersion=None, beta_legacy_1_m=None),
        "fiber.alpha<0": lambda: prm.FiberParams(length_m=1.0, gamma_W_m=0.01, alpha_1_m=-1.0, dispersion=None, beta_legacy_1_m=None),
        "fiber.gamma_nan": lambda: prm.FiberParams(length_m=1.0, gamma_W_m=float("nan"), alpha_1_m=0.0, dispersion=None, beta_legacy_1_m=None),
        "fiber.disp_type": lambda: prm.FiberParams(length_m=1.0, gamma_W_m=0.01, alpha_1_m=0.0, dispersion={"b": 1}, beta_legacy_1_m=None),
        "fiber.legacy_shape": lambda: prm.FiberParams(length_m=1.0, gamma_W_m=0.01, alpha_1_m=0.0, dispersion=None, beta_legacy_1_m=[1.0, 2.0]),
        "waves.ok": lambda: prm.WavesParams(omega=om_ok, symmetric=None),
        "waves.shape": lambda: prm.WavesParams(omega=[1e15, 1e15], symmetric=None),
        "waves.nonpos": lambda: prm.WavesParams(omega=[1e15, 1e15, 0.0, 1e15], symmetric=None),
        "waves.sym_type": lambda: prm.WavesParams(omega=om_ok, symmetric=(1.0, 2.0, 3.0)),
        "grid.ok": lambda: prm.SimulationGrid(dz_m=0.1, z0_m=0.0),
        "grid.dz<=0": lambda: prm.SimulationGrid(dz_m=0.0, z0_m=0.0),
        "grid.z0_nan": lambda: prm.SimulationGrid(dz_m=0.1, z0_m=float("nan")),
        "pmparams.type": lambda: prm.PhaseMatchingParams(config="symmetric_even"),
        "pmparams.default": lambda: prm.make_default_phase_matching_params(),
        "pmparams.general": lambda: prm.make_default_phase_matching_params(method=pm.PhaseMatchingMethod.GENERAL_TAYLOR),
        "model.cache_type": lambda: prm.ModelParams(waves=prm.WavesParams(omega=om_ok, symmetric=None),
                                                    fiber=prm.FiberParams(length_m=1.0, gamma_W_m=0.01, alpha_1_m=0.0, dispersion=None, beta_legacy_1_m=None),
                                                    grid=prm.SimulationGrid(dz_m=0.1, z0_m=0.0),
                                                    phase_matching=prm.make_default_phase_matching_params(), cache={}),
    }

    def show(obj):
        """A JSON view of a carrier: its field values (arrays as lists, nested carriers recursively, enums by value)."""
        import dataclasses
        import enum
        if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
            return {f.name: show(getattr(obj, f.name)) for f in dataclasses.fields(obj)}
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        if isinstance(obj, enum.Enum):
            return obj.value
        if isinstance(obj, (tuple, list)):
            return [show(x) for x in obj]
        if isinstance(obj, dict):
            return {str(k): show(v) for k, v in obj.items()}
        return obj

    for name, fn in carriers.items():
        tryrec("carrier." + name, lambda fn=fn: show(fn()))

    def model_and_cache():
        mp = prm.make_model_params(waves=prm.WavesParams(omega=om_ok, symmetric=None),
                                   fiber=prm.FiberParams(length_m=50.0, gamma_W_m=0.01, alpha_1_m=0.0, dispersion=disp, beta_legacy_1_m=None),
                                   grid=prm.SimulationGrid(dz_m=0.1, z0_m=0.0))
        before = show(mp)
        mp.cache.set_phase_mismatch(0.125)
        res = [before, mp.cache.delta_beta_1_m]
        try:
            mp.cache.set_phase_mismatch(float("nan"))
            res.append("accepted")
        except Exception as e:   # noqa: BLE001
            res.append(["EXC", type(e).__name__])
        return res
    tryrec("carrier.model_and_cache", model_and_cache)
    # the generic stepper with arbitrary Python callables (integrators.py:25-204): real and complex linear systems with a
    # z-dependent coefficient, random grids and save strides; complex results are recorded as interleaved (re, im) floats
    def flat(y):
        y = np.asarray(y)
        return y.view(np.float64).tolist() if np.iscomplexobj(y) else y.tolist()

    for i in range(24):
        dim = int(rng.integers(1, 6))
        M = rng.normal(size=(dim, dim)) + 1j * rng.normal(size=(dim, dim)) * (i % 2)

        def f(z, y, p, M=M):
            return M @ y * np.cos(z) - 0.1 * y
        y0 = rng.normal(size=dim) + (1j * rng.normal(size=dim) if i % 2 else 0)
        zmax, dz, se = float(rng.uniform(0.1, 3.0)), float(rng.uniform(0.01, 0.3)), int(rng.integers(1, 6))
        z_step = float(rng.uniform(0, 1))
        zg = np.sort(rng.uniform(0, 2, int(rng.integers(2, 12))))

        def interval():
            z, y = integ.integrate_interval(f, zmax, dz, y0, None, save_every=se, check_nan=bool(i % 3))
            return [z.tolist(), flat(y), str(y.dtype), list(y.shape)]

        def fixed():
            z, y = integ.integrate_fixed_step(f, zg, y0, None, save_every=se)
            return [z.tolist(), flat(y)]
        tryrec(f"stepper.interval{i}", interval)
        tryrec(f"stepper.rk4_step{i}", lambda: flat(integ.rk4_step(f, z_step, y0, dz, None)))
        tryrec(f"stepper.fixed{i}", fixed)

    def blow_up():
        try:
            with np.errstate(all="ignore"):
                integ.integrate_interval(lambda z, y, p: y * y * 1e6, 10.0, 0.5, np.array([10.0]), None)
            return "no exception"
        except FloatingPointError as e:
            return ["FloatingPointError", str(e)]
    tryrec("stepper.blow_up", blow_up)
    d0 = config.default_simulation_config()
    rec("default_config", [d0.z_max, d0.dz, d0.save_every, d0.check_nan])
    return out
