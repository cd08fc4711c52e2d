"""A seeded random walk over the scalar host API (frequency plans, dispersion builders, Taylor / symmetric phase mismatch,
every phase-matching method, configuration objects), written once and evaluated twice: by gen_golden.py against the
reference (stored in host_api_records.json) and by tests/test_host_logic.py against this package.  Records are plain JSON
(floats survive the round trip exactly); a call that raises is recorded as ["EXC", <exception type>].

``evaluate(imp)``: imp(module_name) -> module of whichever implementation is under test.
"""
import numpy as np


def evaluate(imp):
    fp, dp, pm, config = (imp(m) for m in ("frequency_plan", "dispersion", "phase_matching", "config"))
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
    d0 = config.default_simulation_config()
    rec("default_config", [d0.z_max, d0.dz, d0.save_every, d0.check_nan])
    return out
