"""Invalid calls of the hot-path API, written once and evaluated twice: by gen_golden.py against the reference (result stored
in error_contract.json: exception type + message per case) and by tests/test_host_logic.py against this package.  Every
case must fail in argument validation, i.e. before any device work, so the test runs without a GPU.

``build(imp)``: imp(module_name) -> module ("config", "simulation", ...) of whichever implementation is under test.
"""
import numpy as np


def build(imp):
    config, integrators, simulation, sm, fp, dp, pm = (imp(m) for m in ("config", "integrators", "simulation", "scan_mismtach",
                                                                        "frequency_plan", "dispersion", "phase_matching"))
    cfg = config.custom_simulation_config(z_max=10.0, dz=0.1)
    disp = dp.DispersionParams(omega_ref=1.2e15, beta2=-2e-28)
    om = fp.plan_from_wavelengths(1550e-9, 1558e-9, 1554e-9)
    ok_run = dict(gamma=0.01, alpha=0.0, omega=om, p_in=[0.1, 0.1, 1e-6, 0.0], dispersion=disp)
    ok_drv = dict(cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=[1554e-9], gamma=0.01, alpha=0.0,
                  p_in=[0.1, 0.1, 1e-6, 0.0], phase_in=None, dispersion=disp, show=False, show_progress=False)

    def f_lin(z, y, p):
        return -y

    return {
        "config.z_max<=0": lambda: config.custom_simulation_config(z_max=0.0, dz=0.1),
        "config.dz<=0": lambda: config.custom_simulation_config(z_max=1.0, dz=-0.1),
        "config.dz>z_max": lambda: config.custom_simulation_config(z_max=1.0, dz=2.0),
        "config.save_every": lambda: config.custom_simulation_config(z_max=1.0, dz=0.1, save_every=0),
        "interval.z_max": lambda: integrators.integrate_interval(f_lin, 0.0, 0.1, np.ones(1), None),
        "interval.dz": lambda: integrators.integrate_interval(f_lin, 1.0, 0.0, np.ones(1), None),
        "interval.save_every": lambda: integrators.integrate_interval(f_lin, 1.0, 0.1, np.ones(1), None, save_every=0),
        "fixed.grid_ndim": lambda: integrators.integrate_fixed_step(f_lin, np.zeros((2, 2)), np.ones(1), None),
        "fixed.grid_short": lambda: integrators.integrate_fixed_step(f_lin, np.zeros(1), np.ones(1), None),
        "run.omega_shape": lambda: simulation.run_single_simulation(cfg, **{**ok_run, "omega": [1e15, 1e15]}),
        "run.omega_nonpos": lambda: simulation.run_single_simulation(cfg, **{**ok_run, "omega": [1e15, 1e15, -1e15, 1e15]}),
        "run.p_shape": lambda: simulation.run_single_simulation(cfg, **{**ok_run, "p_in": [0.1, 0.1]}),
        "run.p_negative": lambda: simulation.run_single_simulation(cfg, **{**ok_run, "p_in": [0.1, -0.1, 0.0, 0.0]}),
        "run.phase_shape": lambda: simulation.run_single_simulation(cfg, phase_in=[0.0, 1.0], **ok_run),
        "run.length_unit": lambda: simulation.run_single_simulation(cfg, length_unit="mile", **ok_run),
        "run.no_dispersion": lambda: simulation.run_single_simulation(cfg, **{**ok_run, "dispersion": None}),
        "run.dispersion_type": lambda: simulation.run_single_simulation(cfg, **{**ok_run, "dispersion": {"beta2": 1}}),
        "run.legacy_shape": lambda: simulation.run_single_simulation(cfg, beta_legacy=[1.0, 2.0], **{**ok_run, "dispersion": None}),
        "run.pm_type": lambda: simulation.run_single_simulation(cfg, phase_matching_cfg="symmetric", **ok_run),
        "drv.p_in_shape": lambda: sm.plot_max_gain_and_dbeta_vs_lambda_signal(**{**ok_drv, "p_in": [0.1, 0.1]}),
        "drv.seed_zero": lambda: sm.plot_max_gain_and_dbeta_vs_lambda_signal(**{**ok_drv, "p_in": [0.1, 0.1, 0.0, 0.0]}),
        "drv.lambda_empty": lambda: sm.plot_max_gain_and_dbeta_vs_lambda_signal(**{**ok_drv, "lambda_signal_m": []}),
        "drv.lambda_neg": lambda: sm.plot_max_gain_and_dbeta_vs_lambda_signal(**{**ok_drv, "lambda_signal_m": [-1e-6]}),
        "drv.gain_unit": lambda: sm.plot_max_gain_and_dbeta_vs_lambda_signal(gain_unit="neper", **ok_drv),
        "drv.wavelength_unit": lambda: sm.plot_max_gain_and_dbeta_vs_lambda_signal(return_wavelength_unit="um", **ok_drv),
        "drv.no_dispersion": lambda: sm.plot_max_gain_and_dbeta_vs_lambda_signal(**{**ok_drv, "dispersion": None}),
        "drv.log_db": lambda: sm.plot_max_gain_and_dbeta_vs_lambda_signal(yscale_gain="log", gain_unit="dB", **ok_drv),
        "drv1.gain_unit": lambda: sm.plot_max_signal_gain_vs_lambda_signal(gain_unit="neper", **ok_drv),
        "drv1.phase_shape": lambda: sm.plot_max_signal_gain_vs_lambda_signal(**{**ok_drv, "phase_in": [0.0]}),
        "drv1.wavelength_unit": lambda: sm.plot_max_signal_gain_vs_lambda_signal(return_wavelength_unit="um", **ok_drv),
        "plan.lambda_nonpos": lambda: fp.plan_from_wavelengths(1550e-9, 0.0, 1554e-9),
        "plan.no_idler": lambda: fp.plan_from_wavelengths(1550e-9, 1558e-9, 0.7e-6),
        "plan.energy": lambda: fp.plan_from_omegas(1e15, 1e15, 1e15, 2e15),
        "sym.od_ge_oc": lambda: fp.SymmetricPlan(omega_c=1e15, omega_d=2e15, Omega=1e12),
        "disp.omega_ref": lambda: dp.DispersionParams(omega_ref=-1.0),
        "disp.extra_key": lambda: dp.DispersionParams(omega_ref=1e15, extra={"two": 1.0}),
        "disp.units": lambda: dp.dispersion_params_from_D_S(1550e-9, 1.0, D_units="ps/km"),
        "dbeta.even_odd": lambda: dp.delta_beta_symmetric(1e15, 1e12, 1e13, disp, even_orders=(3,)),
        "dbeta.even_empty": lambda: dp.delta_beta_symmetric(1e15, 1e12, 1e13, disp, even_orders=()),
        "pm.method": lambda: pm.PhaseMatchingConfig(method="magic"),
        "pm.provided_missing": lambda: pm.PhaseMatchingConfig(method="provided"),
        "pm.max_order": lambda: pm.PhaseMatchingConfig(max_order=-1),
        "pm.rtol": lambda: pm.PhaseMatchingConfig(rtol=-1.0),
        "pm.no_disp": lambda: pm.compute_phase_mismatch(om, None, pm.PhaseMatchingConfig()),
    }


def evaluate(imp):
    out = {}
    for name, fn in build(imp).items():
        try:
            fn()
            out[name] = ["NO EXCEPTION", ""]
        except Exception as e:   # noqa: BLE001 -- the point is to record whatever is raised
            out[name] = [type(e).__name__, str(e)]
    return out
