#!/usr/bin/env python3
"""The three scenarios of the reference's main.py (:22-117 single run, :119-203 100-point gain spectrum,
:206-280 30-point gain + dbeta spectrum), run through psa_amd on the GPU.  Numbers only (no plotting).

    python examples/reference_scenarios.py            # needs an MI355X; prints gains and kernel times
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psa_amd  # noqa: E402,F401
from psa_amd.config import custom_simulation_config  # noqa: E402
from psa_amd.dispersion import delta_beta_from_omegas, delta_beta_symmetric, dispersion_params_from_D_S  # noqa: E402
from psa_amd.frequency_plan import describe_plan, infer_symmetry_from_omegas, lambda_from_omega, plan_from_wavelengths  # noqa: E402
from psa_amd.phase_matching import PhaseMatchingConfig, PhaseMatchingMethod  # noqa: E402
from psa_amd.scan_mismtach import plot_max_gain_and_dbeta_vs_lambda_signal, plot_max_signal_gain_vs_lambda_signal  # noqa: E402
from psa_amd.simulation import run_single_simulation  # noqa: E402

PM = PhaseMatchingConfig(method=PhaseMatchingMethod.SYMMETRIC_EVEN, even_orders=(2, 4), max_order=4)
GAMMA = 11.5 / 1000.0                                   # 1/(W m)
alpha_of = lambda db_per_km: (np.log(10.0) / 10.0) * db_per_km / 1000.0   # noqa: E731  (1/m)


def dispersion_at_pump_centre(lp1, lp2, l3, D):
    om = plan_from_wavelengths(lp1, lp2, l3)
    sp = infer_symmetry_from_omegas(*om)
    return om, sp, dispersion_params_from_D_S(lambda_from_omega(sp.omega_c), D, 0.02, 0, D_units="ps/nm/km",
                                              S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km", omega_ref=sp.omega_c)


def single_simulation():
    om, sp, disp = dispersion_at_pump_centre(1550e-9, 1560e-9, 1555e-9, 0.02)
    print(describe_plan(om))
    p_in = np.array([0.5, 0.5, 1e-5, 1e-5])
    for _ in range(2):          # the first call of a process also loads the library and the HIP module (~0.2 s): time the second
        t = time.perf_counter()
        z, A = run_single_simulation(custom_simulation_config(z_max=1000.0, dz=0.1), gamma=GAMMA, alpha=alpha_of(0.9), omega=om,
                                     p_in=p_in, phase_in=np.zeros(4), dispersion=disp, phase_matching_cfg=PM)
        dt = time.perf_counter() - t
    P_out = np.abs(A[-1]) ** 2
    print(f"z_end = {z[-1]:.3f} m, rows = {len(z)}, P_out = {P_out}")
    print(f"signal gain = {10 * np.log10(P_out[2] / p_in[2]):.6f} dB   (reference: 45.292444 dB)")
    print(f"dbeta = {delta_beta_from_omegas(om, disp):.6e} 1/m, dbeta_sym = "
          f"{delta_beta_symmetric(sp.omega_c, sp.omega_d, sp.Omega, disp):.6e} 1/m, gamma(P1+P2) = {GAMMA:.4f} 1/m")
    print(f"wall {dt * 1e3:.1f} ms for 10 000 RK4 steps (the reference takes ~0.8 s)\n")


def gain_spectrum():
    lam3 = np.linspace(1540e-9, 1650e-9, 100)
    _, _, disp = dispersion_at_pump_centre(1550e-9, 1555e-9, float(lam3[0]), 0.2)
    t = time.perf_counter()
    x, g = plot_max_signal_gain_vs_lambda_signal(cfg=custom_simulation_config(z_max=500.0, dz=0.2), lambda_p1_m=1550e-9,
                                                 lambda_p2_m=1555e-9, lambda_signal_m=lam3, gamma=GAMMA, alpha=alpha_of(0.5),
                                                 p_in=[0.5, 0.5, 1e-7, 1e-7], phase_in=np.zeros(4), dispersion=disp,
                                                 phase_matching_cfg=PM, gain_unit="db", show=False)
    print(f"100-point gain spectrum: peak {np.nanmax(g):.4f} dB at {x[np.nanargmax(g)]:.2f} nm (reference 45.4894 dB), "
          f"{np.isnan(g).sum()} NaN, wall {(time.perf_counter() - t) * 1e3:.1f} ms (the reference takes ~21 s)\n")


def gain_and_dbeta_spectrum():
    lam3 = np.linspace(1540e-9, 1565e-9, 30)
    _, _, disp = dispersion_at_pump_centre(1550e-9, 1558e-9, float(lam3[0]), 0.1)
    t = time.perf_counter()
    x, g, db = plot_max_gain_and_dbeta_vs_lambda_signal(cfg=custom_simulation_config(z_max=500.0, dz=0.2),
                                                        lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=lam3,
                                                        gamma=GAMMA, alpha=alpha_of(0.5), p_in=[0.1, 0.1, 1e-7, 1e-7],
                                                        dispersion=disp, gain_unit="dB", phase_in=np.zeros(4), show=False)
    print(f"30-point gain + dbeta spectrum: peak {np.nanmax(g):.4f} dB at {x[np.nanargmax(g)]:.2f} nm (reference 7.6894 dB); "
          f"dbeta in [{db.min():.4e}, {db.max():.4e}] 1/m; wall {(time.perf_counter() - t) * 1e3:.1f} ms (reference ~6.6 s)")


if __name__ == "__main__":
    single_simulation()
    gain_spectrum()
    gain_and_dbeta_spectrum()
