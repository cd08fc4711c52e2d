#!/usr/bin/env python3
"""The reference's own scenarios are SMALL sweeps (main.py: 1 point x 10 000 steps, 100 points x 2 500, 30 points x 2 500):
user-visible wall time of the driver calls against the kernel time inside them and the reference's measured wall
(SURVEY 8d / BASELINE.md: 0.825 s, 21.1 s, 6.56 s on one core).  Steady state (the first call of a process also loads the
HIP module).  Kept as profiles/r03_small_sweeps.log."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psa_amd._native as nat  # noqa: E402
from psa_amd.config import custom_simulation_config  # noqa: E402
from psa_amd.dispersion import dispersion_params_from_D_S  # noqa: E402
from psa_amd.frequency_plan import infer_symmetry_from_omegas, lambda_from_omega, plan_from_wavelengths  # noqa: E402
from psa_amd.phase_matching import PhaseMatchingConfig  # noqa: E402
from psa_amd.scan_mismtach import plot_max_gain_and_dbeta_vs_lambda_signal, plot_max_signal_gain_vs_lambda_signal  # noqa: E402
from psa_amd.simulation import run_single_simulation  # noqa: E402

GAMMA = 0.0115
alpha_of = lambda db_per_km: (np.log(10.0) / 10.0) * db_per_km / 1000.0  # noqa: E731


def disp_for(lp1, lp2, l3, D):
    om = plan_from_wavelengths(lp1, lp2, l3)
    sp = infer_symmetry_from_omegas(*om)
    return om, dispersion_params_from_D_S(lambda_from_omega(sp.omega_c), D, 0.02, 0, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                          dSdlmbd_units="ps/nm^3/km", omega_ref=sp.omega_c)


def timed(fn, n=20):
    fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3, float(min(ts)) * 1e3, r


def kernel_ms(N, n_steps, z_max, se, traj, **kw):
    db = np.linspace(-0.01, 0.01, N) if N > 1 else [3.9e-4]
    a0 = np.sqrt(kw.pop("p")).astype(complex)
    r = [nat.sweep_host(db, n_steps=n_steps, z_max=z_max, save_every=se, gamma=GAMMA, a0=a0, want_traj=traj, exact_step=traj, **kw)
         for _ in range(5)]
    return min(x["elapsed_ms"] for x in r)


om, d1 = disp_for(1550e-9, 1560e-9, 1555e-9, 0.02)
s1 = lambda: run_single_simulation(custom_simulation_config(z_max=1000.0, dz=0.1), gamma=GAMMA, alpha=alpha_of(0.9), omega=om,  # noqa: E731
                                   p_in=[0.5, 0.5, 1e-5, 1e-5], phase_in=np.zeros(4), dispersion=d1,
                                   phase_matching_cfg=PhaseMatchingConfig())
lam100 = np.linspace(1540e-9, 1650e-9, 100)
_, d2 = disp_for(1550e-9, 1555e-9, float(lam100[0]), 0.2)
s2 = lambda: plot_max_signal_gain_vs_lambda_signal(cfg=custom_simulation_config(z_max=500.0, dz=0.2), lambda_p1_m=1550e-9,  # noqa: E731
                                                   lambda_p2_m=1555e-9, lambda_signal_m=lam100, gamma=GAMMA, alpha=alpha_of(0.5),
                                                   p_in=[0.5, 0.5, 1e-7, 1e-7], phase_in=np.zeros(4), dispersion=d2,
                                                   phase_matching_cfg=PhaseMatchingConfig(), gain_unit="db", show=False)
lam30 = np.linspace(1540e-9, 1565e-9, 30)
_, d3 = disp_for(1550e-9, 1558e-9, float(lam30[0]), 0.1)
s3 = lambda: plot_max_gain_and_dbeta_vs_lambda_signal(cfg=custom_simulation_config(z_max=500.0, dz=0.2), lambda_p1_m=1550e-9,  # noqa: E731
                                                      lambda_p2_m=1558e-9, lambda_signal_m=lam30, gamma=GAMMA, alpha=alpha_of(0.5),
                                                      p_in=[0.1, 0.1, 1e-7, 1e-7], dispersion=d3, gain_unit="dB",
                                                      phase_in=np.zeros(4), show=False)
t0 = time.perf_counter()
s1()
cold = (time.perf_counter() - t0) * 1e3
print(f"# {nat.version()}; first call of the process (library + HIP module load included): {cold:.0f} ms")
print(f"{'scenario (main.py)':46s} {'wall median':>11} {'wall min':>9} {'kernel':>8} {'host share':>10} {'reference':>10} {'speed-up':>9}")
rows = [("G1 single run, 1 x 10 000 steps, 1 001 rows out", s1, dict(N=1, n_steps=10_000, z_max=1000.0, se=10, traj=True, alpha=alpha_of(0.9), p=[0.5, 0.5, 1e-5, 1e-5]), 825.0),
        ("G3 gain spectrum, 100 x 2 500 steps", s2, dict(N=100, n_steps=2500, z_max=500.0, se=10, traj=False, alpha=alpha_of(0.5), p=[0.5, 0.5, 1e-7, 1e-7]), 21_100.0),
        ("G2 gain + dbeta spectrum, 30 x 2 500 steps", s3, dict(N=30, n_steps=2500, z_max=500.0, se=10, traj=False, alpha=alpha_of(0.5), p=[0.1, 0.1, 1e-7, 1e-7]), 6_560.0)]
for name, fn, kk, ref_ms in rows:
    med, mn, _ = timed(fn)
    k = kernel_ms(**kk)
    print(f"{name:46s} {med:>9.3f}ms {mn:>7.3f}ms {k:>6.3f}ms {100 * (med - k) / med:>9.1f}% {ref_ms / 1e3:>8.3f} s {ref_ms / med:>8.0f}x", flush=True)
# what a host-buffer call costs besides its kernel
for N, n_steps in ((1, 10), (1, 10_000), (30, 2500), (100, 2500), (4096, 2500)):
    db = np.linspace(-0.01, 0.01, N)
    kw = dict(n_steps=n_steps, z_max=n_steps * 0.1, save_every=10, gamma=GAMMA, alpha=1e-4, a0=np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex))
    med, mn, r = timed(lambda: nat.sweep_host(db, **kw), n=30)
    g_med, _, _ = timed(lambda: nat.gain_summary_host(r["p_max"], r["first_bad_step"], 1e-5), n=30)
    print(f"psa_rk4_sweep_f64 N={N:<5d} n={n_steps:<6d}: wall {med:.3f} ms, kernel {r['elapsed_ms']:.3f} ms -> fixed cost {med - r['elapsed_ms']:.3f} ms;"
          f" psa_gain_summary_f64 wall {g_med:.3f} ms", flush=True)
