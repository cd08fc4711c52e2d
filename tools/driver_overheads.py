#!/usr/bin/env python3
"""Developer probe: user-visible wall time of the sweep drivers against the kernel time inside them, for sweeps short enough
that host work shows (the kernel is 6-90 ms here)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psa_amd import config, dispersion, scan_mismtach
import psa_amd._native as nat
d = dispersion.dispersion_params_from_D_S(1554e-9, 0.1, 0.02, 0.0, D_units="ps/nm/km", S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km")
cfg = config.custom_simulation_config(z_max=1000.0, dz=0.1)       # 10 000 steps
p_in = [0.5, 0.5, 1e-5, 1e-5]
scan_mismtach.scan_dbeta_seeded_signal(cfg=config.custom_simulation_config(z_max=1.0, dz=0.1), delta_beta=np.zeros(8), gamma=0.0115, alpha=0.0, p_in=p_in)


def best(fn, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, r


for N in (4096, 65536, 1 << 20):
    db = np.linspace(-0.05, 0.05, N)
    w, out = best(lambda: scan_mismtach.scan_dbeta_seeded_signal(cfg=cfg, delta_beta=db, gamma=0.0115, alpha=1.15e-4, p_in=p_in, gain_mode="max"))
    k = out["result"].elapsed_ms
    w2, r = best(lambda: nat.sweep_host(db, n_steps=10000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=np.sqrt(p_in).astype(complex)))
    print(f"scan_dbeta_seeded_signal N={N}: wall {w:.1f} ms, kernel {k:.1f} ms; bare sweep_host wall {w2:.1f} ms (kernel {r['elapsed_ms']:.1f})", flush=True)
for N in (4096, 65536):
    lam3 = np.linspace(1535e-9, 1570e-9, N)
    w, out = best(lambda: scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, lambda_signal_m=lam3,
                  gamma=0.0115, alpha=1.15e-4, p_in=[0.1, 0.1, 1e-7, 1e-7], phase_in=None, dispersion=d, show=False, show_progress=False), n=3)
    print(f"plot_max_gain_and_dbeta_vs_lambda_signal N={N}: wall {w:.1f} ms", flush=True)
