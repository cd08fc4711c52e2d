#!/usr/bin/env python3
"""Developer probe: float32 (packed) vs float64 kernel on the same inputs, error vs number of z-steps."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psa_amd._native as nat
rng = np.random.default_rng(4)
N = 512
for (P, name) in (((0.5, 0.5, 1e-5, 1e-5), "45 dB-class pumps 0.5 W"), ((0.1, 0.1, 1e-7, 1e-7), "C3/C4 pumps 0.1 W")):
    a0 = np.sqrt(np.array(P)).astype(complex)
    db = np.linspace(-0.02, 0.02, N)
    for n in (10_000, 100_000, 1_000_000):
        kw = dict(n_steps=n, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
        r64 = nat.sweep_host(db, **kw)
        r32 = nat.sweep_host(db.astype(np.float32), dtype=np.float32, **kw)
        ea = np.abs(r32["a_end"].astype(complex) - r64["a_end"]) / np.abs(r64["a_end"])
        g64 = 10 * np.log10(r64["p_max"] / P[2]); g32 = 10 * np.log10(r32["p_max"].astype(float) / P[2])
        print(f"{name}: n={n:>8}: a_end rel err max {ea.max():.2e} median {np.median(ea):.2e}; gain dB err max {np.abs(g32-g64).max():.2e} "
              f"(max gain {g64.max():.1f} dB); f32 kernel {r32['elapsed_ms']:.1f} ms, f64 {r64['elapsed_ms']:.1f} ms", flush=True)
