#!/usr/bin/env python3
"""What the per-row work of the summary-mode z-loop costs: the config-2 sweep (65 536 points x 1e5 steps, float64) at
save_every = 10 (the headline), 100, 1 000 and n_steps, kernel ms (hipEvents inside psa_rk4_sweep_f64), best of 3.
The rows differ only in how often |A_sig|^2 / the finite test / the replay checkpoint run and the event loop is re-entered.
    python tools/save_every_cost.py [n_points] [n_steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psa_amd._native as nat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65_536
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
db = np.linspace(-0.05, 0.05, n)
a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
for exact in (True, False):
    for se in (10, 20, 64, 100, 1_000, n_steps):
        kw = dict(n_steps=n_steps, z_max=n_steps * 0.01, save_every=se, gamma=0.0115, alpha=1.15e-4, a0=a0, exact_step=exact)
        nat.sweep_host(db, **kw)
        t = min(nat.sweep_host(db, **kw)["elapsed_ms"] for _ in range(3))
        print(f"N={n} n={n_steps} save_every={se:>6} {'exact (replay)' if exact else 'block check   '}: {t:8.3f} ms", flush=True)
