#!/usr/bin/env python3
"""A/B of the two float64 lane layouts around the points where the automatic choice flips: kernel time
(hipEvents inside psa_rk4_sweep_f64) for one lane per point, two lanes per point and the library's own choice, next to the
cost model's prediction (csrc/psa_rk4_f64.hip: T ~ I * k / eff(k), k = waves per SIMD).
    python tools/split_cliff.py [n_steps]   -> table on stdout (kept as profiles/r03_split_cliff.log)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psa_amd._native as nat  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
SIMDS = 1024
I = {4: (300.7, 183.6), 6: (471.9, 296.8)}
eff = lambda k: 0.896 if k <= 1 else (0.94 if k == 2 else 0.96)  # noqa: E731


def model(nw, n):
    i1, i2 = I[nw]
    k1, k2 = -(-(-(-n // 64)) // SIMDS), -(-(-(-2 * n // 64)) // SIMDS)
    return i1 * k1 / eff(k1), i2 * k2 / eff(k2)


def run(nw, n, flags):
    db = np.linspace(-0.05, 0.05, n)
    p = [0.5, 0.5, 1e-5, 1e-5] if nw == 4 else [0.3, 0.25, 1e-6, 1e-6, 2e-6, 5e-7]
    kw = dict(n_steps=n_steps, z_max=n_steps * 0.01, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=np.sqrt(p).astype(complex),
              dbeta2=(0.5 * db[::-1] if nw == 6 else None), extra_flags=flags)
    nat.sweep_host(db, **kw)
    return min(nat.sweep_host(db, **kw)["elapsed_ms"] for _ in range(3))


print(f"# {n_steps} z-steps, float64, kernel ms (best of 3); model = relative cost, lower wins; auto = the library's choice")
print(f"{'waves':>5} {'N':>8} {'one lane':>9} {'two lanes':>9} {'auto':>8} {'auto picks':>10} {'model 1':>8} {'model 2':>8} {'model picks':>11} {'best':>9} {'four lanes':>10}")
for nw in (4, 6):
    for n in (1, 100, 4_096, 16_384, 16_385, 32_768, 32_769, 40_000, 49_152, 65_536, 65_537, 70_000, 81_920, 98_304, 98_305, 114_688, 131_072, 163_840, 262_144):
        t1, t2, ta = run(nw, n, nat.OPT_ONE_LANE), run(nw, n, nat.OPT_SPLIT_POINT), run(nw, n, 0)
        m1, m2 = model(nw, n)
        t4 = run(nw, n, nat.OPT_QUAD_POINT) if (nw == 4 and n <= 32_768) else float("nan")
        cand = {"one": t1, "two": t2, **({"four": t4} if t4 == t4 else {})}
        picks = min(cand, key=lambda k: abs(ta - cand[k]))
        print(f"{nw:>5} {n:>8} {t1:>9.2f} {t2:>9.2f} {ta:>8.2f} {picks:>10} {m1:>8.0f} {m2:>8.0f} {'two' if m2 < m1 else 'one':>11} "
              f"{min(cand, key=cand.get):>9} {t4:>10.2f}", flush=True)
