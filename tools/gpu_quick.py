#!/usr/bin/env python3
"""Developer probe (not part of the product or the test-suite): parity of the HIP sweep against the C oracle on
small seeded inputs, then kernel timings for a few launch shapes.  Usage: python tools/gpu_quick.py [--big]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import psa_amd._native as nat  # noqa: E402
import oracle as O  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def parity():
    a0 = np.sqrt(np.array([0.5, 0.5, 1e-5, 1e-5])).astype(complex)
    db = np.linspace(-0.05, 0.05, 257)
    for (n, se, alpha) in [(10_000, 10, 1.15e-4), (10_000, 10, 0.0), (1005, 10, 1.15e-4), (1005, 1, 1.15e-4), (3, 2, 0.0)]:
        ref = O.sweep(db, z_max=1000.0, n=n, save_every=se, gamma=0.0115, alpha=alpha, a0=a0, threads=0)
        for exact in (False, True):
            got = nat.sweep_host(db, n_steps=n, z_max=1000.0, save_every=se, gamma=0.0115, alpha=alpha, a0=a0,
                                 check_nan=True, exact_step=exact, want_traj=(n <= 1005))
            print(f"n={n} se={se} alpha={alpha} exact={exact}: a_end rel {rel(got['a_end'], ref['a_end']):.2e} "
                  f"p_end {rel(got['p_end'], ref['p_end']):.2e} p_max {rel(got['p_max'], ref['p_max']):.2e} "
                  f"bad_eq {np.array_equal(got['first_bad_step'], ref['first_bad_step'])} ms {got['elapsed_ms']:.3f}")
    # trajectory vs oracle single point
    z, A, bad = O.integrate(a0, z_max=100.5, dz=0.1, save_every=10, gamma=0.0115, alpha=1.15e-4, dbeta=0.013)
    got = nat.sweep_host([0.013], n_steps=1005, z_max=100.5, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0,
                         want_traj=True)
    print("traj N=1 rel", rel(got["traj"][0], A), got["traj"].shape, A.shape)
    got = nat.sweep_host([0.013, -0.02, 0.0], n_steps=1005, z_max=100.5, save_every=10, gamma=0.0115, alpha=1.15e-4,
                         a0=a0, want_traj=True)
    print("traj N=3 row0 rel", rel(got["traj"][0], A))
    # blow-up: exact first bad step
    for g in (50.0, 200.0, 1e3):
        ref = O.sweep(np.array([0.01]), z_max=100.0, n=1000, save_every=10, gamma=g, alpha=0.0, a0=a0)
        got = nat.sweep_host([0.01], n_steps=1000, z_max=100.0, save_every=10, gamma=g, alpha=0.0, a0=a0,
                             check_nan=True, exact_step=True)
        gotb = nat.sweep_host([0.01], n_steps=1000, z_max=100.0, save_every=10, gamma=g, alpha=0.0, a0=a0,
                              check_nan=True, exact_step=False)
        print(f"gamma={g}: oracle bad {ref['first_bad_step']} hip exact {got['first_bad_step']} block {gotb['first_bad_step']}")
    # rhs
    rng = np.random.default_rng(1)
    a = rng.normal(size=(64, 4)) + 1j * rng.normal(size=(64, 4))
    zz = rng.uniform(0, 1000, 64); gg = rng.uniform(5e-3, 2e-2, 64); al = rng.uniform(0, 3e-4, 64); dd = rng.uniform(-0.1, 0.1, 64)
    out = nat.yaman_rhs_host(zz, a, gg, al, dd)
    exp = np.array([O.rhs4(zz[i], a[i], gg[i], al[i], dd[i])[0] for i in range(64)])
    print("rhs rel", rel(out, exp))
    gain, bi, bg, nf = nat.gain_summary_host(ref["p_max"], ref["first_bad_step"], 1e-5)
    print("gain summary", gain, bi, bg, nf)
    # f32 and 6-wave smoke
    ref = O.sweep(db, z_max=1000.0, n=10_000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    got = nat.sweep_host(db, n_steps=10_000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0, dtype=np.float32)
    print("f32 vs f64 oracle: a_end rel", rel(got["a_end"].astype(complex), ref["a_end"]), "p_max", rel(got["p_max"].astype(float), ref["p_max"]))
    a06 = np.concatenate([a0, [0, 0]])
    got6 = nat.sweep_host(db, n_steps=10_000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a06, dbeta2=db * 0.5)
    print("6-wave (pair 2 zero) vs 4-wave oracle: a_end rel", rel(got6["a_end"][:, :4], ref["a_end"]), "pair2 max", np.abs(got6["a_end"][:, 4:]).max())


def timing(big):
    a0 = np.sqrt(np.array([0.5, 0.5, 1e-5, 1e-5])).astype(complex)
    for N in (65536, 1048576):   # register-resident vs LDS-staged (A/B of DESIGN.md section 5)
        db = np.linspace(-0.05, 0.05, N)
        for name, fl in (("registers/256", 0), ("registers/64", nat.OPT_BLOCK64), ("LDS-staged/64", nat.OPT_LDS_STAGING)):
            best = min(nat.sweep_host(db, n_steps=10_000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4,
                                      a0=a0, extra_flags=fl)["elapsed_ms"] for _ in range(3))
            print(f"A/B N={N} n=10000 {name}: {best:.2f} ms -> {N * 1e4 / best / 1e6:.2f} G steps/s", flush=True)
    for N in (65536, 262144) + ((1048576,) if big else ()):
        db = np.linspace(-0.05, 0.05, N)
        for n in ((10_000, 100_000) if N == 65536 else (10_000,)):
            for blk in (0, nat.OPT_BLOCK64):
                for chk, ex in ((False, False), (True, False), (True, True)):
                    got = nat.sweep_host(db, n_steps=n, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4,
                                         a0=a0, check_nan=chk, exact_step=ex, extra_flags=blk)
                    ms = got["elapsed_ms"]
                    print(f"N={N} n={n} block={'64' if blk else '256'} check={chk} exact={ex}: {ms:.2f} ms "
                          f"-> {N * n / ms / 1e6:.2f} G steps/s = {4 * N * n / ms / 1e6:.2f} G upd/s "
                          f"= {652 * N * n / ms / 1e9:.2f} TFLOP/s(alg)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    print(nat.version(), "devices:", nat.device_count())
    t = time.time()
    if not args.no_parity:
        parity()
    timing(args.big)
    print("total", time.time() - t)
