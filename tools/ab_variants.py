#!/usr/bin/env python3
"""Developer probe: kernel times of A/B builds (tools/ab_build.sh) and of the launch variants inside one build.
    python tools/ab_variants.py                 # every ab/libpsa_hip_*.so, each in its own process
    python tools/ab_variants.py --worker [tag]   # the cases below with the library PSA_HIP_LIB names
"""
import glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def worker(tag):
    import psa_amd._native as nat
    import oracle as O
    a4 = np.sqrt(np.array([0.5, 0.5, 1e-5, 1e-5])).astype(complex)
    a6 = np.concatenate([a4, np.sqrt([2e-5, 1e-6])])
    kw = dict(z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4)
    nat.sweep_host(np.zeros(256), n_steps=100, a0=a4, **kw)

    def t(N, n, a0, reps=3, **extra):
        db = np.linspace(-0.05, 0.05, N)
        if a0.size == 6:
            extra["dbeta2"] = db[::-1] * 0.5
        k = dict(kw); k.update(extra)
        return min(nat.sweep_host(db, n_steps=n, a0=a0, **k)["elapsed_ms"] for _ in range(reps))

    def line(name, N, n, nw, ms):
        print(f"[{tag}] {name}: N={N} n={n}: {ms:.3f} ms -> {N*n/ms/1e6:.1f} G steps/s, {nw*N*n/ms/1e6:.1f} G upd/s", flush=True)

    if "--split-only" not in sys.argv:
        line("f64 4w one-lane C2-shape", 65536, 20000, 4, t(65536, 20000, a4))
        line("f64 4w one-lane C3-shape", 1 << 20, 10000, 4, t(1 << 20, 10000, a4))
        line("f64 4w lossless C3-shape", 1 << 20, 10000, 4, t(1 << 20, 10000, a4, alpha=0.0))
        line("f64 6w one-lane 2^18", 1 << 18, 10000, 6, t(1 << 18, 10000, a6))
        ms = t(262144, 100, a4, reps=2, save_every=1, z_max=1.0, want_traj=True)
        print(f"[{tag}] trajectory N=262144 n=100 se=1: {ms:.3f} ms -> {262144*101*64/ms/1e6:.0f} GB/s", flush=True)
        ms = t(1 << 20, 50, a4, reps=2, save_every=1, z_max=0.5, want_traj=True)
        print(f"[{tag}] trajectory N=2^20 n=50 se=1: {ms:.3f} ms -> {(1<<20)*51*64/ms/1e6:.0f} GB/s", flush=True)
    if True:
        for nw, a0 in ((4, a4), (6, a6)):
            for N in (1, 4096, 32768):
                n = 20000
                one = t(N, n, a0, extra_flags=nat.OPT_ONE_LANE)
                two = t(N, n, a0, extra_flags=nat.OPT_SPLIT_POINT)
                print(f"[{tag}] split A/B {nw}w N={N} n={n}: one lane/point {one:.3f} ms, two lanes/point {two:.3f} ms -> x{two/one:.3f}", flush=True)
        # parity of the split kernel against the oracle
        db = np.linspace(-0.05, 0.05, 257)
        for nw, a0 in ((4, a4), (6, a6)):
            ex = dict(dbeta2=db[::-1] * 0.5) if nw == 6 else {}
            ref = O.sweep(db, z_max=1000.0, n=10000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0, **ex)
            for fl, nm in ((nat.OPT_ONE_LANE, "one"), (nat.OPT_SPLIT_POINT, "split")):
                got = nat.sweep_host(db, n_steps=10000, a0=a0, extra_flags=fl, **kw, **ex)
                err = float(np.max(np.abs(got["a_end"] - ref["a_end"]) / np.abs(ref["a_end"])))
                errp = float(np.max(np.abs(got["p_max"] - ref["p_max"]) / ref["p_max"]))
                print(f"[{tag}] parity {nw}w {nm}: a_end rel {err:.2e} p_max rel {errp:.2e} bad {np.array_equal(got['first_bad_step'], ref['first_bad_step'])}", flush=True)


if "--worker" in sys.argv:
    worker(sys.argv[sys.argv.index("--worker") + 1])
else:
    libs = sorted(glob.glob(os.path.join(ROOT, "ab", "libpsa_hip_*.so")))
    for lib in libs:
        tag = os.path.basename(lib)[len("libpsa_hip_"):-3]
        env = dict(os.environ, PSA_HIP_LIB=lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", tag] + [a for a in sys.argv[1:]], env=env, check=False)
