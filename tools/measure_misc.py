#!/usr/bin/env python3
"""Developer probe: (1) PCIe-inclusive rate of the host-buffer C-ABI on the bench workload, (2) trajectory mode
(save_every = 1), the HBM-bound regime: kernel time vs bytes written, (3) f32 / 6-wave kernel rates."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psa_amd._native as nat

a0 = np.sqrt(np.array([0.5, 0.5, 1e-5, 1e-5])).astype(complex)
N, n = 65536, 100_000
db = np.linspace(-0.05, 0.05, N)
nat.sweep_host(db[:256], n_steps=100, z_max=1.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)   # context up
for rep in range(3):
    t = time.perf_counter()
    r = nat.sweep_host(db, n_steps=n, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    w = time.perf_counter() - t
    print(f"host API C2: wall {w*1e3:.2f} ms (kernel {r['elapsed_ms']:.2f} ms) -> {4*N*n/w/1e9:.1f} G upd/s PCIe-inclusive", flush=True)

for (Nt, nt, se) in ((65536, 1000, 1), (262144, 400, 1), (65536, 4000, 4), (1048576, 100, 1)):
    dbt = np.linspace(-0.05, 0.05, Nt)
    best = None
    for rep in range(2):
        r = nat.sweep_host(dbt, n_steps=nt, z_max=nt * 0.01, save_every=se, gamma=0.0115, alpha=1.15e-4, a0=a0, want_traj=True)
        best = r["elapsed_ms"] if best is None else min(best, r["elapsed_ms"])
    rows = nt // se + 1
    byts = Nt * rows * 64
    base = min(nat.sweep_host(dbt, n_steps=nt, z_max=nt * 0.01, save_every=se, gamma=0.0115, alpha=1.15e-4, a0=a0)["elapsed_ms"] for _ in range(2))
    print(f"trajectory N={Nt} n={nt} se={se}: kernel {best:.3f} ms (summary-only {base:.3f} ms), {byts/1e9:.2f} GB written "
          f"-> {byts/best/1e6:.0f} GB/s, {Nt*nt/best/1e6:.1f} G steps/s", flush=True)
    del r

for name, kw in (("f32 4-wave", dict(dtype=np.float32)), ("f64 6-wave", dict(dbeta2=db * 0.5))):
    a = a0 if "4-wave" in name else np.concatenate([a0, np.sqrt([2e-5, 1e-6])])
    best = min(nat.sweep_host(db, n_steps=10_000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a, **kw)["elapsed_ms"] for _ in range(3))
    print(f"{name}: N={N} n=10000 kernel {best:.2f} ms -> {N*1e4/best/1e6:.1f} G steps/s = {a.size*N*1e4/best/1e6:.1f} G upd/s")

for Nf in (65536, 131072, 1048576):
    dbf = np.linspace(-0.05, 0.05, Nf)
    for name, fl in (("f32 scalar", nat.OPT_F32_SCALAR), ("f32 packed", nat.OPT_F32_PACKED)):
        best = min(nat.sweep_host(dbf, n_steps=10_000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0,
                                  dtype=np.float32, extra_flags=fl)["elapsed_ms"] for _ in range(3))
        print(f"{name}: N={Nf} n=10000 kernel {best:.2f} ms -> {Nf*1e4/best/1e6:.1f} G steps/s = {4*Nf*1e4/best/1e6:.1f} G upd/s", flush=True)

for Nl in (65536, 1048576):
    dbl = np.linspace(-0.05, 0.05, Nl)
    for name, al in (("alpha = 0 broadcast (lossless instantiation)", 0.0), ("alpha = zeros[N] (generic kernel)", np.zeros(Nl))):
        best = min(nat.sweep_host(dbl, n_steps=10_000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=al, a0=a0)["elapsed_ms"] for _ in range(3))
        print(f"lossless A/B N={Nl}: {name}: {best:.2f} ms -> {Nl*1e4/best/1e6:.1f} G steps/s", flush=True)
